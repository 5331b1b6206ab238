"""The algebra behind the structured pass of the R-stream predictor (gpirt_amd/csrc/rs_lr.hip), in NumPy on the CPU.

S = K(theta, theta) + eps I with the reference's unit squared-exponential kernel (src/covariance-function.cpp:10) and jitter
(src/gpirtMCMC.cpp:16).  With the Lagrange basis V of r Chebyshev nodes c on [-5, 5], K = V K(c, theta) to rounding, and the
blocks of chol(S) below a block J of columns are V C_J with C_J = D_J V_J^T L_JJ^-T, where D_J = eps (eps I + M G_J)^-1 M is a
function of the PREFIX GRAM G_J = sum_{i < J} V[i]^T V[i] alone (M = K(c, c)).  The HIP kernels do exactly this per 64-column
block; what they produce only PREDICTS (the exact phase verifies every count), so this test pins the mathematics, not a
tolerance of the product path."""
import numpy as np
import pytest
import scipy.linalg as sla

EPS = 1e-3
R = 64


def _basis(theta, r=R):
    k = np.arange(r)
    c = 5.0 * np.cos((2 * k + 1) * np.pi / (2 * r))
    w = (-1.0) ** k * np.sin((2 * k + 1) * np.pi / (2 * r))
    V = np.zeros((len(theta), r))
    for i, t in enumerate(np.clip(theta, -5.0, 5.0)):
        d = t - c
        if np.any(d == 0.0):
            V[i, np.argmax(d == 0.0)] = 1.0
        else:
            q = w / d
            V[i] = q / q.sum()
    return c, V


@pytest.mark.parametrize("order", ["random", "sorted", "two_values"])
def test_blocks_below_the_diagonal_are_low_rank_in_the_lagrange_basis(order):
    rng = np.random.default_rng(3)
    n, B = 768, 64
    theta = np.clip(np.round(rng.standard_normal(n), 2), -5, 5)          # draw_theta's values lie on the grid -5:0.01:5
    if order == "sorted":
        theta = np.sort(theta)
    elif order == "two_values":
        theta = np.where(np.arange(n) % 3 == 0, -1.25, 2.5)
    K = np.exp(-0.5 * (theta[:, None] - theta[None, :]) ** 2)
    L = np.linalg.cholesky(K + EPS * np.eye(n))
    c, V = _basis(theta)
    assert np.abs(V @ np.exp(-0.5 * (c[:, None] - theta[None, :]) ** 2) - K).max() < 1e-13
    M = np.exp(-0.5 * (c[:, None] - c[None, :]) ** 2)
    G = np.zeros((R, R))
    worst = 0.0
    for b in range(n // B):
        J = slice(b * B, (b + 1) * B)
        D = EPS * np.linalg.solve(EPS * np.eye(R) + M @ G, M)
        D = 0.5 * (D + D.T)
        # the block's diagonal block of the factor IS the Cholesky factor of the Schur complement carried in the basis
        A = V[J] @ D @ V[J].T + EPS * np.eye(B)
        assert np.abs(np.linalg.cholesky(0.5 * (A + A.T)) - np.tril(L[J, J])).max() < 1e-9
        C = sla.solve_triangular(L[J, J], (D @ V[J].T).T, lower=True).T            # r x B
        lo = (b + 1) * B
        if lo < n:
            worst = max(worst, np.abs(V[lo:] @ C - L[lo:, J]).max())
        G += V[J].T @ V[J]
    assert worst < 1e-8, worst


def test_structured_product_in_single_precision_is_a_usable_prediction():
    """nu = L z from the diagonal 512-column parts + V (prefix of C_J z_J), everything rounded to float as the kernels do: a few
    1e-7 of max|nu| -- the same order as a dense single-precision product, far below what decides a slice loop."""
    rng = np.random.default_rng(4)
    n, B, P = 1536, 64, 512
    theta = np.clip(np.round(rng.standard_normal(n), 2), -5, 5)
    K = np.exp(-0.5 * (theta[:, None] - theta[None, :]) ** 2)
    L = np.linalg.cholesky(K + EPS * np.eye(n))
    c, V = _basis(theta)
    M = np.exp(-0.5 * (c[:, None] - c[None, :]) ** 2)
    G = np.zeros((R, R))
    C = np.zeros((R, n))
    for b in range(n // B):
        J = slice(b * B, (b + 1) * B)
        D = EPS * np.linalg.solve(EPS * np.eye(R) + M @ G, M)
        C[:, J] = sla.solve_triangular(L[J, J], (0.5 * (D + D.T) @ V[J].T).T, lower=True).T
        G += V[J].T @ V[J]
    z = rng.standard_normal((n, 4))
    nu = L @ z
    V32, C32, L32, z32 = V.astype(np.float32), C.astype(np.float32), L.astype(np.float32), z.astype(np.float32)
    out = np.zeros((n, 4), np.float32)
    pref = np.zeros((R, 4), np.float32)
    for I in range(n // P):
        rows = slice(I * P, (I + 1) * P)
        out[rows] = np.tril(L32[rows, rows]) @ z32[rows] + V32[rows] @ pref
        pref = pref + C32[:, rows] @ z32[rows]
    assert np.abs(out - nu).max() / np.abs(nu).max() < 5e-6
