"""GPU parity tests, operator level: every C-ABI operator against the CPU oracle on seeded inputs.

Tolerances (fp64): element-wise kernels 1e-14 relative (libm exp/log may differ by an ulp between
glibc and the ROCm device library); Cholesky factor max|L - L_ref| <= 1e-11 and relative residual
<= 1e-14 * n; solves / products 1e-10 absolute.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _theta_grid(n, seed):
    rng = np.random.default_rng(seed)
    k = np.clip(np.rint((rng.standard_normal(n) + 5.0) / 0.01), 0, 1000)
    return -5.0 + k * 0.01


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(128, 128, 64), (257, 131, 77), (1001, 300, 513), (64, 1, 200), (5, 7, 3)])
def test_gemm_matches_numpy(handle, ta, tb, shape):
    from gpirt_amd.ops import to_device, to_host
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    C0 = rng.standard_normal((M, N))
    Cd = to_device(C0)
    out = handle.gemm(to_device(A), to_device(B), ta=ta, tb=tb, alpha=-1.5, beta=0.5, C_out=Cd)
    ref = -1.5 * (A.T if ta else A) @ (B.T if tb else B) + 0.5 * C0
    assert np.abs(to_host(out) - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("K", [16, 32, 48, 64, 80, 112, 176])
def test_gemm_pipelined_loop_64_tiles(handle, ta, tb, K):
    """Interior 64-tiles take the branch-free software-pipelined K loop (register ring of 2): every prologue /
    steady-state / tail combination of K-steps (K = 16 ... 176 = 1 ... 11 steps), plus one ragged block row."""
    from gpirt_amd.ops import to_device, to_host
    M, N = 192 + 5, 128
    rng = np.random.default_rng(K + 2 * ta + tb)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    C0 = rng.standard_normal((M, N))
    out = handle.gemm(to_device(A), to_device(B), ta=ta, tb=tb, alpha=-1.0, beta=1.0, C_out=to_device(C0))
    ref = C0 - (A.T if ta else A) @ (B.T if tb else B)
    assert np.abs(to_host(out) - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("K", [16, 32, 48, 96])
def test_gemm_pipelined_loop_128_tiles(handle, ta, tb, K):
    """The same for the 128-tile kernel (chosen from 448 tiles on): 22 x 22 interior tiles + a ragged edge."""
    from gpirt_amd.ops import to_device, to_host
    M, N = 22 * 128 + 40, 22 * 128
    rng = np.random.default_rng(K + 2 * ta + tb + 100)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    out = handle.gemm(to_device(A), to_device(B), ta=ta, tb=tb)
    ref = (A.T if ta else A) @ (B.T if tb else B)
    assert np.abs(to_host(out) - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_gemm_mfma_layout_asymmetric(handle):
    """A = I with an asymmetric B catches a transposed C/D lane map (guide section 3)."""
    from gpirt_amd.ops import to_device, to_host
    n = 256
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) / 7.0
    out = handle.gemm(to_device(np.eye(n)), to_device(B))
    assert np.array_equal(to_host(out), B)


@pytest.mark.parametrize("n1,n2", [(100, 100), (1024, 1001), (513, 77)])
def test_se_kernel(handle, oracle, n1, n2):
    from gpirt_amd.ops import to_device, to_host
    rng = np.random.default_rng(n1 + n2)
    x1, x2 = rng.standard_normal(n1) * 2, rng.standard_normal(n2) * 2
    K = to_host(handle.se_kernel(to_device(x1), to_device(x2), jitter=0.001 if n1 == n2 else 0.0))
    ref = oracle.se_kernel(x1, x2)
    if n1 == n2:
        ref[np.diag_indices(n1)] += 0.001
    assert np.abs(K - ref).max() <= 4e-16


@pytest.mark.parametrize("n", [8, 64, 100, 200, 256, 300, 1000, 1024])
def test_potrf_matches_oracle(handle, oracle, n):
    from gpirt_amd.ops import to_device, to_host
    theta = _theta_grid(n, n)
    Lref, info = oracle.factor(theta)
    assert info == 0
    L = to_host(handle.factor(to_device(theta)))
    assert np.array_equal(np.triu(L, 1), np.zeros_like(L))
    S = oracle.se_kernel(theta, theta)
    S[np.diag_indices(n)] += 0.001
    resid = np.linalg.norm(L @ L.T - S) / np.linalg.norm(S)
    assert resid <= 1e-14 * n
    assert np.abs(L - Lref).max() <= 1e-11


def _oracle_factor_blocked(oracle, theta):
    """arma::chol(K + 0.001 I, "lower") by the oracle's blocked all-core potrf (seconds up to n ~ 3000)."""
    S = oracle.se_kernel(theta, theta)
    S[np.diag_indices(len(theta))] += 0.001
    L, info = oracle.potrf_lower(S, blocked=True)
    assert info == 0
    return np.tril(L)


@pytest.mark.parametrize("n,G", [(5000, 3), (8192, 8), (1500, 2), (900, 4), (2600, 3)])
def test_potrf_in_pieces_is_bit_identical(handle, oracle, n, G):
    """the factorisation as a distributing host drives it -- 1-D block-cyclic ownership of the outer
    panels over G ranks (played here by G copies of S on the one device), owner factors, the panel travels through the
    dense buffer, every rank applies it to the block columns it owns -- must reproduce gpirt_potrf_lower BIT FOR BIT on
    every rank (ragged last panel; more ranks than panels; a single panel)."""
    import torch
    from gpirt_amd.ops import to_device
    from gpirt_amd.synthetic import make_responses
    _, th0 = make_responses(n, 2, seed=n)
    th = to_device(th0)
    ref = handle.factor(th)
    W = handle.panel_width
    NP = (n + W - 1) // W
    ranks = [handle.se_kernel(th, th, jitter=0.001) for _ in range(G)]           # <= 8 x 0.5 GiB
    buf = torch.empty(n * min(W, n), dtype=torch.float64, device="cuda")
    handle.potrf_begin()
    for p in range(NP):
        own = p % G
        handle.potrf_panel_factor(ranks[own], p)
        handle.potrf_panel_copy(ranks[own], p, buf, True)
        for r in range(G):
            if r != own:
                handle.potrf_panel_copy(ranks[r], p, buf, False)
            for c in range(p + 1, NP):
                if c % G == r:
                    handle.potrf_panel_update(ranks[r], p, c)
    handle.potrf_finish()
    for r in range(G):
        assert torch.equal(torch.tril(ranks[r]), torch.tril(ref)), f"rank {r} differs from the single-GPU factor"
    if n <= 2600:          # ... and the assembled factor is the ORACLE's (src/gpirtMCMC.cpp:17), not only the single-GPU one
        Lo = _oracle_factor_blocked(oracle, th0)
        for r in range(G):
            assert np.abs(np.tril(ranks[r].cpu().numpy()) - Lo).max() <= 1e-11, f"rank {r} differs from the oracle's factor"


@pytest.mark.parametrize("n,G,panel", [(5000, 3, 1), (8192, 8, 1), (2600, 2, 1), (1500, 3, 1), (900, 2, 1), (2600, 2, 2), (1500, 3, 2)])
def test_potrf_in_half_panels_is_bit_identical(handle, oracle, n, G, panel):
    """The pipelined form of the same host (round 3): an outer panel travels as its first sub-panel and the rest, and the
    next owner applies the first half's share of the update of its first columns before the second half exists
    (gpirt_potrf_panel_*_part, in the order gpirt_amd/distributed.py issues them).  Same launches, same order: every
    rank must again hold gpirt_potrf_lower's factor BIT FOR BIT -- and, at the sizes the oracle reaches in seconds, the
    ORACLE's factor to 1e-11.  panel = 2: the same with the launch-per-step panel (GPIRT_PANEL=2, what ShardedSampler(chol=
    "distributed") runs when more than two ranks share a card, and what a hang-guard fallback refactors with)."""
    import torch
    from gpirt_amd.ops import to_device
    from gpirt_amd.synthetic import make_responses
    _, th0 = make_responses(n, 2, seed=n)
    th = to_device(th0)
    old_panel = handle.config_get("GPIRT_PANEL")
    handle.config_set("GPIRT_PANEL", panel)
    try:
        _half_panels_body(handle, oracle, n, G, th, th0)
    finally:
        handle.config_set("GPIRT_PANEL", old_panel)


def _half_panels_body(handle, oracle, n, G, th, th0):
    import torch
    ref = handle.factor(th)
    W, H = handle.panel_width, handle.subpanel_width(n)
    NP = (n + W - 1) // W
    ranks = [handle.se_kernel(th, th, jitter=0.001) for _ in range(G)]
    buf = torch.empty(n * min(H, n), dtype=torch.float64, device="cuda")
    own = lambda p: p % G

    def travel(p, half):
        handle.potrf_panel_copy_part(ranks[own(p)], p, half, buf, True)
        for r in range(G):
            if r != own(p):
                handle.potrf_panel_copy_part(ranks[r], p, half, buf, False)

    handle.potrf_begin()
    handle.potrf_panel_factor_part(ranks[own(0)], 0, 0)
    travel(0, 0)
    handle.potrf_panel_factor_part(ranks[own(0)], 0, 1)
    for p in range(NP):
        nxt = p + 1
        if nxt < NP:
            handle.potrf_panel_update_part(ranks[own(nxt)], p, nxt, 0)      # before B(p) has arrived anywhere
        travel(p, 1)
        if nxt >= NP:
            break
        handle.potrf_panel_update_part(ranks[own(nxt)], p, nxt, 1)
        handle.potrf_panel_factor_part(ranks[own(nxt)], nxt, 0)
        travel(nxt, 0)
        handle.potrf_panel_factor_part(ranks[own(nxt)], nxt, 1)
        for c in range(nxt + 1, NP):
            handle.potrf_panel_update_part(ranks[own(c)], p, c, 2)
    handle.potrf_finish()
    for r in range(G):
        assert torch.equal(torch.tril(ranks[r]), torch.tril(ref)), f"rank {r} differs from the single-GPU factor"
    if n <= 2600:
        Lo = _oracle_factor_blocked(oracle, th0)
        for r in range(G):
            assert np.abs(np.tril(ranks[r].cpu().numpy()) - Lo).max() <= 1e-11, f"rank {r} differs from the oracle's factor"


def test_potrf_operator_on_user_matrix(handle, oracle):
    from gpirt_amd.ops import to_device, to_host
    n = 200
    rng = np.random.default_rng(3)
    A = rng.standard_normal((n, n))
    S = A @ A.T + n * np.eye(n)
    L = to_host(handle.potrf_lower(to_device(S)))
    assert np.abs(L - np.linalg.cholesky(S)).max() <= 1e-11


def test_potrf_not_positive_definite_raises(handle):
    from gpirt_amd.ops import to_device
    S = np.eye(70)
    S[40, 40] = -1.0
    with pytest.raises(RuntimeError, match="decomposition failed"):
        handle.potrf_lower(to_device(S))


# (512, 256) .. (1339, 300): forward solves whose full 256-row leaves go through block inverses -- an even
# and an odd number of 256-blocks, 512-aligned and unaligned leaves, a ragged tail
@pytest.mark.parametrize("n,m", [(100, 7), (300, 130), (1024, 256), (512, 256), (768, 257), (1339, 300), (1792, 256)])
def test_trmm_trsm(handle, oracle, n, m):
    from gpirt_amd.ops import to_device, to_host
    theta = _theta_grid(n, 5)
    Lref, _ = oracle.factor(theta)
    Ld = to_device(Lref)
    rng = np.random.default_rng(n)
    Z = rng.standard_normal((n, m))
    out = to_host(handle.trmm_lz(Ld, to_device(Z)))
    assert np.abs(out - Lref @ Z).max() <= 1e-11
    for trans in (False, True):
        X = to_host(handle.trsm_lower(Ld, to_device(Z), trans=trans))
        ref = oracle.trsm_lower(Lref, Z, trans=trans)
        scale = np.abs(ref).max()
        assert np.abs(X - ref).max() <= 1e-10 * scale
        back = (Lref.T if trans else Lref) @ X
        assert np.abs(back - Z).max() <= 1e-9


def test_item_rng_matches_oracle(handle, oracle):
    from gpirt_amd.ops import to_host
    seed, it, stage = 0x1234567890ABCDEF, 3, oracle.ST_F_Z
    U = to_host(handle.item_uniforms(seed, it, stage, 5, 4, 33))
    Z = to_host(handle.item_normals(seed, it, stage, 5, 4, 33))
    for j in range(4):
        for i in range(33):
            u = oracle.item_uniform(seed, it, stage, 5 + j, i)
            assert U[i, j] == u
            assert abs(Z[i, j] - oracle.qnorm(u)) <= 4e-15 * max(1.0, abs(Z[i, j]))


def test_ll_bar(handle, oracle):
    from gpirt_amd.ops import to_device
    from gpirt_amd.synthetic import make_responses
    n, m = 300, 9
    y, _ = make_responses(n, m, seed=11)
    rng = np.random.default_rng(0)
    f, mu = rng.standard_normal((n, m)), rng.standard_normal((n, m))
    got = handle.ll_bar(to_device(f), to_device(y), to_device(mu)).cpu().numpy()
    for j in range(m):
        ref = oracle.ll_bar(f[:, j], y[:, j], mu[:, j])
        assert abs(got[j] - ref) <= 1e-12 * abs(ref)
    got = handle.ll_bar(to_device(f), to_device(y)).cpu().numpy()
    assert abs(got[0] - oracle.ll(f[:, 0], y[:, 0])) <= 1e-12 * abs(got[0])


def _problem(n, m, seed):
    from gpirt_amd.synthetic import make_responses
    y, theta = make_responses(n, m, seed=seed)
    rng = np.random.default_rng(seed + 1)
    f = rng.standard_normal((n, m))
    beta = np.vstack([rng.uniform(-1, 1, m), rng.uniform(0.5, 2, m)])
    mu = beta[0][None, :] + theta[:, None] * beta[1][None, :]
    return y, theta, f, beta, mu


@pytest.mark.parametrize("n,m", [(100, 5), (520, 33)])
def test_draw_f_item_rng(handle, oracle, n, m):
    from gpirt_amd.ops import to_device, to_host
    y, theta, f, beta, mu = _problem(n, m, 21)
    L, _ = oracle.factor(theta)
    seed, it = 77, 4
    rng = oracle.ItemStream(seed)
    ref, kref = oracle.draw_f(rng, f, y, L, mu, it=it)
    fd = to_device(f)
    out, k = handle.draw_f(fd, to_device(y), to_device(L), to_device(mu), seed, it)
    assert np.array_equal(k.cpu().numpy(), kref)
    assert np.abs(to_host(out) - ref).max() <= 1e-9


@pytest.mark.parametrize("n,m", [(9000, 4), (16000, 3)])
def test_draw_f_item_rng_beyond_8192_rows(handle, oracle, n, m):
    """The slice kernel keeps a column in registers up to 8192 rows with four arrays (f, nu, mu, y) and up to 16384 with three
    (y f, y nu, y mu: y is +-1, so the folded form is the same bits; rng_ess.hip, FOLD): draw_f against the oracle's at n = 9000
    and 16000 on a LAPACK factor (scipy dpotrf -- the oracle's own unblocked factorisation would take minutes here): rejection
    counts exact, f to 1e-9."""
    import scipy.linalg as sl
    from gpirt_amd.ops import to_device, to_host
    y, theta, f, beta, mu = _problem(n, m, 33)
    K = np.exp(-0.5 * (theta[:, None] - theta[None, :]) ** 2)
    K[np.diag_indices(n)] += 1e-3
    L = np.asfortranarray(np.tril(sl.cholesky(K, lower=True, overwrite_a=True, check_finite=False)))
    del K
    seed, it = 5, 2
    ref, kref = oracle.draw_f(oracle.ItemStream(seed), f, y, L, mu, it=it)
    out, k = handle.draw_f(to_device(f), to_device(y), to_device(L), to_device(mu), seed, it)
    assert np.array_equal(k.cpu().numpy(), kref)
    assert np.abs(to_host(out) - ref).max() <= 1e-9


@pytest.mark.parametrize("fused", [False, True])
def test_draw_fstar_item_rng(handle, oracle, fused):
    from gpirt_amd.ops import to_device, to_host
    n, m = 300, 6
    y, theta, f, beta, mu = _problem(n, m, 5)
    L, _ = oracle.factor(theta)
    f = L @ np.random.default_rng(1).standard_normal((n, m))      # a GP-plausible f
    ts = oracle.theta_star()
    mu_star = beta[0][None, :] + ts[:, None] * beta[1][None, :]
    seed, it = 99, 2
    ref, sref, meanref = oracle.draw_fstar(oracle.ItemStream(seed), f, theta, L, mu_star, it=it)
    out, s, mean = handle.draw_fstar(to_device(f), to_device(theta), to_device(L), to_device(mu_star), seed, it, fused=fused)
    assert np.abs(s.cpu().numpy() - sref).max() <= 1e-9
    assert np.abs(to_host(mean) - meanref).max() <= 1e-9
    assert np.abs(to_host(out) - ref).max() <= 1e-9


@pytest.mark.parametrize("stabilise", [False, True])
def test_draw_theta_item_rng(handle, oracle, stabilise):
    from gpirt_amd.ops import to_device
    n, m = 200, 12
    y, theta, f, beta, mu = _problem(n, m, 8)
    ts = oracle.theta_star()
    fstar = beta[0][None, :] + ts[:, None] * beta[1][None, :] + 0.1 * np.sin(ts)[:, None]
    seed, it = 5, 9
    ref, deg = oracle.draw_theta(oracle.ItemStream(seed), y, fstar, it=it, stabilise=stabilise)
    out, degd = handle.draw_theta(to_device(y), to_device(fstar), seed, it, stabilise=stabilise)
    assert deg == 0 and degd == 0
    assert np.array_equal(out.cpu().numpy(), ref)       # theta is a grid value: exact


def test_draw_beta_item_rng(handle, oracle):
    from gpirt_amd.ops import to_device, to_host
    n, m = 257, 10
    y, theta, f, beta, mu = _problem(n, m, 13)
    pm, ps, st = np.zeros((2, m)), np.full((2, m), 3.0), np.full((2, m), 0.1)
    st[0, 3] = 0.0                                       # rnorm(mu, 0) consumes nothing
    seed, it = 31, 6
    ref = oracle.draw_beta(oracle.ItemStream(seed), beta, theta, y, f, pm, ps, st, it=it)
    bd = to_device(beta)
    handle.draw_beta(bd, to_device(theta), to_device(y), to_device(f), to_device(pm), to_device(ps), to_device(st), seed, it)
    assert np.abs(to_host(bd) - ref).max() <= 1e-12


def test_calibrate_mfma(handle):
    tf = handle.calibrate_mfma_f64()
    print("fp64 MFMA peak (measured):", tf, "TFLOP/s")
    assert 20.0 < tf < 200.0
