"""Full-size (BASELINE.json metric shape, N = 8192 x m = 1024) checks through size-independent
properties -- the oracle cannot run these sizes in seconds, so nothing here calls it:
  * Cholesky: relative residual ||L L^T - S||_F / ||S||_F <= 1e-14 n  (S rebuilt with the HIP kernel)
  * trsm round trips: L (L^-1 B) = B and L^T (L^-T B) = B to 1e-9
  * trmm = the same product through the general GEMM
  * draw_f: every accepted proposal lies on the ellipse through (f, nu) and is NaN free
  * one sampler iteration is bit-reproducible, leaves theta on the -5:0.01:5 grid, and does not depend
    on how the item columns are split between samplers (global-item RNG keys)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N_FULL, M_FULL = 8192, 1024


@pytest.fixture(scope="module")
def problem():
    from gpirt_amd.synthetic import make_responses
    return make_responses(N_FULL, M_FULL, seed=20240)


def test_cholesky_residual_full_size(handle, problem):
    import torch
    from gpirt_amd.ops import to_device
    _, th0 = problem
    th = to_device(th0)
    L = handle.factor(th)
    S = handle.se_kernel(th, th, jitter=0.001)
    R = handle.gemm(L, L, tb=True)                      # L L^T
    resid = (torch.linalg.norm(R - S) / torch.linalg.norm(S)).item()
    assert resid <= 1e-14 * N_FULL, resid
    assert torch.count_nonzero(torch.triu(L, 1)).item() == 0
    assert torch.isfinite(L).all()
    d = torch.diagonal(L)
    assert (d > 0).all() and d.min().item() > 0.03      # sqrt(jitter) = 0.0316 bounds the pivots


def test_trsm_round_trips_full_size(handle, problem):
    import torch
    from gpirt_amd.ops import colmajor, to_device
    _, th0 = problem
    L = handle.factor(to_device(th0))
    torch.manual_seed(0)
    B = colmajor(N_FULL, 192)
    B.normal_()
    for trans in (False, True):
        X = handle.trsm_lower(L, B.clone().T.contiguous().T, trans=trans)
        back = handle.gemm(L, X, ta=trans)
        err = (back - B).abs().max().item()
        assert err <= 1e-9, (trans, err)


def test_trmm_equals_gemm_full_size(handle, problem):
    import torch
    from gpirt_amd.ops import colmajor, to_device
    _, th0 = problem
    L = handle.factor(to_device(th0))
    Z = colmajor(N_FULL, 256)
    Z.normal_()
    a = handle.trmm_lz(L, Z)
    b = handle.gemm(L, Z)
    assert (a - b).abs().max().item() <= 1e-11


def test_draw_f_properties_full_size(handle, problem):
    import torch
    from gpirt_amd.ops import colmajor, to_device
    y, th0 = problem
    m = 256
    yd = to_device(y[:, :m])
    L = handle.factor(to_device(th0))
    f = handle.trmm_lz(L, handle.item_normals(3, 0, 1, 0, m, N_FULL))
    mu = colmajor(N_FULL, m, fill=0.0)
    f0 = f.clone().T.contiguous().T
    ll0 = handle.ll_bar(f0, yd, mu)
    out, k = handle.draw_f(f, yd, L, mu, seed=9, it=1)
    assert torch.isfinite(out).all() and (k >= 0).all() and k.float().mean().item() < 20
    # f' = f cos(eps) + nu sin(eps)  =>  for nu = L z the point stays on the ellipse: check via a second
    # identical call (bit-reproducible) and via the slice condition ll(f') > ll(f) + log(u) >= -inf
    f2 = f0.clone().T.contiguous().T
    out2, k2 = handle.draw_f(f2, yd, L, mu, seed=9, it=1)
    assert torch.equal(out, out2) and torch.equal(k, k2)
    ll1 = handle.ll_bar(out, yd, mu)
    assert torch.isfinite(ll1).all() and (ll1 > ll0 - 40.0).all()      # log(u) > -40 for a 52-bit uniform


def test_iteration_reproducible_and_shard_invariant(handle, problem):
    from gpirt_amd import Sampler
    y, th0 = problem
    m = 128                                               # keeps three samplers resident comfortably
    a = Sampler(handle, y[:, :m], th0, rng="item", seed=5)
    b = Sampler(handle, y[:, :m], th0, rng="item", seed=5)
    for s in (a, b):
        s.init()
        s.step()
        s.check()
    fa, fb = a.get("f"), b.get("f")
    assert np.array_equal(fa, fb) and np.array_equal(a.get("theta"), b.get("theta"))
    th = a.get("theta")
    kk = (th + 5.0) / 0.01
    assert np.abs(kk - np.rint(kk)).max() < 1e-9 and th.min() > -5.0
    # item columns [64, 128) as their own shard: same draws for those columns given the same theta path
    c = Sampler(handle, y[:, 64:m], th0, rng="item", seed=5, item0=64, m_total=m)
    c.init()
    assert np.abs(c.get("f") - a_init_f(handle, y, th0, m)[:, 64:]).max() < 1e-12
    for s in (a, b, c):
        s.close()


def a_init_f(handle, y, th0, m):
    from gpirt_amd import Sampler
    s = Sampler(handle, y[:, :m], th0, rng="item", seed=5)
    s.init()
    f = s.get("f")
    s.close()
    return f
