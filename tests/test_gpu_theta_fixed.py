"""draw_theta's log-posterior product in exact fixed point on the int8 matrix cores (gpirt_amd/csrc/theta_fixed.hip)
against the fp64 GEMM it replaces (GPIRT_THETA_FIXED=2) and against long double.

The product  logpost[g, i] = sum_j [y_ij = +1] G+[g, j] + [y_ij = -1] G-[g, j]  (src/draw-theta.cpp:15-19) has a 0/1 operand;
the fixed-point form rounds every term ONCE (to 54 bits of its grid row's range) and adds exactly, the fp64 GEMM rounds every
addition.  Both take the SAME fp64 terms (-log(1 + exp(-+f*)) through the device's exp and log), so the referee here is
those terms (read back through gpirt_debug_ll_term) summed in long double."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NGRID = 1001


def _inputs(n, m, seed, sd=3.0, na=0.07):
    rng = np.random.default_rng(seed)
    fstar = np.asfortranarray(rng.standard_normal((NGRID, m)) * sd)
    y = np.where(rng.random((n, m)) < 0.5, 1.0, -1.0)
    y[rng.random((n, m)) < na] = np.nan
    return np.asfortranarray(y), fstar


def _both(h, y, fstar):
    import torch
    from gpirt_amd.ops import to_device
    yd, fd = to_device(y), to_device(fstar)
    out = []
    for mode in (1, 2):
        with h.config("GPIRT_THETA_FIXED", mode):
            lp, fb = h.theta_logpost(yd, fd)
            out.append((lp.cpu().numpy().copy(), fb))
    torch.cuda.synchronize()
    return out


def _referee(h, y, fstar):
    """the device's own fp64 terms, summed in long double"""
    import torch
    f = torch.from_numpy(np.ascontiguousarray(fstar.ravel(order="F"))).cuda()
    gp = -h.ll_term(f, fast=False).cpu().numpy().reshape(fstar.shape, order="F").astype(np.longdouble)
    gm = -h.ll_term(-f, fast=False).cpu().numpy().reshape(fstar.shape, order="F").astype(np.longdouble)
    yp = (y == 1.0).astype(np.longdouble)
    ym = (y == -1.0).astype(np.longdouble)
    return gp @ yp.T + gm @ ym.T                              # NGRID x n


@pytest.mark.parametrize("n,m", [(300, 77), (256, 1024)])
def test_fixed_point_product_against_long_double(n, m):
    from gpirt_amd.ops import Handle
    h = Handle()
    y, fstar = _inputs(n, m, seed=3 + m)
    (fx, fb1), (ge, fb2) = _both(h, y, fstar)
    assert fb1 == 0 and fb2 == 0
    ref = _referee(h, y, fstar)
    err_fx = np.abs((fx.astype(np.longdouble) - ref).astype(np.float64))
    err_ge = np.abs((ge.astype(np.longdouble) - ref).astype(np.float64))
    # the bound of the header: every term off by at most half a unit of its row's 54-bit grid (2^e above the row's largest
    # term: log(1 + exp(x)) <= x + log 2), one rounding of the result
    e = np.floor(np.log2(np.abs(fstar).max(axis=1) + np.log(2.0))) + 1
    bound = m * 2.0 ** (e - 55)[:, None] + np.spacing(np.abs(ref).astype(np.float64))
    assert (err_fx <= bound).all(), (err_fx.max(), bound.min())
    assert err_fx.max() < 2e-12 and err_ge.max() < 1e-9
    if m >= 512:                                             # many additions: the fp64 chain's rounding shows, the exact sums' does not
        assert err_fx.max() < err_ge.max()
        assert np.sqrt((err_fx ** 2).mean()) < np.sqrt((err_ge ** 2).mean())


@pytest.mark.parametrize("n,m", [(1, 1), (33, 5), (128, 16), (129, 17), (257, 48), (1000, 130), (33024, 3), (4100, 1300)])
def test_shapes_against_the_fp64_product(n, m):
    """ragged respondent tiles, item counts off the 16-column padding, a single item: every entry written, equal to rounding"""
    from gpirt_amd.ops import Handle
    h = Handle()
    y, fstar = _inputs(n, m, seed=11 * n + m)
    (fx, fb1), (ge, _) = _both(h, y, fstar)
    assert fb1 == 0
    assert fx.shape == (NGRID, n) and np.isfinite(fx).all()
    assert np.abs(fx - ge).max() <= 1e-12 * max(1.0, np.abs(ge).max())


def test_the_sums_do_not_depend_on_the_order_of_the_items():
    """exact accumulation: permuting the items leaves every bit of the product alone (the fp64 GEMM's result moves)"""
    from gpirt_amd.ops import Handle
    h = Handle()
    n, m = 200, 300
    y, fstar = _inputs(n, m, seed=5)
    perm = np.random.default_rng(1).permutation(m)
    (fx, _), (ge, _) = _both(h, y, fstar)
    (fxp, _), (gep, _) = _both(h, np.asfortranarray(y[:, perm]), np.asfortranarray(fstar[:, perm]))
    assert np.array_equal(fx, fxp)
    assert not np.array_equal(ge, gep)


@pytest.mark.parametrize("bad", [800.0, -800.0, float("nan"), float("inf")])
def test_rows_that_cannot_be_scaled_hand_over_to_the_fp64_product(bad):
    """|f*| beyond exp()'s range (the formula as written gives -inf there) or a non-finite f*: the device-side flag makes the
    int8 kernel return and the fp64 product behind it run -- bit for bit what GPIRT_THETA_FIXED=2 computes"""
    from gpirt_amd.ops import Handle
    h = Handle()
    n, m = 150, 40
    y, fstar = _inputs(n, m, seed=8)
    fstar[500, 7] = bad
    (fx, fb1), (ge, fb2) = _both(h, y, fstar)
    assert fb1 == 1 and fb2 == 0
    assert np.array_equal(fx, ge, equal_nan=True)
    # and the next product on the same handle is a fixed-point one again (the flag is per launch)
    y2, f2 = _inputs(n, m, seed=9)
    (fx2, fb3), (ge2, _) = _both(h, y2, f2)
    assert fb3 == 0 and np.abs(fx2 - ge2).max() < 1e-11


def test_chain_with_either_product_draws_the_same_theta():
    """two iterations of the whole sampler: theta is an inverse-CDF draw on the grid, so 1e-13 in the log-posterior does
    not move it (a uniform would have to land within ~1e-13 of a step of the CDF)"""
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 1024, 96
    y, th0 = make_responses(n, m, seed=12)
    h = Handle()
    outs = []
    for mode in (1, 2):
        with h.config("GPIRT_THETA_FIXED", mode):
            s = Sampler(h, y, th0, rng="item", seed=5, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
            s.init()
            for _ in range(2):
                s.step()
            outs.append((s.get("theta"), s.get("f")))
            s.close()
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.array_equal(outs[0][1], outs[1][1])


def test_a_block_of_respondents_and_a_shard_of_items_get_the_same_bits():
    """What the sharded hosts rely on: the product for a block of respondents is the matching columns
    of the whole product bit for bit; and the exact partial sums of two item shards -- each with its own row scales -- add up
    to the whole product within the two roundings the partial results carry."""
    import torch
    from gpirt_amd.ops import Handle, to_device
    h = Handle()
    n, m = 700, 96
    y, fstar = _inputs(n, m, seed=17)
    with h.config("GPIRT_THETA_FIXED", 1):
        full, fb = h.theta_logpost(to_device(y), to_device(fstar))
        blk, _ = h.theta_logpost(to_device(np.asfortranarray(y[37:333])), to_device(fstar))
        lo, _ = h.theta_logpost(to_device(np.asfortranarray(y[:, :40])), to_device(np.asfortranarray(fstar[:, :40])))
        hi, _ = h.theta_logpost(to_device(np.asfortranarray(y[:, 40:])), to_device(np.asfortranarray(fstar[:, 40:])))
    full, blk, lo, hi = (t.cpu().numpy() for t in (full, blk, lo, hi))
    assert fb == 0
    assert np.array_equal(full[:, 37:333], blk)
    assert np.abs(lo + hi - full).max() <= 4 * np.spacing(np.abs(full).max())
