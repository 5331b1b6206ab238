"""GPU parity tests, sampler level: whole gpirtMCMC() runs against the CPU oracle.

 * rng="reference": the HIP path replays R's Mersenne-Twister stream; every stored draw (theta, beta,
   f) and the IRFs must match the oracle's restatement of src/gpirtMCMC.cpp draw for draw
   (f, f*: abs 1e-9; beta 1e-9; theta exact -- grid values), and the R stream must end in the same state.
 * rng="item": same, against the oracle run with the same counter-based sub-streams.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(res, ref, n_exact_theta=True):
    assert np.array_equal(res["theta"], ref["theta"]) if n_exact_theta else True
    assert np.abs(res["beta"] - ref["beta"]).max() <= 1e-9
    assert np.abs(res["f"] - ref["f"]).max() <= 1e-9
    assert np.abs(res["IRFs"] - ref["IRFs"]).max() <= 1e-9


@pytest.mark.parametrize("n,m,S,B", [(40, 6, 3, 2), (100, 17, 2, 1), (130, 5, 2, 0)])
def test_mcmc_reference_rng_matches_oracle(handle, oracle, n, m, S, B):
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, _ = make_responses(n, m, seed=7 + n, snap_theta=False)
    rs = RStream(1234)
    res = gpirtMCMC(y, S, B, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=rs)
    r = oracle.RStream(1234)
    th0 = r.rnorm(n)
    ref = oracle.gpirt_mcmc(r, y, th0, S, B)
    assert np.array_equal(res["theta"][0], th0)
    _check(res, ref)
    mt, mti = rs.state()
    mt_ref, mti_ref = r.mt_state()
    assert mti == mti_ref and np.array_equal(mt, mt_ref)


# rank 64: draw_fstar through the Chebyshev factorisation of K(theta, theta*) -- the oracle solves all 1001 columns
@pytest.mark.parametrize("n,m,S,B,fused,rank", [(64, 9, 3, 1, False, 0), (200, 20, 2, 2, False, 0), (200, 20, 2, 2, True, 0),
                                                (200, 20, 2, 2, True, 64), (600, 12, 2, 1, True, 64)])
def test_mcmc_item_rng_matches_oracle(handle, oracle, n, m, S, B, fused, rank):
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=3 + n)
    seed = 4242
    res = gpirtMCMC(y, S, B, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), theta_init=th0, rng="item",
                    seed=seed, theta_stabilise=True, fstar_fused=fused, kstar_rank=rank)
    ref = oracle.gpirt_mcmc(oracle.ItemStream(seed), y, th0, S, B, theta_stabilise=True, fstar_fused=fused)
    _check(res, ref)


@pytest.mark.parametrize("fused,rank", [(True, 0), (True, 64)])
def test_mcmc_reference_rng_with_the_cheaper_fstar_forms(handle, oracle, fused, rank):
    """R's stream replayed (the default RNG contract) TOGETHER with the algebraically identical forms of draw_fstar
    (options(gpirt.hip.fstar_fused = TRUE, gpirt.hip.kstar_rank = 64) under the default gpirt.hip.rng): the stream is consumed
    exactly as the reference consumes it -- the 1001 x m predictive normals do not depend on how their means were computed --
    so the chain is the oracle's (its `fused` wording of src/draw-fstar.cpp:17-25) to the stated tolerance and the
    generator ends where the oracle's does."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m, S, B = 256, 14, 2, 2
    y, _ = make_responses(n, m, seed=77, snap_theta=False)
    rs = RStream(321)
    res = gpirtMCMC(y, S, B, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=rs,
                    fstar_fused=fused, kstar_rank=rank)
    r = oracle.RStream(321)
    th0 = r.rnorm(n)
    ref = oracle.gpirt_mcmc(r, y, th0, S, B, fstar_fused=True)
    _check(res, ref)
    mt, mti = rs.state()
    mt_ref, mti_ref = r.mt_state()
    assert mti == mti_ref and np.array_equal(mt, mt_ref)


def test_mcmc_fast_preset_is_the_item_rng_rank64_chain(handle, oracle):
    """preset="fast" = gpirt_fast_options() through the drop-in (what bench.py times): the same draws as the options spelled
    out one by one, and the oracle's to the stated tolerance."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.synthetic import make_responses
    n, m, S, B, seed = 200, 20, 2, 2, 4242
    y, th0 = make_responses(n, m, seed=3 + n)
    codes = dict(yea=[1], nay=[-1], missing=[None])
    a = gpirtMCMC(y, S, B, vote_codes=codes, theta_init=th0, preset="fast", seed=seed)
    b = gpirtMCMC(y, S, B, vote_codes=codes, theta_init=th0, rng="item", seed=seed, theta_stabilise=True, fstar_fused=True,
                  kstar_rank=64)
    for k in ("theta", "beta", "f", "IRFs"):
        assert np.array_equal(a[k], b[k]), k
    ref = oracle.gpirt_mcmc(oracle.ItemStream(seed), y, th0, S, B, theta_stabilise=True, fstar_fused=True)
    _check(a, ref)


def test_senate116_plumbing(handle, oracle):
    """Config C1: the senate116-derived matrix (n=100, m=418) through the drop-in, 3 iterations."""
    import os
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "senate116_y.npz"))
    y = d["y"].astype(np.float64)
    y[y == 0] = np.nan
    res = gpirtMCMC(y, 2, 1, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=RStream(1119))
    r = oracle.RStream(1119)
    th0 = r.rnorm(100)
    ref = oracle.gpirt_mcmc(r, y, th0, 2, 1)
    _check(res, ref)
    assert res["IRFs"].shape == (1001, 418) and np.all((res["IRFs"] >= 0) & (res["IRFs"] <= 1))


def test_senate116_one_hundred_iterations_under_r_stream(handle, oracle):
    """Config C1 AS BASELINE.json STATES IT: senate116 (n = 100, m = 418), 100 iterations (60 burn-in + 40 sampled), the
    default contract -- R's Mersenne-Twister stream replayed draw for draw, src/draw-fstar.cpp and src/draw-theta.cpp as
    written -- against the CPU restatement of the reference: every stored theta, beta, f and the IRFs."""
    import os
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "senate116_y.npz"))
    y = d["y"].astype(np.float64)
    y[y == 0] = np.nan
    res = gpirtMCMC(y, 40, 60, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=RStream(1119))
    r = oracle.RStream(1119)
    th0 = r.rnorm(100)
    ref = oracle.gpirt_mcmc(r, y, th0, 40, 60)
    _check(res, ref)
    assert res["theta"].shape == (41, 100) and res["f"].shape == (100, 418, 41)


def test_sampler_stage_api_equals_step(handle, oracle):
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 150, 11
    y, th0 = make_responses(n, m, seed=9)
    a = Sampler(handle, y, th0, rng="item", seed=5)
    b = Sampler(handle, y, th0, rng="item", seed=5)
    a.init(); b.init()
    for _ in range(2):
        a.step()
        b.draw_f(); b.draw_fstar(); b.theta_partial(); b.theta_finish(); b.draw_beta(); b.factor()
    a.check(); b.check()
    for name in ("theta", "f", "beta", "fstar", "L"):
        assert np.array_equal(a.get(name), b.get(name))
    assert a.iteration == b.iteration == 2
    a.close(); b.close()


@pytest.mark.parametrize("n,m", [(150, 11), (1200, 40)])
def test_theta_by_respondent_blocks_equals_theta_by_items(handle, n, m):
    """The item-sharded runs draw theta per block of respondents from the gathered f* (gpirt_sampler_theta_block).
    Two 'ranks' on one device -- each owning half of the item columns, one of two unequal respondent blocks -- must
    reproduce the single sampler's theta bit for bit, through the same collectives ShardedSampler issues."""
    import torch
    from gpirt_amd import Sampler
    from gpirt_amd.distributed import item_range
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=21)
    one = Sampler(handle, y, th0, rng="item", seed=5)
    one.init()
    ranks = []
    cut = [0, n // 3, n]
    for r in range(2):
        lo, hi = item_range(m, r, 2)
        e = Sampler(handle, y[:, lo:hi], th0, rng="item", seed=5, item0=lo, m_total=m)
        e.set_theta_block(y[cut[r]:cut[r + 1], :], cut[r], m)
        e.init()
        ranks.append((e, lo, hi))
    N = 1001
    for _ in range(2):
        one.step()
        for e, lo, hi in ranks:
            e.draw_f(); e.draw_fstar()
        for e, _, _ in ranks:                                  # "all-gather" of the f* columns
            full = e.device_tensor("fstar_full")
            for src, lo, hi in ranks:
                full[N * lo: N * hi].copy_(src.device_tensor("fstar")[: N * (hi - lo)])
        for e, _, _ in ranks:
            e.theta_block()
        total = sum(e.device_tensor("theta_stage").clone() for e, _, _ in ranks)      # "all-reduce"
        for e, _, _ in ranks:
            e.device_tensor("theta_stage").copy_(total)
            e.theta_commit(); e.draw_beta(); e.factor()
        torch.cuda.synchronize()
        for e, lo, hi in ranks:
            e.check()
            assert np.array_equal(e.get("theta"), one.get("theta"))
            assert np.array_equal(e.get("f"), one.get("f")[:, lo:hi])
    for e, _, _ in ranks:
        e.close()
    one.close()


@pytest.mark.parametrize("n,m,S,B", [(5, 1, 1, 0), (64, 2, 1, 1), (65, 3, 0, 1), (257, 4, 1, 0)])
def test_mcmc_edge_shapes(handle, oracle, n, m, S, B):
    """Tiny / ragged problems (n below, at and just above a 64-column panel; a single item; S = 0 -> NaN IRFs)."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, _ = make_responses(n, m, seed=n + m, snap_theta=False)
    codes = dict(yea=[1], nay=[-1], missing=[None])
    with np.errstate(all="ignore"):
        res = gpirtMCMC(y, S, B, vote_codes=codes, rng="reference", rstream=RStream(714))
        r = oracle.RStream(714)
        ref = oracle.gpirt_mcmc(r, y, r.rnorm(n), S, B)
    assert res["theta"].shape == (S + 1, n) and res["f"].shape == (n, m, S + 1)
    assert np.array_equal(res["theta"], ref["theta"])
    assert np.abs(res["f"] - ref["f"]).max() <= 1e-9
    if S == 0:
        assert np.isnan(res["IRFs"]).all() and np.isnan(ref["IRFs"]).all()        # quirk Q7
    else:
        assert np.abs(res["IRFs"] - ref["IRFs"]).max() <= 1e-9


def test_mcmc_reference_rng_with_zero_step_and_missing_column(handle, oracle):
    """rnorm(mu, 0) consumes nothing (src/draw-beta.cpp:22 through R::rnorm) and an all-but-two-NA item."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m = 70, 5
    y, _ = make_responses(n, m, seed=2, snap_theta=False)
    y[2:, 3] = np.nan
    y[0, 3], y[1, 3] = 1.0, -1.0
    st = np.full((2, m), 0.1)
    st[1, 2] = 0.0
    codes = dict(yea=[1], nay=[-1], missing=[None])
    rs = RStream(99)
    res = gpirtMCMC(y, 2, 1, vote_codes=codes, beta_proposal_sds=st, rng="reference", rstream=rs)
    r = oracle.RStream(99)
    ref = oracle.gpirt_mcmc(r, y, r.rnorm(n), 2, 1, step=st)
    _check(res, ref)
    mt, mti = rs.state()
    mt_ref, mti_ref = r.mt_state()
    assert mti == mti_ref and np.array_equal(mt, mt_ref)


def test_theta_degenerate_without_stabilisation_is_reported(handle):
    """With ~2000 items exp() underflows in the reference's draw_theta (0/0, out-of-bounds read):
    the unstabilised device path reports it, the stabilised one draws normally."""
    import torch
    from gpirt_amd.ops import to_device
    from gpirt_amd.synthetic import make_responses
    n, m = 64, 2500
    y, th = make_responses(n, m, seed=1)
    ts = -5.0 + 0.01 * np.arange(1001)
    fstar = 2.0 * ts[:, None] * np.ones((1, m))
    out, deg = handle.draw_theta(to_device(y), to_device(fstar), 3, 1, stabilise=False)
    assert deg > 0 and int(torch.isnan(out).sum().item()) == deg
    out, deg = handle.draw_theta(to_device(y), to_device(fstar), 3, 1, stabilise=True)
    assert deg == 0 and not torch.isnan(out).any()


def test_mixed_precision_kernel_build_option(handle):
    """Config C5: K(theta,theta) built with single-precision exp(), factored in fp64.  S is perturbed by
    ~6e-8 relative, so the chain is only statistically equivalent; the factorisation must still succeed
    (jitter 1e-3 >> perturbation) and L must stay within ~1e-4 of the fp64 build."""
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 1024, 8
    y, th0 = make_responses(n, m, seed=12)
    a = Sampler(handle, y, th0, rng="item", seed=3)
    b = Sampler(handle, y, th0, rng="item", seed=3, kernel_fp32=True)
    a.init(); b.init(); a.check(); b.check()
    La, Lb = a.get("L"), b.get("L")
    d = np.abs(La - Lb).max()
    assert 0 < d < 5e-3 and np.isfinite(Lb).all()
    b.step(); b.check()
    a.close(); b.close()


def test_long_run_item_rng_stays_on_the_oracle_trajectory(handle, oracle):
    """60 iterations: every discrete decision (ESS accepts, grid picks, MH accepts) must agree with the oracle,
    otherwise the two chains diverge visibly; posterior means agree to 1e-8 relative (north-star tolerance)."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.synthetic import make_responses
    n, m, S, B = 128, 12, 40, 20
    y, th0 = make_responses(n, m, seed=44)
    codes = dict(yea=[1], nay=[-1], missing=[None])
    res = gpirtMCMC(y, S, B, vote_codes=codes, theta_init=th0, rng="item", seed=2024, theta_stabilise=True)
    ref = oracle.gpirt_mcmc(oracle.ItemStream(2024), y, th0, S, B, theta_stabilise=True)
    assert np.array_equal(res["theta"], ref["theta"])
    assert np.abs(res["f"] - ref["f"]).max() <= 1e-8
    pm_gpu, pm_ref = res["f"][:, :, 1:].mean(axis=2), ref["f"][:, :, 1:].mean(axis=2)
    assert np.abs(pm_gpu - pm_ref).max() <= 1e-8 * max(1.0, np.abs(pm_ref).max())
    assert np.abs(res["IRFs"] - ref["IRFs"]).max() <= 1e-8


def test_sampler_recovers_the_latent_trait(handle):
    """Statistical sanity of the whole device path (no oracle involved): on 2PL data the posterior mean of
    theta must track the generating theta up to the model's reflection."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.synthetic import make_responses
    n, m = 512, 64
    y, th0, truth = make_responses(n, m, seed=77, return_truth=True)
    res = gpirtMCMC(y, 60, 40, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), theta_init=th0, rng="item",
                    seed=5, theta_stabilise=True, fstar_fused=True)
    assert np.isfinite(res["theta"]).all() and np.isfinite(res["f"]).all() and np.isfinite(res["IRFs"]).all()
    post = res["theta"][1:].mean(axis=0)
    r = np.corrcoef(post, truth)[0, 1]
    assert abs(r) > 0.9, r
    irf = res["IRFs"]
    assert irf.min() >= 0.0 and irf.max() <= 1.0


def test_config_c2_reference_rng_draw_for_draw(handle, oracle):
    """BASELINE config C2 (N = 1024 x m = 256): one full iteration under R's stream, every stored draw
    against the oracle (~10 s of CPU for the oracle's draw_theta)."""
    from gpirt_amd import gpirtMCMC
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m = 1024, 256
    y, _ = make_responses(n, m, seed=20241, snap_theta=False)
    rs = RStream(1234)
    res = gpirtMCMC(y, 1, 0, vote_codes=dict(yea=[1], nay=[-1], missing=[None]), rng="reference", rstream=rs)
    r = oracle.RStream(1234)
    ref = oracle.gpirt_mcmc(r, y, r.rnorm(n), 1, 0)
    _check(res, ref)
    mt, mti = rs.state()
    mt_ref, mti_ref = r.mt_state()
    assert mti == mti_ref and np.array_equal(mt, mt_ref)


def test_two_runs_are_bit_identical(handle):
    """The Cholesky panels hand blocks between work-groups through progress counters (panel.hip).  A lost
    hand-off or a stale read would make a run irreproducible long before it made it visibly wrong: two fresh
    samplers on the same inputs must agree to the last bit after every iteration (n spans 24 row blocks, three
    512-column sub-panels and a ragged tail; the trsm goes through the block-inverse leaves)."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m, iters = 1500, 300, 12
    y, th0 = make_responses(n, m, seed=7)

    def run():
        s = Sampler(handle, y, th0, rng="item", seed=11, theta_stabilise=True, fstar_fused=True)
        s.init()
        out = []
        for _ in range(iters):
            s.step()
            s.check()
            out.append([np.array(s.get(k)) for k in ("theta", "f", "beta", "L")])
        return out

    a, b = run(), run()
    for it, (sa, sb) in enumerate(zip(a, b)):
        for xa, xb in zip(sa, sb):
            assert np.isfinite(xa).all()
            assert np.array_equal(xa, xb), f"iteration {it} differs between two identical runs"


@pytest.mark.parametrize("n,m,rank", [(1339, 40, 64), (2048, 24, 48)])
def test_lowrank_kstar_equals_full_solve(handle, n, m, rank):
    """draw_fstar with K(theta, theta*) = K(theta, c) V^T (r Chebyshev nodes, two solves with r right-hand sides,
    forward and transposed, through the 512-block inverses) against the solve of all 1001 + m columns: same
    theta, f, L and RNG keys, so f* must agree to rounding (1e-9 is the tolerance of every f* comparison)."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=3)
    out = []
    for r in (0, rank):
        s = Sampler(handle, y, th0, rng="item", seed=5, theta_stabilise=True, fstar_fused=True, kstar_rank=r)
        s.init()
        for _ in range(2):
            s.step()
        s.check()
        out.append(np.array(s.get("fstar")))
        s.close()
    assert np.isfinite(out[0]).all() and np.isfinite(out[1]).all()
    assert np.abs(out[0] - out[1]).max() <= 1e-9


@pytest.mark.parametrize("form", ["lowrank", "fused", "double_solve"])
def test_bordered_factorisation_equals_explicit_solve(handle, form):
    """L^-1 K(theta, c) (rank-64 form) and L^-1 k* (the 1001 grid columns; fused and as-written forms) come out of the
    factorisation as extra rows below S (potrf.hip, extra_rows).  With GPIRT_BORDERED=2 the same sampler solves for them
    explicitly (src/draw-fstar.cpp:19 as a trsm): same theta, f, L and RNG keys, so f*, theta and f must agree to
    rounding over whole iterations."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 2048, 48
    y, th0 = make_responses(n, m, seed=21)
    kw = dict(lowrank=dict(fstar_fused=True, kstar_rank=64), fused=dict(fstar_fused=True), double_solve=dict())[form]
    out = []
    for bordered in (1, 2):
        with handle.config("GPIRT_BORDERED", bordered):      # (the layout is chosen when the sampler is created)
            s = Sampler(handle, y, th0, rng="item", seed=9, theta_stabilise=True, **kw)
        s.init()
        for _ in range(3):
            s.step()
        s.check()
        out.append({k: np.array(s.get(k)) for k in ("fstar", "theta", "f", "L")})
        s.close()
    a, b = out
    assert np.array_equal(a["theta"], b["theta"]) and np.array_equal(a["L"], b["L"]) and np.array_equal(a["f"], b["f"])
    assert np.abs(a["fstar"] - b["fstar"]).max() <= 1e-9 * max(1.0, np.abs(b["fstar"]).max())


@pytest.mark.parametrize("form", ["lowrank", "fused"])
def test_copy_state_and_set_theta_rebuild_the_border_rows(handle, form):
    """gpirt_sampler_copy_state / gpirt_sampler_set("theta") leave the rows below L stale; the next draw_fstar must
    rebuild them by the explicit solve: a sampler that received another one's state draws the same f* (same RNG keys)."""
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 1024, 24
    y, th0 = make_responses(n, m, seed=22)
    kw = dict(lowrank=dict(fstar_fused=True, kstar_rank=64), fused=dict(fstar_fused=True))[form]
    a = Sampler(handle, y, th0, rng="item", seed=4, theta_stabilise=True, **kw)
    b = Sampler(handle, y, th0[::-1].copy(), rng="item", seed=4, theta_stabilise=True, **kw)     # another chain
    a.init(); b.init()
    for _ in range(2):
        a.step()
    b.step()
    b.copy_state_from(a)
    assert b.iteration == a.iteration
    a.draw_fstar(); b.draw_fstar()
    a.check(); b.check()
    fa, fb = a.get("fstar"), b.get("fstar")
    assert np.abs(fa - fb).max() <= 1e-9 * max(1.0, np.abs(fa).max())
    # set("theta") alone: the rows follow theta at the next draw (compared with a sampler built on that theta and L)
    th = a.get("theta")
    b.set("theta", th)
    b.draw_fstar(); b.check()
    assert np.abs(b.get("fstar") - fa).max() <= 1e-9 * max(1.0, np.abs(fa).max())
    a.close(); b.close()


@pytest.mark.parametrize("n,kw", [(1024, dict(fstar_fused=False)), (1024, dict(fstar_fused=True)),
                                  (320, dict(fstar_fused=True, kstar_rank=64))])
def test_factor_set_from_outside_rebuilds_the_bordered_rows(handle, n, kw):
    """A C-API host that moves only the n x n factor (gpirt_sampler_set("L") / a broadcast of n x n) and then calls
    gpirt_sampler_skip_factor must NOT see the rows below L of the previous (theta, L) re-validated: draw_fstar has to
    rebuild them by the explicit solve.  (Round-2 advisor finding: skip_factor forced rows_valid.)"""
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    m = 6
    y, th0 = make_responses(n, m, seed=31)
    ref = Sampler(handle, y, th0, rng="item", seed=9, **kw)
    ref.init(); ref.step(); ref.check()                 # theta_1, L_1 = chol(K(theta_1)), rows valid
    tst = Sampler(handle, y, th0, rng="item", seed=9, **kw)
    tst.init()                                          # its rows belong to (theta_0, L_0)
    for name in ("theta", "f", "beta", "mu", "mu_star"):
        tst.set(name, ref.get(name))
    tst.set("L", ref.get("L"))                          # the n x n factor only
    tst.skip_factor()                                   # closes iteration 1 without factoring
    assert tst.iteration == ref.iteration == 1
    ref.draw_fstar(); tst.draw_fstar()
    ref.check(); tst.check()
    fs_ref, fs = ref.get("fstar"), tst.get("fstar")
    assert np.isfinite(fs).all()
    assert np.abs(fs - fs_ref).max() <= 1e-9 * max(1.0, np.abs(fs_ref).max())
    assert np.abs(tst.get("s") - ref.get("s")).max() <= 1e-9
    ref.close(); tst.close()


def test_null_options_need_an_r_stream(handle):
    """include/gpirt_hip.h: the default options are the reference's contract (R-stream replay), so a NULL options
    pointer without an R stream state is an argument error with a message that says so."""
    import ctypes as C
    from gpirt_amd import _lib
    lib = _lib.load()
    dp = C.POINTER(C.c_double)
    y = np.asfortranarray(np.where(np.arange(32).reshape(8, 4) % 3 == 0, 1.0, -1.0))
    th = np.zeros(8)
    p = np.asfortranarray(np.full((2, 4), 0.5))
    s = C.c_void_p()
    rc = lib.gpirt_sampler_create(C.byref(s), handle.ptr, y.ctypes.data_as(dp), 8, 4, th.ctypes.data_as(dp),
                                  p.ctypes.data_as(dp), p.ctypes.data_as(dp), p.ctypes.data_as(dp), None, None)
    assert rc == _lib.E_ARG and "R stream" in _lib.last_error()
    out = [np.zeros(sh, order="F") for sh in ((2, 8), (2, 4, 2), (8, 4, 2), (1001, 4))]
    rc = lib.gpirt_mcmc(y.ctypes.data_as(dp), 8, 4, th.ctypes.data_as(dp), 1, 0, p.ctypes.data_as(dp),
                        p.ctypes.data_as(dp), p.ctypes.data_as(dp), None, None, _lib.TICK_FN(0), None,
                        *[o.ctypes.data_as(dp) for o in out])
    assert rc == _lib.E_ARG and "R stream" in _lib.last_error()


def test_block_inverses_built_behind_the_factorisation_change_nothing(handle):
    """Round 3: the inverses of L's diagonal blocks that the low-rank draw_fstar's transposed solve applies are built by
    ranges -- every outer panel but the last while the last one is still being factored (sampler.hip do_factor), the rest
    behind the factorisation (fstar_prep).  A range only reads the blocks it inverts, so the result must be BIT-IDENTICAL to
    the one-range build (GPIRT_EARLY_INV=2), iteration after iteration.  (The ranged build was dead, untested code in round
    2 -- advisor finding.)"""
    import os
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 24
    y, th0 = make_responses(n, m, seed=41)
    kw = dict(rng="item", seed=6, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
    a = Sampler(handle, y, th0, **kw)
    a.init()
    for _ in range(3):
        a.step()
    a.check()
    with handle.config("GPIRT_EARLY_INV", 2):
        b = Sampler(handle, y, th0, **kw)
        b.init()
        for _ in range(3):
            b.step()
        b.check()
    for name in ("fstar", "theta", "f", "beta"):
        assert np.array_equal(a.get(name), b.get(name)), name
    a.close(); b.close()


def test_side_chain_beside_the_product_changes_nothing(handle):
    """The factor-only part of the rank-64 draw_fstar (last range of block inverses, 64-column transposed solve) starts
    beside nu = L z, with the 256-register form of the inverse leaf so that it fits on a CU next to the product's
    work-groups (sampler.hip do_draw_f, trsm.hip slim_leaf).  Same arithmetic, other streams and register budget: every
    draw must be BIT-IDENTICAL to the run that keeps it behind the product (GPIRT_PREP_EARLY=2)."""
    import os
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 160            # (m > 128: below that the side chain always started early)
    y, th0 = make_responses(n, m, seed=47)
    kw = dict(rng="item", seed=9, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
    outs = []
    for mode in (1, 2):
        with handle.config("GPIRT_PREP_EARLY", mode):
            s = Sampler(handle, y, th0, **kw)
            s.init()
            for _ in range(3):
                s.step()
            s.check()
            outs.append({name: s.get(name) for name in ("fstar", "theta", "f", "beta")})
            s.close()
    for name in outs[0]:
        assert np.array_equal(outs[0][name], outs[1][name]), name


def test_r_stream_generator_runs_ahead_and_is_handed_back_exactly(handle, oracle):
    """Under the R-stream contract the host generator runs AHEAD of the chain between iterations (the next window's uniforms
    are generated while the device works; sampler.hip, stream_end).  Whoever looks at the RStream -- its state, a draw from
    it -- must find it at exactly the consumed position, and the chain must go on from there as if nobody had looked:
    A steps 4 times; B steps twice, reads the state, steps, draws 3 uniforms (which the chain then does not get), steps;
    C is the oracle's whole call for the first three iterations' consumption."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m = 300, 12
    y, th0 = make_responses(n, m, seed=5)
    ra = RStream(77)
    a = Sampler(handle, y, th0, rng="reference", rstream=ra)
    a.init()
    for _ in range(3):
        a.step()
    a.check()
    fa3, tha3 = a.get("f"), a.get("theta")
    state3 = ra.state()                                   # (hands the generator back; A re-attaches at its next step)
    a.step(); a.check()
    fa4 = a.get("f")
    rb = RStream(77)
    b = Sampler(handle, y, th0, rng="reference", rstream=rb)
    b.init()
    b.step(); b.step()
    rb.state()
    b.step(); b.check()
    assert np.array_equal(b.get("f"), fa3) and np.array_equal(b.get("theta"), tha3)
    sb = rb.state()
    assert sb[1] == state3[1] and np.array_equal(sb[0], state3[0])
    # the oracle's generator after init + 3 iterations sits at the same place
    r = oracle.RStream(77)
    oracle.gpirt_mcmc(r, y, th0, 3, 0)
    mt_ref, mti_ref = r.mt_state()
    assert sb[1] == mti_ref and np.array_equal(sb[0], mt_ref)
    # a draw taken from the stream in between belongs to the caller, not to the chain
    taken = rb.runif(3)
    assert np.array_equal(taken, r.runif(3))
    b.step(); b.check()
    assert not np.array_equal(b.get("f"), fa4)            # (the chain went on with the uniforms BEHIND the three taken)
    a.close(); b.close()


@pytest.mark.parametrize("n,m,limit", [(100, 17, 2), (512, 9, 1), (1030, 6, 3), (2048, 5, 2)])
def test_r_stream_draw_f_redoes_an_item_whose_candidates_ran_out(handle, oracle, n, m, limit):
    """The replay's draw_f takes item j's nu = L z from 32 candidates computed beside item j - 1's slice loop, one per possible
    rejection count of that loop (rng_ess.hip); a longer loop leaves item j without a candidate and the host redoes it the
    plain way.  With the limit lowered to 1-3 that happens every few items: every draw must still be the oracle's."""
    from gpirt_amd import Sampler, _lib
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    lib = _lib.load()
    y, th0 = make_responses(n, m, seed=n + m)
    _lib.check(lib.gpirt_debug_rs_cand_limit(handle._h, limit))
    try:
        rs = RStream(4242)
        s = Sampler(handle, y, th0, rng="reference", rstream=rs)
        s.init()
        for _ in range(2):
            s.step()
        s.check()
        got = {k: s.get(k) for k in ("theta", "f", "beta")}
        ks = s.get("ess_k")
        state = rs.state()
        s.close()
    finally:
        _lib.check(lib.gpirt_debug_rs_cand_limit(handle._h, 0))
    assert (ks >= limit).any()                            # (the path under test did run)
    r = oracle.RStream(4242)
    ref = oracle.gpirt_mcmc(r, y, th0, 2, 0)
    assert np.array_equal(got["theta"], ref["theta"][2])
    assert np.abs(got["f"] - ref["f"][:, :, 2]).max() <= 1e-9
    assert np.abs(got["beta"] - ref["beta"][:, :, 2]).max() <= 1e-9
    mt_ref, mti_ref = r.mt_state()
    assert state[1] == mti_ref and np.array_equal(state[0], mt_ref)


@pytest.mark.parametrize("n,m,every", [(100, 17, 1), (512, 23, 3), (1030, 14, 4), (2048, 40, 7)])
def test_r_stream_predicted_replay_survives_a_wrong_predictor(handle, oracle, n, m, every):
    """The replay's draw_f predicts every item's start in R's stream on a single-precision copy of L and then computes all
    items exactly at those starts, committing in order what the prediction did not break (rs_predict.hip).  With
    gpirt_debug_rs_mispredict the predictor is off by one at every `every`-th item (every = 1: at EVERY item, so each round
    commits one item): the verification must find each of them, keep the item whose own start was right, correct the next
    start and resume -- every draw the oracle's, the generator handed back at the oracle's position."""
    from gpirt_amd import Sampler, _lib
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    lib = _lib.load()
    y, th0 = make_responses(n, m, seed=n + m)
    _lib.check(lib.gpirt_debug_rs_mispredict(handle._h, every))
    try:
        rs = RStream(4242)
        s = Sampler(handle, y, th0, rng="reference", rstream=rs)
        s.init()
        for _ in range(2):
            s.step()
        s.check()
        got = {k: s.get(k) for k in ("theta", "f", "beta")}
        ks = s.get("ess_k")
        stats = s.get("rs_stats")
        state = rs.state()
        s.close()
    finally:
        _lib.check(lib.gpirt_debug_rs_mispredict(handle._h, 0))
    assert stats[1] >= 2 * (m // every) - 2               # (mispredictions found by the verification: the path under test ran)
    assert stats[2] == 0                                  # (... and the predictor never stalled into the one-phase replay)
    r = oracle.RStream(4242)
    ref = oracle.gpirt_mcmc(r, y, th0, 2, 0)
    assert np.array_equal(got["theta"], ref["theta"][2])
    assert np.abs(got["f"] - ref["f"][:, :, 2]).max() <= 1e-9
    assert np.abs(got["beta"] - ref["beta"][:, :, 2]).max() <= 1e-9
    mt_ref, mti_ref = r.mt_state()
    assert state[1] == mti_ref and np.array_equal(state[0], mt_ref)


@pytest.mark.parametrize("n,m", [(64, 1), (130, 2), (257, 9), (1024, 33), (3000, 12), (8200, 4)])
def test_r_stream_predicted_replay_agrees_with_the_one_phase_replay(handle, n, m):
    """GPIRT_RS_PREDICT=2 runs every pass over L in fp64 (the one-phase replay of rng_ess.hip); the default predicts in single
    precision and verifies in fp64.  Same rejection counts, same theta, same stream position; f to rounding (nu = L z is summed
    in another order: one triangular product instead of 512-column parts)."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=7 * n + m)
    outs = []
    for mode in (1, 2):
        with handle.config("GPIRT_RS_PREDICT", mode):
            rs = RStream(31)
            s = Sampler(handle, y, th0, rng="reference", rstream=rs)
            s.init()
            for _ in range(3):
                s.step()
            s.check()
            outs.append((s.get("f"), s.get("ess_k"), s.get("theta"), s.get("beta"), rs.state(), s.get("rs_stats")))
            s.close()
    (f1, k1, t1, b1, st1, q1), (f2, k2, t2, b2, st2, q2) = outs
    assert np.array_equal(k1, k2) and np.array_equal(t1, t2)
    assert st1[1] == st2[1] and np.array_equal(st1[0], st2[0])
    assert np.abs(f1 - f2).max() <= 1e-10 and np.abs(b1 - b2).max() <= 1e-10
    assert q1[1] == 0 and q1[2] == 0                      # (no misprediction, no stall on these chains: the predictor earns its keep)


@pytest.mark.parametrize("n,m,every", [(4096, 30, 0), (4500, 21, 0), (8192, 16, 0), (4500, 21, 5)])
def test_r_stream_structured_predictor_agrees_with_the_dense_predictor(handle, n, m, every):
    """From 4096 respondents on the predictor's pass does not read L: the blocks of the factor below the 512-column diagonal
    parts are applied as V C -- the Lagrange basis of 64 Chebyshev nodes at theta times coefficients built from theta alone
    (rs_lr.hip); GPIRT_RS_LR=2 keeps the dense single-precision pass.  Both only PREDICT; what is returned is verified exactly:
    same counts, theta, stream position and f as the dense predictor, and the structured predictor is right (no misprediction
    over these chains, n a multiple of 512 or not).  every > 0: the predictor is made wrong on purpose at every fifth item --
    the first round of a draw is structured, the dense pass takes the rest of the draw over, three such draws retire the
    structured form for the sampler; the draws are unchanged."""
    from gpirt_amd import Sampler, _lib
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    lib = _lib.load()
    y, th0 = make_responses(n, m, seed=5 * n + m)
    outs = []
    for mode in (1, 2):
        with handle.config("GPIRT_RS_LR", mode):
            if every and mode == 1:
                _lib.check(lib.gpirt_debug_rs_mispredict(handle._h, every))
            try:
                rs = RStream(77)
                s = Sampler(handle, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
                s.init()
                for _ in range(5):
                    s.step()
                s.check()
                outs.append((s.get("f"), s.get("ess_k"), s.get("theta"), rs.state(), s.get("rs_stats")))
                s.close()
            finally:
                _lib.check(lib.gpirt_debug_rs_mispredict(handle._h, 0))
    (f1, k1, t1, st1, q1), (f2, k2, t2, st2, q2) = outs
    assert np.array_equal(k1, k2) and np.array_equal(t1, t2)
    assert st1[1] == st2[1] and np.array_equal(st1[0], st2[0])
    assert np.abs(f1 - f2).max() <= 1e-10
    assert q1[2] == 0 and q2[2] == 0
    if every == 0:
        assert q1[1] == 0 and q2[1] == 0, (q1, q2)
    else:
        assert q1[1] >= 5 * (m // every) - 5, q1


@pytest.mark.parametrize("kind", ["sorted", "constant", "wide", "two_values"])
def test_r_stream_structured_predictor_with_awkward_theta(handle, kind):
    """The structured pass is built from theta alone (rs_lr.hip).  Starting values that stress the construction -- respondents
    sorted by theta (every 64-row block sees a sliver of the axis), all respondents at one point or at two (a kernel matrix of
    rank one or two), values beyond the nodes' interval [-5, 5] (clamped for the basis: those rows are predicted badly) -- may
    cost mispredictions but never a wrong draw, a stall into the one-phase replay or a flagged coefficient block that goes
    unnoticed: counts, theta, stream position and f equal the dense predictor's."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 12
    y, th0 = make_responses(n, m, seed=99, snap_theta=False)
    if kind == "sorted":
        th0 = np.sort(th0)
    elif kind == "constant":
        th0 = np.full(n, 0.37)
    elif kind == "two_values":
        th0 = np.where(np.arange(n) % 3 == 0, -1.25, 2.5).astype(np.float64)
    else:
        th0 = 2.4 * th0                                    # a few hundred respondents beyond +-5
    outs = []
    for mode in (1, 2):
        with handle.config("GPIRT_RS_LR", mode):
            rs = RStream(5)
            s = Sampler(handle, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
            s.init()
            for _ in range(2):
                s.step()
            s.check()
            outs.append((s.get("f"), s.get("ess_k"), s.get("theta"), rs.state(), s.get("rs_stats")))
            s.close()
    (f1, k1, t1, st1, q1), (f2, k2, t2, st2, q2) = outs
    assert np.array_equal(k1, k2) and np.array_equal(t1, t2)
    assert st1[1] == st2[1] and np.array_equal(st1[0], st2[0])
    assert np.abs(f1 - f2).max() <= 1e-10
    assert q1[2] == 0 and q2[2] == 0, (q1, q2)
    print(f"[structured predictor, theta_init {kind}] mispredictions {int(q1[1])} (dense predictor: {int(q2[1])})")


def test_r_stream_predicted_replay_long_slice_loops(handle):
    """Behind the burn-in of a chain with 8192 respondents the slice loops lengthen (mean count ~6, some above 16): a loop that
    rejects all sixteen points of a pass goes on in the next pass (round + 1), the candidate starts of the items behind it begin
    at what the lost rounds consumed, and slot 2's window of counts is centred on twice the last draw's mean (rs_predict.hip).
    Sixty iterations both ways: same counts, theta and stream position as the one-phase replay; the predictor is (almost)
    never wrong and never stalls, and needs fewer than 0.42 passes per item."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m, its = 8192, 48, 60
    y, th0 = make_responses(n, m, seed=20240)
    outs = []
    for mode in (1, 2):
        with handle.config("GPIRT_RS_PREDICT", mode):
            rs = RStream(20240)
            s = Sampler(handle, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
            s.init()
            kmax, ks = 0, []
            for i in range(its):
                s.step()
                if i >= its - 20:
                    s.check()
                    k = s.get("ess_k")
                    kmax = max(kmax, int(k.max())); ks.append(k.copy())
            s.check()
            outs.append((np.array(ks), s.get("theta"), rs.state(), s.get("rs_stats"), kmax))
            s.close()
    (k1, t1, st1, q1, kmax1), (k2, t2, st2, q2, _) = outs
    assert np.array_equal(k1, k2) and np.array_equal(t1, t2)
    assert st1[1] == st2[1] and np.array_equal(st1[0], st2[0])
    assert kmax1 >= 16, kmax1                             # (the regime this test is about was reached)
    assert q1[1] <= 2 and q1[2] == 0, q1
    assert q1[3] < 0.42 * its * m, q1


@pytest.mark.parametrize("n,m", [(97, 11), (1025, 7), (640, 40)])
def test_r_stream_draw_f_three_items_per_pass_odd_shapes(handle, oracle, n, m):
    """The replay's draw_f resolves up to three items per pass over L (rng_ess.hip): odd n (rows and columns past the last
    32-row group / 512-column part are padding), m not a multiple of three (the last pass has one or two slots), and enough
    items that some pass ends early on its own (a count beyond a slot's candidates) -- every draw must be the oracle's."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=3 * n + m)
    rs = RStream(99)
    s = Sampler(handle, y, th0, rng="reference", rstream=rs)
    s.init()
    for _ in range(3):
        s.step()
    s.check()
    got = {k: s.get(k) for k in ("theta", "f", "beta")}
    state = rs.state()
    s.close()
    r = oracle.RStream(99)
    ref = oracle.gpirt_mcmc(r, y, th0, 3, 0)
    assert np.array_equal(got["theta"], ref["theta"][3])
    assert np.abs(got["f"] - ref["f"][:, :, 3]).max() <= 1e-9
    assert np.abs(got["beta"] - ref["beta"][:, :, 3]).max() <= 1e-9
    mt_ref, mti_ref = r.mt_state()
    assert state[1] == mti_ref and np.array_equal(state[0], mt_ref)


@pytest.mark.parametrize("n,m", [(9216, 5), (16448, 4), (33024, 3)])
def test_r_stream_draw_f_at_large_n_against_the_oracle_stage(handle, oracle, n, m):
    """The replay's slice kernel keeps 1 / 2 / 4 / 8 rows per thread (n <= 8192 / 16384 / 32768 / 65536, rng_ess.hip); the
    whole-chain tests only reach the first.  Here draw_f of ONE iteration at n = 9216, 16448 (257 row blocks of 64) and
    33024 is compared with the oracle's draw_f (src/draw-f.cpp:64-73) run on the device's own state -- f, L, mu after init and
    the generator at the consumed position: rejection counts exact, f to 1e-9."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, m, seed=n + m)
    rs = RStream(1234)
    s = Sampler(handle, y, th0, rng="reference", rstream=rs, theta_stabilise=True)
    s.init(); s.check()
    mt, mti = rs.state()                                  # the generator where the chain stands
    f0, L0, mu0 = s.get("f"), s.get("L"), s.get("mu")
    s.step(); s.check()                                   # draw_f is the iteration's first stage; nothing later touches f
    f1, k_dev = s.get("f"), s.get("ess_k")
    s.close()
    r = oracle.RStream(0)
    for q in range(624):
        r.s.mt[q] = int(mt[q])
    r.s.mti = int(mti)
    f_ref, k_ref = oracle.draw_f(r, f0, y, np.tril(L0), mu0)
    assert np.array_equal(k_dev, k_ref), (k_dev, k_ref)
    assert np.abs(f1 - f_ref).max() <= 1e-9 * max(1.0, np.abs(f_ref).max())


def test_r_stream_destroyed_under_an_attached_sampler_is_an_error_not_a_dangling_pointer(handle):
    """gpirt_rstream_destroy while a sampler created on the stream is alive but NOT running ahead of it (right after init,
    or after another sampler took the generator over): the sampler must learn that its stream is gone -- the next step fails
    with a named error instead of reading freed memory (round-4 advisor finding)."""
    from gpirt_amd import Sampler, _lib
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    lib = _lib.load()
    y, th0 = make_responses(128, 5, seed=4)
    rs = RStream(5)
    a = Sampler(handle, y, th0, rng="reference", rstream=rs)
    b = Sampler(handle, y, th0, rng="reference", rstream=rs)
    a.init(); b.init()
    a.step(); a.check()                                   # a runs ahead
    b.step(); b.check()                                   # b took the generator over: a is attached, not the owner
    lib.gpirt_rstream_destroy(rs._r); rs._r = None
    for s in (a, b):
        with pytest.raises(_lib.GpirtError) as e:
            s.step()
        assert "has been destroyed" in str(e.value)
    a.close(); b.close()


def test_two_samplers_alternating_on_one_r_stream(handle, oracle):
    """Two chains drawing from ONE R stream in turns (each step continues where the other's left the generator): each
    sampler runs ahead of its own chain between ITS steps, so every step of the other must first get the generator back
    at the consumed position.  Reference: the same interleaving done with a hand-over of the state after every step."""
    from gpirt_amd import Sampler
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    n, m = 200, 8
    ya, tha = make_responses(n, m, seed=11)
    yb, thb = make_responses(n, m, seed=12)

    def run(peek):
        rs = RStream(5)
        a = Sampler(handle, ya, tha, rng="reference", rstream=rs)
        b = Sampler(handle, yb, thb, rng="reference", rstream=rs)
        a.init(); b.init()
        for _ in range(3):
            a.step()
            if peek: rs.state()
            b.step()
            if peek: rs.state()
        a.check(); b.check()
        out = (a.get("f"), a.get("theta"), b.get("f"), b.get("theta"), rs.state())
        a.close(); b.close()
        return out

    x, y = run(False), run(True)
    for u, v in zip(x[:4], y[:4]):
        assert np.array_equal(u, v)
    assert x[4][1] == y[4][1] and np.array_equal(x[4][0], y[4][0])
