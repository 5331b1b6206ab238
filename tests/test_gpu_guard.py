"""The hang-guard fallback (round-3 verdict, item 3) and the interleaving of samplers on one handle (item 2a).

A factorisation whose persistent sub-panel kernel gave up on a progress counter (flagsync.h: its work-groups were not
co-resident within the spin bound) used to end the whole gpirt_mcmc call.  Now it is repeated once, from the intact theta,
with the launch-per-step panel, and the chain goes on.  The expiry is staged through gpirt_debug_trip_guard (guard word
raised, eight columns of the result poisoned with NaN: no kernel spins), at the operator, the stage API and the whole-call
boundary; every time the disturbed chain must equal the undisturbed one -- theta exactly (grid values), f / f* / beta to
1e-10 (L differs by schedule only, <= 1e-12)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _theta_grid(n, seed):
    rng = np.random.default_rng(seed)
    k = np.clip(np.rint((rng.standard_normal(n) + 5.0) / 0.01), 0, 1000)
    return -5.0 + k * 0.01


@pytest.mark.parametrize("n", [2600, 4096])
def test_operator_factor_survives_a_guard_expiry(handle, n):
    from gpirt_amd.ops import to_device, to_host
    th = to_device(_theta_grid(n, n))
    L0 = to_host(handle.factor(th))
    before = handle.guard_fallbacks
    handle.debug_trip_guard(1)
    L1 = to_host(handle.factor(th))                   # trips, is rebuilt from theta and refactored on the fallback panel
    assert handle.guard_fallbacks == before + 1
    assert np.isfinite(L1).all()
    assert np.abs(L1 - L0).max() <= 1e-12
    L2 = to_host(handle.factor(th))                   # and the persistent kernel is back afterwards, bit for bit
    assert np.array_equal(L2, L0)
    assert handle.guard_fallbacks == before + 1


@pytest.mark.parametrize("form", ["lowrank", "fused"])
def test_sampler_check_repairs_a_guard_expiry_in_place(handle, form):
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 40
    y, th0 = make_responses(n, m, seed=61)
    kw = dict(rng="item", seed=3, theta_stabilise=True, fstar_fused=True)
    if form == "lowrank":
        kw["kstar_rank"] = 64
    outs = []
    for trip in (False, True):
        s = Sampler(handle, y, th0, **kw)
        s.init()
        before = handle.guard_fallbacks
        for it in range(4):
            if trip and it == 1:
                handle.debug_trip_guard(1)            # the factorisation that closes this iteration
            s.step()
            s.check()                                 # (finds the guard word, repairs, goes on)
        assert handle.guard_fallbacks == before + (1 if trip else 0)
        outs.append({k: s.get(k) for k in ("theta", "f", "fstar", "beta", "L")})
        s.close()
    a, b = outs
    assert np.array_equal(a["theta"], b["theta"])
    assert np.abs(np.tril(a["L"]) - np.tril(b["L"])).max() <= 1e-12
    for k in ("f", "fstar", "beta"):
        assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(1.0, np.abs(a[k]).max()), k


def test_a_guard_expiry_that_was_consumed_is_still_an_error(handle):
    """check() can only repair while nothing has read the unfinished factor; behind a draw_f it must say so."""
    from gpirt_amd import Sampler, _lib
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(2600, 8, seed=5)
    s = Sampler(handle, y, th0, rng="item", seed=3, theta_stabilise=True)
    s.init()
    handle.debug_trip_guard(1)
    s.step()
    s.draw_f()                                        # consumes the poisoned L
    with pytest.raises(_lib.GpirtError, match="hang guard"):
        s.check()
    s.close()


@pytest.mark.parametrize("rng,n,m,S,B,nth", [("item", 2600, 12, 3, 2, 3), ("item", 2600, 12, 2, 3, 5), ("reference", 2600, 3, 2, 1, 2)])
def test_gpirt_mcmc_rolls_back_and_continues(rng, n, m, S, B, nth):
    """The whole-call boundary: the nth factorisation of the call (1 = init's) ends in a staged expiry.  Item RNG: the host
    is up to two iterations ahead when the words come back -- the chain is rolled back to the last verified checkpoint,
    the lost iterations are repeated on the fallback panel, and every returned array equals the undisturbed call's."""
    from gpirt_amd import gpirtMCMC, _lib
    from gpirt_amd.ops import RStream
    from gpirt_amd.synthetic import make_responses
    lib = _lib.load()
    y, th0 = make_responses(n, m, seed=71)
    codes = dict(yea=[1], nay=[-1], missing=[None])
    kw = dict(vote_codes=codes, theta_init=th0, rng=rng, theta_stabilise=(rng == "item"))

    def run():
        if rng == "reference":
            return gpirtMCMC(y, S, B, rstream=RStream(1234), **kw)
        return gpirtMCMC(y, S, B, seed=11, **kw)

    ref = run()
    assert lib.gpirt_debug_last_mcmc_fallbacks() == 0
    _lib.check(lib.gpirt_debug_trip_guard(None, nth))
    got = run()
    assert lib.gpirt_debug_last_mcmc_fallbacks() == 1
    assert np.array_equal(got["theta"], ref["theta"])
    for k in ("beta", "f", "IRFs"):
        assert np.isfinite(got[k]).all(), k
        assert np.abs(got[k] - ref[k]).max() <= 1e-10 * max(1.0, np.abs(ref[k]).max()), k


def test_interleaved_samplers_on_one_handle_equal_samplers_alone(handle):
    """Round 3, gpurun_out/r3p: theta differed between two forms stepped alternately on one handle, on exactly the
    machinery that is now default -- work for the NEXT stage launched on the shared side handle behind the factorisation's
    last outer panel (early block inverses, the next draw_f's normals, draw_beta).  Two rank-64 samplers and one `fused`
    sampler, different seeds, interleaved step by step on one handle, must reproduce what each does alone."""
    from gpirt_amd import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 48
    y, th0 = make_responses(n, m, seed=83)
    specs = [dict(seed=5, fstar_fused=True, kstar_rank=64), dict(seed=6, fstar_fused=True, kstar_rank=64),
             dict(seed=5, fstar_fused=True)]
    names = ("theta", "f", "fstar", "beta")
    alone = []
    for kw in specs:
        s = Sampler(handle, y, th0, rng="item", theta_stabilise=True, **kw)
        s.init()
        for _ in range(3):
            s.step()
        s.check()
        alone.append({k: s.get(k) for k in names})
        s.close()
    S = [Sampler(handle, y, th0, rng="item", theta_stabilise=True, **kw) for kw in specs]
    for s in S:
        s.init()
    for _ in range(3):
        for s in S:                                   # no check() in between: nothing drains the streams
            s.step()
    for s, ref in zip(S, alone):
        s.check()
        for k in names:
            assert np.array_equal(s.get(k), ref[k]), k
    # teardown in the WRONG order on purpose: the handle's close() must take its samplers with it (the r3p abort at
    # interpreter exit was a sampler draining a freed handle)
    from gpirt_amd.ops import Handle
    h2 = Handle()
    s2 = Sampler(h2, y[:, :4], th0, rng="item", seed=1, theta_stabilise=True)
    s2.init()
    h2.close()
    s2.close()
