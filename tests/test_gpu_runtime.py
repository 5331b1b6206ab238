"""The dependency-driven factorisation (GPIRT_RUNTIME=2, gpirt_amd/csrc/runtime.hip: one persistent update kernel working
off tile tasks, sub-panel kernels on reserved compute units, rows below a sub-panel's window as X = A W^T) against the
launch-ordered schedule on the same theta: the rows every sub-panel kernel sweeps itself (its own outer panel and the next
two) receive the same products in the same order -- BIT-IDENTICAL; below, inverse-based instead of substitution -- equal to
rounding (<= 1e-12).  Opt-in: it measured slower, so this is a correctness test of the machinery:
task lists for 3, 5 and 8 outer panels, a ragged last panel, the bordered rows, repeated factorisations on one handle
(the counters run on across them), a whole sampler chain, and the hang-guard fallback underneath it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _factor(n, runtime, reps=1, kw=None):
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    y, th0 = make_responses(n, 8, seed=n)
    th0 = -5.0 + np.clip(np.rint((th0 + 5.0) / 0.01), 0, 1000) * 0.01
    h = Handle()
    h.config_set("GPIRT_RUNTIME", runtime)
    s = Sampler(h, y, th0, rng="item", seed=1, theta_stabilise=True, **(kw or dict(fstar_fused=True, kstar_rank=64)))
    s.init(); s.check()
    for _ in range(reps - 1):
        s.factor()
    s.check()
    buf = s.device_tensor("L")
    ldl = buf.numel() // n
    L = buf.reshape(n, ldl).T[: n + 64].clone().cpu().numpy()
    fb = h.guard_fallbacks
    s.close(); h.close()
    return L, fb


@pytest.mark.parametrize("n,reps", [(3072, 1), (4096, 3), (5120, 2), (8192, 2), (4480, 1)])
def test_runtime_factor_equals_launch_ordered_factor(n, reps):
    a, _ = _factor(n, 1)
    b, fb = _factor(n, 2, reps)
    assert fb == 0, "the dependency-driven factorisation fell back to the launch-per-step panel"
    assert np.isfinite(b).all()
    win = min(n, 2048)                                   # rows every sub-panel kernel sweeps itself from panel 0 on
    assert np.array_equal(np.tril(a[:win, :]), np.tril(b[:win, :]))
    assert np.abs(np.tril(a[:n]) - np.tril(b[:n])).max() <= 1e-12
    assert np.abs(a[n:] - b[n:]).max() <= 1e-11         # the rows of the bordered layout: (L^-1 K(theta, c))^T, entries up to ~30


def test_runtime_chain_equals_launch_ordered_chain():
    from gpirt_amd.ops import Handle
    from gpirt_amd.sampler import Sampler
    from gpirt_amd.synthetic import make_responses
    n, m = 4096, 40
    y, th0 = make_responses(n, m, seed=17)
    outs = []
    for mode in (1, 2):
        h = Handle()
        h.config_set("GPIRT_RUNTIME", mode)
        s = Sampler(h, y, th0, rng="item", seed=3, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
        s.init()
        for _ in range(3):
            s.step()
        s.check()
        outs.append({k: s.get(k) for k in ("theta", "f", "fstar", "beta")})
        assert h.guard_fallbacks == 0
        s.close(); h.close()
    a, b = outs
    assert np.array_equal(a["theta"], b["theta"])
    for k in ("f", "fstar", "beta"):
        assert np.abs(a[k] - b[k]).max() <= 1e-9 * max(1.0, np.abs(a[k]).max()), k
