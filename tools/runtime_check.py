"""The dependency-driven factorisation (GPIRT_RUNTIME=2, runtime.hip) against the launch-ordered schedule on the SAME
theta: max |dL| over the factor and the rows of the bordered layout, residual, and the time of the factorisation alone.
    python tools/runtime_check.py [n ...] [--time]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses

sizes = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [3072]
do_time = "--time" in sys.argv
for n in sizes:
    y, th0 = make_responses(n, 8, seed=n)
    th0 = -5.0 + np.clip(np.rint((th0 + 5.0) / 0.01), 0, 1000) * 0.01          # grid-valued: the steady-state condition
    out = {}
    for mode in (1, 2):
        h = Handle()
        h.config_set("GPIRT_RUNTIME", mode)
        s = Sampler(h, y, th0, rng="item", seed=1, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
        t0 = time.perf_counter()
        s.init(); s.check()
        print(f"n = {n} runtime {mode}: init + check {time.perf_counter() - t0:.2f} s, guard fallbacks {h.guard_fallbacks}", flush=True)
        buf = s.device_tensor("L")
        ldl = buf.numel() // n
        Lfull = buf.reshape(n, ldl).T[: n + 64].clone()                       # (n + 64) x n: factor + bordered rows
        out[mode] = Lfull.cpu().numpy()
        if do_time:
            for _ in range(3):
                s.factor()
            s.check()
            ts = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    s.factor()
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10 * 1e3)
            s.check()
            print(f"n = {n} runtime {mode}: factor min {min(ts):.3f} ms  mean {sum(ts) / len(ts):.3f} ms", flush=True)
        s.close(); h.close()
    a, b = np.tril(out[1][:n]), np.tril(out[2][:n])
    d = np.abs(a - b)
    print(f"n = {n}: max|dL| {d.max():.3e} (rows differing: {int((d.max(axis=1) > 0).sum())} of {n}; first {int(np.argmax(d.max(axis=1) > 0))}), "
          f"bordered rows {np.abs(out[1][n:] - out[2][n:]).max():.3e}, finite {np.isfinite(b).all()}", flush=True)
