"""Fixed cost of a whole gpirt_mcmc() call (handle, sampler, pinned windows, init) against its per-iteration cost, both
contracts.   python tools/call_overhead_probe.py [n = 8192] [m = 1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd import gpirtMCMC
from gpirt_amd.ops import RStream
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
y, th0 = make_responses(n, m, seed=20240)
codes = dict(yea=[1], nay=[-1], missing=[None])
for rng in ("reference", "item"):
    kw = dict(vote_codes=codes, theta_init=th0, rng=rng, theta_stabilise=True)
    extra = (lambda: dict(rstream=RStream(7))) if rng == "reference" else (lambda: dict(seed=7))
    gpirtMCMC(y, 1, 0, **kw, **extra())
    ts = {}
    for B in (0, 8):
        t0 = time.perf_counter()
        gpirtMCMC(y, 1, B, **kw, **extra())
        ts[B] = time.perf_counter() - t0
    per = (ts[8] - ts[0]) / 8
    print(f"{n} x {m} rng={rng}: 1 iteration {ts[0]:.3f} s, 9 iterations {ts[8]:.3f} s -> {per * 1e3:.1f} ms per iteration, fixed {ts[0] - per:.3f} s", flush=True)
