"""Wall time of gpirt_mcmc with stored vs burn-in iterations at the metric size (host arrays, PCIe included)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd import gpirtMCMC
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
codes = dict(yea=[1], nay=[-1], missing=[None])
kw = dict(vote_codes=codes, theta_init=th0, rng="item", seed=3, theta_stabilise=True, fstar_fused=True)
gpirtMCMC(y, 1, 1, **kw)                                    # warm-up (library load, allocations)
for S, B in ((1, 11), (12, 0)):
    t0 = time.perf_counter(); r = gpirtMCMC(y, S, B, **kw); dt = time.perf_counter() - t0
    print(f"S={S:2d} B={B:2d}: {dt*1e3:8.1f} ms total, {dt*1e3/(S+B):6.1f} ms per iteration (incl. setup/upload)")
