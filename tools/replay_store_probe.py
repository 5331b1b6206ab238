"""What storing a draw costs under the default (R-stream) contract: whole gpirt_mcmc() calls at n x m with the same number
of iterations, all stored against one stored.   python tools/replay_store_probe.py [n = 8192] [m = 1024] [iters = 4]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd import gpirtMCMC
from gpirt_amd.ops import RStream
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
it = int(sys.argv[3]) if len(sys.argv) > 3 else 4
y, th0 = make_responses(n, m, seed=20240)
codes = dict(yea=[1], nay=[-1], missing=[None])
kw = dict(vote_codes=codes, theta_init=th0, rng="reference", theta_stabilise=True)
gpirtMCMC(y, 1, 0, rstream=RStream(1), **kw)
for S, B in ((1, it - 1), (it, 0), (1, it - 1), (it, 0)):
    t0 = time.perf_counter()
    gpirtMCMC(y, S, B, rstream=RStream(7), **kw)
    dt = time.perf_counter() - t0
    print(f"{n} x {m}: {S} stored + {B} burn-in iterations: {dt:.3f} s whole call", flush=True)
