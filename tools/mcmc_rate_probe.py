"""Iterations per second THROUGH THE DROP-IN ENTRY (gpirt_mcmc: checkpoints, flag read-backs, stored draws on the copy
stream) against the sampler's own step loop that bench.py times.   python tools/mcmc_rate_probe.py [n = 8192] [m = 1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd import gpirtMCMC, Sampler
from gpirt_amd.ops import Handle
from gpirt_amd.response_matrix import response_matrix
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
y, th0 = make_responses(n, m, seed=20240)
y = response_matrix(y, dict(yea=[1], nay=[-1], missing=[None]))      # (so that the calls below skip the recode)
form = dict(theta_stabilise=True, fstar_fused=True, kstar_rank=64)
kw = dict(theta_init=th0, rng="item", seed=7, **form)
gpirtMCMC(y, 1, 0, **kw)
res = {}
for S, B in ((1, 4), (1, 44), (41, 4)):
    t0 = time.perf_counter(); out = gpirtMCMC(y, S, B, **kw); res[(S, B)] = time.perf_counter() - t0
    del out                                               # (freeing 2.8 GB of touched pages is not the library's time)
burn = (res[(1, 44)] - res[(1, 4)]) / 40
stored = (res[(41, 4)] - res[(1, 4)]) / 40
print(f"gpirt_mcmc {n} x {m}: {burn * 1e3:.2f} ms per burn-in iteration ({1 / burn:.1f} it/s), {stored * 1e3:.2f} ms per stored iteration ({1 / stored:.1f} it/s)")
h = Handle(); s = Sampler(h, np.asarray(y), th0, rng="item", seed=7, **form); s.init(); s.check()
for _ in range(3): s.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): s.step()
s.check(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print(f"sampler step loop: {dt * 1e3:.2f} ms per iteration ({1 / dt:.1f} it/s)")
