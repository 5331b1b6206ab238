"""The drop-in at the reference's OWN sizes (configs C1 senate116 100 x 418, C2 1024 x 256): whole gpirt_mcmc() calls per
second under both RNG contracts, beside the CPU restatement (oracle/, one thread) on the same inputs -- is the library
worth loading for the problems the package's users actually have?  (tools/: may import the oracle, it is not product code.)
    python tools/small_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from gpirt_amd import gpirtMCMC
from gpirt_amd.ops import RStream
from gpirt_amd.synthetic import make_responses
from oracle import oracle
oracle.build()

codes = dict(yea=[1], nay=[-1], missing=[None])
d = np.load(os.path.join(ROOT, "tests", "golden", "senate116_y.npz"))
y1 = d["y"].astype(float); y1[y1 == 0] = np.nan
cases = [("C1 senate116", y1, None, 30)]
y2, th2 = make_responses(1024, 256, seed=20241)
cases.append(("C2 1024 x 256", y2, th2, 6))
for name, y, th0, iters in cases:
    n, m = y.shape
    if th0 is None:
        th0 = np.random.default_rng(1).standard_normal(n)
    for rng, kw in (("reference", dict(rstream=RStream(1119))), ("item", dict(seed=7, theta_stabilise=True))):
        gpirtMCMC(y, 2, 0, vote_codes=codes, theta_init=th0, rng=rng, **(dict(rstream=RStream(5)) if rng == "reference" else dict(seed=7, theta_stabilise=True)))
        t0 = time.perf_counter()
        gpirtMCMC(y, iters, 0, vote_codes=codes, theta_init=th0, rng=rng, **kw)
        dt = time.perf_counter() - t0
        print(f"{name} ({n} x {m}) GPU rng={rng}: {iters} iterations in {dt:.2f} s = {iters / dt:.1f} it/s (whole call, setup included)", flush=True)
    k = max(1, iters // 6)
    t0 = time.perf_counter()
    oracle.gpirt_mcmc(oracle.RStream(1119), y, th0, k, 0)
    dt = time.perf_counter() - t0
    print(f"{name} CPU restatement, 1 thread: {k} iterations in {dt:.2f} s = {k / dt:.2f} it/s", flush=True)
