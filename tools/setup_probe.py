"""Where the fixed cost of a call goes: handle creation, sampler creation (device buffers, pinned windows, y upload), init.
    python tools/setup_probe.py [n = 8192] [m = 1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
y, th0 = make_responses(n, m, seed=20240)
torch.cuda.synchronize()
for rep in range(2):
    for rng in ("reference", "item"):
        t0 = time.perf_counter(); h = Handle(); torch.cuda.synchronize(); t1 = time.perf_counter()
        kw = dict(rstream=RStream(7)) if rng == "reference" else dict(seed=7)
        s = Sampler(h, y, th0, rng=rng, theta_stabilise=True, **kw); torch.cuda.synchronize(); t2 = time.perf_counter()
        s.init(); s.check(); torch.cuda.synchronize(); t3 = time.perf_counter()
        s.step(); s.check(); torch.cuda.synchronize(); t4 = time.perf_counter()
        s.close(); h.close(); t5 = time.perf_counter()
        print(f"rng={rng}: handle {t1 - t0:.3f} s, sampler create {t2 - t1:.3f} s, init {t3 - t2:.3f} s, first step {t4 - t3:.3f} s, close {t5 - t4:.3f} s", flush=True)
