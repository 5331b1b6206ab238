"""Three iterations of the DEFAULT contract (R-stream replay, everything as written but theta_stabilise) for a kernel trace:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/rstream_step.py [n = 8192] [m = 64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=True, fstar_fused=False)
s.init(); s.check(); s.step(); s.check()
s.enable_timing(True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    s.step()
s.check(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
import time as _t
t1 = _t.perf_counter(); s.step(); t2 = _t.perf_counter()
import numpy as _np
_k = s.get("ess_k"); print("rejection counts: max", int(_k.max()), "histogram", _np.bincount(_k.astype(int), minlength=12)[:24].tolist())
print("stage ms:", {k: round(v, 2) for k, v in s.stage_times().items()}, f"; wall of that step {1e3 * (t2 - t1):.1f} ms")
print(f"{n} x {m}: {dt * 1e3:.2f} ms per iteration = {dt / m * 1e6:.1f} us per item all told; mean k {s.get('ess_k').mean():.2f}")
