"""A few factorisations of the sampler's (bordered) S at n (default 8192), for rocprofv3 traces (tools/trace_factor.sh).
GPIRT_* switches are read by the library.   python tools/factor_only.py [n] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
y, th0 = make_responses(n, 8, seed=n)
th0 = -5.0 + np.clip(np.rint((th0 + 5.0) / 0.01), 0, 1000) * 0.01
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=1, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
s.init(); s.check()
for _ in range(reps):
    s.factor()
s.check()
print("guard fallbacks", h.guard_fallbacks)
