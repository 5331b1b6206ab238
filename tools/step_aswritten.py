"""A few iterations with draw_fstar AS WRITTEN (src/draw-fstar.cpp:17-25, item-keyed RNG) at 8192 x 1024 for a kernel trace:
    rocprofv3 --kernel-trace -d out -- python3 tools/step_aswritten.py [n] [m] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=20240, theta_stabilise=True, fstar_fused=False, kstar_rank=0)
s.init(); s.check()
for _ in range(steps):
    s.step()
s.check()
s.enable_timing(True); s.step(); s.check()
torch.cuda.synchronize()
print("stage ms:", {k: round(v, 3) for k, v in s.stage_times().items()})
