"""A few iterations of the headline configuration (gpirt_fast_options, 8192 x 1024) for a kernel trace:
    rocprofv3 --kernel-trace -d out -- python3 tools/step_only.py [n = 8192] [m = 1024] [steps = 6]
tools/trace_step.sh / tools/timeline_window.py turn the trace into the per-launch timeline of the last iteration."""
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
s = Sampler(h, y, th0, preset="fast", seed=20240)
s.init(); s.check()
for _ in range(steps):
    s.step()
s.check()
torch.cuda.synchronize()
print("done")
