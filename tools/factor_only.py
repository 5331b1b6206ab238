"""A few factorisations of K(theta, theta) + jitter at n (default 8192): the target of rocprofv3 traces (tools/trace_factor.sh).
GPIRT_SCHED etc. are read by the library."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
h = Handle()
theta = torch.from_numpy(make_responses(n, 2, seed=1)[1]).cuda()
for _ in range(3):
    L = h.factor(theta)
torch.cuda.synchronize()
