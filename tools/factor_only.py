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
