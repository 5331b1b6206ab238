import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=1, fstar_fused=True, kstar_rank=int(os.environ.get("KSTAR_RANK", "64")))
s.init()
for _ in range(3): s.step()
s.check()
