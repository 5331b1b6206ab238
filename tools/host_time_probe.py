"""Host enqueue time vs device time of one sampler step (is the host ahead of the GPU?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=7)
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=11, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
s.init()
for _ in range(3): s.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); host = []
for _ in range(10):
    a = time.perf_counter(); s.step(); host.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue per step: {1e3*sum(host)/10:.2f} ms (max {1e3*max(host):.2f}); wall per step incl. drain: {1e3*(t2-t0)/10:.2f} ms")
