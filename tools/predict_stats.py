"""How often is the R-stream replay's predictor wrong?  N iterations of the default contract at n x m; prints the counters of
the predicted replay (gpirt_sampler_get "rs_stats": mispredictions found by the verification, predictor stalls handed to the
one-phase replay) and the rate.   python tools/predict_stats.py [n = 8192] [m = 1024] [iterations = 200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, RStream
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
its = int(sys.argv[3]) if len(sys.argv) > 3 else 200
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="reference", rstream=RStream(20240), theta_stabilise=True, fstar_fused=False)
s.init(); s.check()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(its):
    s.step()
    if (i + 1) % 50 == 0:
        s.check()
        st = s.get("rs_stats")
        k = s.get("ess_k")
        print(f"after {i + 1:4d} iterations: mispredictions {int(st[1])}, stalls {int(st[2])}, real passes so far {int(st[3])} "
              f"({st[3] / ((i + 1) * m):.3f} per item), mean k {k.mean():.2f}, max k {int(k.max())}", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = s.get("rs_stats")
print(f"{n} x {m}: {its} iterations in {dt:.2f} s = {its / dt:.1f} it/s; {int(st[1])} mispredictions in {its * m} items "
      f"= one per {its / max(int(st[1]), 1):.0f} iterations" + (" (none)" if int(st[1]) == 0 else "") + f"; {int(st[2])} stalls")
