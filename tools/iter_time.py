"""Whole-iteration time at the metric size (8192 x 1024, lowrank form, item RNG), best / mean of `reps` runs of `k` steps,
with the library's own stage times.  GPIRT_* switches are read by the library.
    python tools/iter_time.py [reps = 5] [k = 10] [form = lowrank]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
form = sys.argv[3] if len(sys.argv) > 3 else "lowrank"
kw = {"double_solve": dict(fstar_fused=False, kstar_rank=0), "fused": dict(fstar_fused=True, kstar_rank=0),
      "lowrank": dict(fstar_fused=True, kstar_rank=64)}[form]
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=20240, theta_stabilise=True, **kw)
s.init()
for _ in range(3):
    s.step()
s.check()
ts = []
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        s.step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k * 1e3)
s.check()
s.enable_timing(True); s.step(); st = s.stage_times(); s.enable_timing(False)
tag = " ".join(f"{a}={b}" for a, b in sorted(os.environ.items()) if a.startswith("GPIRT_"))
print(f"{form}: min {min(ts):.3f} ms ({1e3/min(ts):.1f} it/s)  mean {sum(ts)/len(ts):.3f} ms  stages { {a: round(b, 3) for a, b in st.items()} }  [{tag}]", flush=True)
