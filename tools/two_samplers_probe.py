"""Does a second sampler on the same handle run as fast as it does alone?  (bench.py times the other draw_fstar forms
after the headline form, with the first sampler still alive.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
KW = {"fused": dict(fstar_fused=True, kstar_rank=0), "lowrank": dict(fstar_fused=True, kstar_rank=64)}
def run(form, tag):
    s = Sampler(h, y, th0, rng="item", seed=20240, theta_stabilise=True, **KW[form])
    s.init()
    for _ in range(3): s.step()
    s.check()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): s.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
    s.enable_timing(True); s.step(); st = s.stage_times(); s.enable_timing(False)
    print(f"{tag}: {form} {dt:.3f} ms  { {k: round(v, 2) for k, v in st.items()} }", flush=True)
    return s
a = run("fused", "first sampler")
b = run("lowrank", "second, first alive")
c = run("fused", "third, two alive")
a.close(); b.close()
d = run("fused", "fourth, others closed")
