import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n, m = 8192, 1024
y, th0 = make_responses(n, m, seed=20240)
h = Handle()
s = Sampler(h, y, th0, rng="item", seed=1, fstar_fused=True, kstar_rank=64)
s.init()
for _ in range(3): s.step()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(3):
    torch.cuda.synchronize()
    e0 = ev(); s.draw_f(); e1 = ev(); s.draw_fstar(); e2 = ev(); s.theta_partial(); s.theta_finish(); e3 = ev(); s.draw_beta(); e4 = ev(); s.factor(); e5 = ev()
    torch.cuda.synchronize()
    print(os.environ.get("GPIRT_AUX"), "draw_f %.3f draw_fstar %.3f theta %.3f beta %.3f factor %.3f total %.3f" % (e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3), e3.elapsed_time(e4), e4.elapsed_time(e5), e0.elapsed_time(e5)))
