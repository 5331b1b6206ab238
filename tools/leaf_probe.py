import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor, to_device
from gpirt_amd.synthetic import make_responses
h = Handle()
n = 256
th = to_device(make_responses(n, 2, seed=1)[1])
L = h.factor(th)
B = colmajor(n, 2025); B.normal_()
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("leaf256 fwd us:", t(lambda: h.trsm_lower(L, B)))
print("leaf256 bwd us:", t(lambda: h.trsm_lower(L, B, trans=True)))
