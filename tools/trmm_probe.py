"""nu = L z (draw_f's product, TRI_A_LOWER) at the per-rank item counts of 1, 2, 4, 8 GPUs: time and rate.
usage: gpurun -- 'python tools/trmm_probe.py'"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor, to_device
from gpirt_amd.synthetic import make_responses

n = 8192
h = Handle()
_, th0 = make_responses(n, 2, seed=11)
L = h.factor(to_device(th0))


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for m in (1024, 512, 256, 128):
    Z = colmajor(n, m); Z.normal_()
    us = t(lambda: h.trmm_lz(L, Z))
    print(f"m={m:5d}: {us:8.1f} us  {n * n * m / us / 1e6:6.1f} TFLOP/s", flush=True)
