"""The two large per-iteration products at the shapes the BASELINE configurations and their per-rank shards produce:
nu = L z (TRI_A_LOWER, n x n x m) and the draw_theta product (1001 x n x 2m).  Flags anything under 45 TFLOP/s.
usage: gpurun -- 'python tools/shape_probe.py'"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor, to_device
from gpirt_amd.synthetic import make_responses

h = Handle()


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (4096, 8192, 16384):
    _, th0 = make_responses(n, 2, seed=11)
    L = h.factor(to_device(th0))
    for m in (128, 256, 512, 1024, 2048, 4096):
        if n * m * 8 > 1 << 30: continue
        Z = colmajor(n, m); Z.normal_()
        us = t(lambda: h.trmm_lz(L, Z))
        tf = n * n * m / us / 1e6
        G = colmajor(1001, 2 * m); G.normal_()
        Y = colmajor(2 * m, n); Y.normal_()
        C = colmajor(1001, n, fill=0.0)
        us2 = t(lambda: h.gemm(G, Y, C_out=C))
        tf2 = 2.0 * 1001 * n * 2 * m / us2 / 1e6
        print(f"n={n:6d} m={m:5d}: L z {us:8.1f} us {tf:5.1f} TF{'  <--' if tf < 45 else ''} | theta product {us2:8.1f} us {tf2:5.1f} TF{'  <--' if tf2 < 45 else ''}", flush=True)
    del L
