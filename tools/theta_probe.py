"""draw_theta's log-posterior product (NT, 1001 x nb x 2m) at the respondent-block sizes of 1, 2, 4, 8 ranks.
(Through the operator entry the last, partial row tile takes the predicated path; the sampler pads its operand to whole
128-row tiles and reads them unpredicated: 0.52 ms = 65 TFLOP/s for the first line instead of 0.66 ms here.)
usage: gpurun -- 'python tools/theta_probe.py'"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor

h = Handle()


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, m) in ((8192, 1024), (8192, 2048), (4096, 1024)):
    for G in (1, 2, 4, 8):
        nb = n // G
        A = colmajor(1008, 2 * m)[:1001]; A.normal_()       # leading dimension padded to an even number, as the sampler's is
        Bt = colmajor(nb, 2 * m); Bt.normal_()
        C = colmajor(1001, nb, fill=0.0)
        us = t(lambda: h.gemm(A, Bt, tb=True, C_out=C))
        print(f"n={n} m={m} ranks={G}: 1001 x {nb:5d} x {2 * m}: {us:8.1f} us  {2.0 * 1001 * nb * 2 * m / us / 1e6:5.1f} TFLOP/s", flush=True)
