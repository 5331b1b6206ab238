import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from gpirt_amd.ops import Handle, to_device
import test_gpu_theta_fixed as T
h = Handle()
n, m = 8192, 1024
y, fstar = T._inputs(n, m, seed=1)
yd, fd = to_device(y), to_device(fstar)
for mode in (1, 2):
    with h.config("GPIRT_THETA_FIXED", mode):
        for _ in range(3): h.theta_logpost(yd, fd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): h.theta_logpost(yd, fd)
        torch.cuda.synchronize(); print("mode", mode, (time.perf_counter() - t0) / 20 * 1e3, "ms per call (incl. indicators + copy)")
