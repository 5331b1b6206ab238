import sys, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from gpirt_amd.ops import Handle, to_device
import test_gpu_theta_fixed as T
h = Handle()
for n, m in [(64, 77), (64, 96), (64, 112), (64, 128), (64, 144), (64, 200), (64, 256), (64, 512), (256, 1024)]:
    y, fstar = T._inputs(n, m, seed=1)
    (fx, fb1), (ge, fb2) = T._both(h, y, fstar)
    d = np.abs(fx - ge)
    bad = np.argwhere(d > 1e-9)
    print(n, m, "max diff", d.max(), "bad entries", len(bad), "of", d.size, "first", bad[:4].tolist(), flush=True)
    if len(bad):
        g, i = bad[0]
        print("   fx", fx[g, i], "ge", ge[g, i], "ratio", fx[g, i] / ge[g, i])
        print("   bad g range", bad[:, 0].min(), bad[:, 0].max(), "i range", bad[:, 1].min(), bad[:, 1].max())
