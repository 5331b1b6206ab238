"""NT product C -= A B^T on block-column shapes (M x 1024, K = 1024) with the 64-tile and the 128-tile kernel (GPIRT_T128_MIN picks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor
h = Handle()
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in [(6144, 1024, 1024), (5120, 1024, 1024), (4096, 1024, 1024), (3072, 1024, 1024), (2048, 1024, 1024), (1024, 1024, 1024), (6144, 1024, 2048), (6144, 512, 1024), (7168, 512, 512)]:
    A = colmajor(M, K); A.normal_(); Bt = colmajor(N, K); Bt.normal_(); C = colmajor(M, N, fill=0.0)
    c1 = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=1.0, C_out=C))
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d} tiles64={(M//64)*(N//64):5d} tiles128={(M//128)*(N//128):4d}: {c1:8.1f} us ({fl/c1/1e6:6.2f} TF)")
