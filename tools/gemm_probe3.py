"""trsm-recursion update shapes (NN, beta = 1) under the 128-tile and the 64-tile kernel (GPIRT_T128_MIN picks)."""
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
for (M, N, K) in [(4096, 2025, 4096), (2048, 2025, 2048), (1024, 2025, 1024), (512, 2025, 512), (256, 2025, 256), (1001, 1024, 8192)]:
    A = colmajor(M, K); A.normal_(); B = colmajor(K, N); B.normal_(); C = colmajor(M, N, fill=0.0)
    c1 = t(lambda: h.gemm(A, B, alpha=-1.0, beta=1.0, C_out=C))
    fl = 2.0 * M * N * K
    print(f"T128_MIN={os.environ.get('GPIRT_T128_MIN','default')} M={M:5d} N={N:5d} K={K:5d}: NN beta1 {c1:8.1f} us ({fl/c1/1e6:6.2f} TF)")
At = colmajor(8192, 1001); At.normal_(); B = colmajor(8192, 1024); B.normal_(); C = colmajor(1001, 1024, fill=0.0)
c1 = t(lambda: h.gemm(At, B, ta=True, C_out=C))
print(f"TN 1001x1024x8192: {c1:8.1f} us ({2.0*1001*1024*8192/c1/1e6:6.2f} TF)")
