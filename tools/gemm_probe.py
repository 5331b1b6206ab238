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
for (M, N, K) in [(64, 1024, 64), (128, 1024, 128), (64, 2048, 64), (8192, 192, 64), (8192, 64, 64), (2048, 2048, 2048), (4096, 1024, 4096), (7936, 7936, 256), (4096,4096,256), (4096, 4096, 512)]:
    A = colmajor(M, K); A.normal_(); B = colmajor(K, N); B.normal_(); Bt = colmajor(N, K); Bt.normal_(); C = colmajor(M, N, fill=0.0)
    a = t(lambda: h.gemm(A, B, C_out=C))
    b = t(lambda: h.gemm(A, B, alpha=-1.0, beta=1.0, C_out=C))
    c = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=1.0, C_out=C))
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d}: NN beta0 {a:8.1f} us ({fl/a/1e6:6.2f} TF) | NN beta1 {b:8.1f} us | NT beta1 {c:8.1f} us ({fl/c/1e6:6.2f} TF)")
