"""Plain NT GEMM rate vs tile count: separates main-loop efficiency from wave quantisation and the C read-modify-write."""
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
# tiles = (M/128)*(N/128); 512 resident work-groups (2 per CU)
for (M, N, K) in [(2048, 4096, 512), (4096, 4096, 512), (4096, 8192, 512), (8192, 8192, 512), (4096, 4096, 1024), (4096, 4096, 2048), (4096, 4096, 4096), (8192, 8192, 1024),
                  (128 * 16, 128 * 16, 4096), (128 * 23, 128 * 22, 2048), (128*32, 128*32, 256)]:
    A = colmajor(M, K); A.normal_(); Bt = colmajor(N, K); Bt.normal_(); C = colmajor(M, N, fill=0.0)
    c0 = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=0.0, C_out=C))
    c1 = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=1.0, C_out=C))
    fl = 2.0 * M * N * K
    tiles = (M // 128) * (N // 128)
    print(f"M={M:5d} N={N:5d} K={K:5d} tiles={tiles:5d} ({tiles/512:5.2f} rounds): NT beta0 {c0:8.1f} us ({fl/c0/1e6:6.2f} TF) | beta1 {c1:8.1f} us ({fl/c1/1e6:6.2f} TF)")
