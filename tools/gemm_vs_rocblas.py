"""The library's fp64 MFMA GEMM beside rocBLAS (torch.mm) on the shapes the sampler issues: how much of the gap to the
78.6 TFLOP/s peak is the kernel's and how much is what any fp64 GEMM gets on this chip.
usage: gpurun -- 'python tools/gemm_vs_rocblas.py'"""
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


for (M, N, K, tb) in [(8192, 8192, 8192, False), (8192, 1024, 8192, False), (7168, 7168, 1024, True), (4096, 4096, 1024, True),
                      (1001, 8192, 2048, False), (8192, 8192, 1024, True), (3072, 3072, 1024, True)]:
    A = colmajor(M, K); A.normal_()
    B = colmajor(N, K) if tb else colmajor(K, N); B.normal_()
    C = colmajor(M, N, fill=0.0)
    mine = t(lambda: h.gemm(A, B, tb=tb, alpha=-1.0, beta=1.0, C_out=C))
    # rocBLAS through torch: column-major X (M x K) is the row-major tensor X.T; C^T = op(B)^T A^T
    At, Bt, Ct = A.t(), B.t(), C.t()          # row-major views (K x M), (.. ), (N x M)
    def lib():
        if tb: torch.addmm(Ct, B, At, beta=1.0, alpha=-1.0, out=Ct)       # (N x K)(K x M)
        else:  torch.addmm(Ct, Bt, At, beta=1.0, alpha=-1.0, out=Ct)      # (N x K)(K x M)
    try:
        roc = t(lib)
    except Exception as e:
        roc = float("nan"); print("rocBLAS call failed:", e)
    fl = 2.0 * M * N * K
    print(f"M={M:5d} N={N:5d} K={K:5d} {'NT' if tb else 'NN'}: this library {mine:9.1f} us {fl/mine/1e6:6.2f} TF ({fl/mine/1e6/78.6:.3f})"
          f" | rocBLAS {roc:9.1f} us {fl/roc/1e6:6.2f} TF ({fl/roc/1e6/78.6:.3f})", flush=True)
