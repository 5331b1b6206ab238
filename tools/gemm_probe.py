"""GEMM core rates on named shape sets (one MI355X; replaces the five round-1/2 probes).

    python tools/gemm_probe.py [set ...]      sets: small tiles recursion percu blockcol (default: tiles)
  small      short-K / skinny products of the panel and leaf paths, NN beta 0/1 and NT
  tiles      plain NT rate vs tile count: main-loop efficiency vs wave quantisation and the C read-modify-write
  recursion  trsm-recursion update shapes (NN, beta = 1); GPIRT_T128_MIN picks the tile
  percu      per-CU rate of the 128-tile NT kernel vs number of busy CUs (one tile per CU, K = 8192)
  blockcol   C -= A B^T on block-column shapes (M x 1024, K = 1024), the deferred trailing updates
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor

SETS = {
    "small": [(64, 1024, 64), (128, 1024, 128), (64, 2048, 64), (8192, 192, 64), (8192, 64, 64), (2048, 2048, 2048),
              (4096, 1024, 4096), (7936, 7936, 256), (4096, 4096, 256), (4096, 4096, 512)],
    "tiles": [(2048, 4096, 512), (4096, 4096, 512), (4096, 8192, 512), (8192, 8192, 512), (4096, 4096, 1024),
              (4096, 4096, 2048), (4096, 4096, 4096), (8192, 8192, 1024), (128 * 16, 128 * 16, 4096),
              (128 * 23, 128 * 22, 2048), (128 * 32, 128 * 32, 256)],
    "recursion": [(4096, 2025, 4096), (2048, 2025, 2048), (1024, 2025, 1024), (512, 2025, 512), (256, 2025, 256),
                  (1001, 1024, 8192)],
    "percu": [(128, 128, 8192), (512, 512, 8192), (1024, 1024, 8192), (1024, 2048, 8192), (2048, 2048, 8192),
              (2048, 4096, 8192)],
    "blockcol": [(6144, 1024, 1024), (5120, 1024, 1024), (4096, 1024, 1024), (3072, 1024, 1024), (2048, 1024, 1024),
                 (1024, 1024, 1024), (6144, 1024, 2048), (6144, 512, 1024), (7168, 512, 512)],
}


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    names = sys.argv[1:] or ["tiles"]
    if "percu" in names:
        os.environ.setdefault("GPIRT_T128_MIN", "1")
    h = Handle()
    for name in names:
        print(f"--- {name} (GPIRT_T128_MIN={os.environ.get('GPIRT_T128_MIN', 'default')})")
        for (M, N, K) in SETS[name]:
            A = colmajor(M, K); A.normal_()
            B = colmajor(K, N); B.normal_()
            Bt = colmajor(N, K); Bt.normal_()
            C = colmajor(M, N, fill=0.0)
            fl = 2.0 * M * N * K
            nn0 = t(lambda: h.gemm(A, B, C_out=C))
            nn1 = t(lambda: h.gemm(A, B, alpha=-1.0, beta=1.0, C_out=C))
            nt0 = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=0.0, C_out=C))
            nt1 = t(lambda: h.gemm(A, Bt, tb=True, alpha=-1.0, beta=1.0, C_out=C))
            tiles = (M // 128) * (N // 128)
            print(f"M={M:5d} N={N:5d} K={K:5d} tiles128={tiles:5d}: NN b0 {nn0:8.1f} us {fl/nn0/1e6:6.2f} TF | NN b1 {nn1:8.1f} us "
                  f"{fl/nn1/1e6:6.2f} TF | NT b0 {nt0:8.1f} us {fl/nt0/1e6:6.2f} TF | NT b1 {nt1:8.1f} us {fl/nt1/1e6:6.2f} TF")
        if name == "recursion":
            At = colmajor(8192, 1001); At.normal_(); B = colmajor(8192, 1024); B.normal_(); C = colmajor(1001, 1024, fill=0.0)
            c1 = t(lambda: h.gemm(At, B, ta=True, C_out=C))
            print(f"TN 1001x1024x8192: {c1:8.1f} us ({2.0*1001*1024*8192/c1/1e6:6.2f} TF)")


if __name__ == "__main__":
    main()
