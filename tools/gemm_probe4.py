"""Per-CU rate of the 128-tile NT kernel as a function of how many CUs are busy (one tile per CU, K = 8192): separates
a clock / power ceiling from a per-CU pipeline limit."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault("GPIRT_T128_MIN", "1")
from gpirt_amd.ops import Handle, colmajor
h = Handle()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
K = 8192
for (M, N) in [(128, 128), (512, 512), (1024, 1024), (1024, 2048), (2048, 2048), (2048, 4096)]:
    A = colmajor(M, K); A.normal_(); Bt = colmajor(N, K); Bt.normal_(); C = colmajor(M, N, fill=0.0)
    c0 = t(lambda: h.gemm(A, Bt, tb=True, alpha=1.0, beta=0.0, C_out=C))
    fl = 2.0 * M * N * K
    tiles = (M // 128) * (N // 128)
    print(f"tiles={tiles:4d}: {c0:8.1f} us  {fl/c0/1e6:6.2f} TF  per-tile rate {fl/c0/1e3/tiles:6.1f} GF/s  ({c0*1e3/(K/16*64):.1f} ns per MFMA per wave)")
