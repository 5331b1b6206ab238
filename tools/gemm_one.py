"""One plain NT product (default 4096 x 4096 x 4096), a few launches: the target of PMC passes on the GEMM main loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpirt_amd.ops import Handle, colmajor
M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (4096, 4096, 4096)))
h = Handle()
A = colmajor(M, K); A.normal_(); Bt = colmajor(N, K); Bt.normal_(); C = colmajor(M, N, fill=0.0)
for _ in range(5):
    h.gemm(A, Bt, tb=True, alpha=-1.0, beta=0.0, C_out=C)
torch.cuda.synchronize()
