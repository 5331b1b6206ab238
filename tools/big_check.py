"""Property check at n = 16384 (config C5's respondent count): residual of the factorisation, trsm round trip."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd.ops import Handle, colmajor, to_device
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
h = Handle()
_, th0 = make_responses(n, 2, seed=11)
th = to_device(th0)
torch.cuda.synchronize(); t0 = time.time()
L = h.factor(th)
torch.cuda.synchronize(); t1 = time.time()
L = h.factor(th)
torch.cuda.synchronize(); t2 = time.time()
S = h.se_kernel(th, th, jitter=0.001)
R = h.gemm(L, L, tb=True)
print(f"n={n} factor {1e3*(t2-t1):.1f} ms ({n**3/3/(t2-t1)/1e12:.1f} TF)  resid {(torch.linalg.norm(R-S)/torch.linalg.norm(S)).item():.3e}")
del R, S
B = colmajor(n, 512); B.normal_()
X = h.trsm_lower(L, B.clone().T.contiguous().T)
print("trsm round trip", (h.gemm(L, X) - B).abs().max().item())
X = h.trsm_lower(L, B.clone().T.contiguous().T, trans=True)
print("trsm^T round trip", (h.gemm(L, X, ta=True) - B).abs().max().item())
