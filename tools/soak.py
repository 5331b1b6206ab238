"""Repeatability soak for the flag-synchronised kernels: run the same chain twice in one process (fresh samplers) and
require bit-identical state after every iteration -- a lost hand-off or a stale read would show up as a mismatch,
NaN or the panel guard.  usage: python tools/soak.py [n=3000] [m=96] [iters=150] [load=0] [rng=item|reference]
rng = reference: the default contract -- the R-stream replay's draw_f (rs3_slice_kernel: three meetings of up to 256
work-groups per pass) under the same repeat / foreign-load regime.
load = 1: the SECOND run has foreign work beside it -- a torch stream streaming 1 GiB copies and fp64 matmuls of changing
size, unsynchronised with the sampler -- so the hand-offs are exercised under UNEVEN load (MI355X_MICROARCH.md: "test every
hand-off under uneven load ... checking every word"): every checkpoint must still match the quiet first run bit for bit."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 96
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 150
load = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rng = sys.argv[5] if len(sys.argv) > 5 else "item"
y, th0 = make_responses(n, m, seed=7)
h = Handle()
bg = torch.cuda.Stream()
src = torch.empty(1 << 27, dtype=torch.float64, device="cuda").normal_() if load else None
dst = torch.empty_like(src) if load else None
def disturb(it):
    k = 512 + 256 * (it % 7)
    with torch.cuda.stream(bg):
        dst.copy_(src)                                   # 1 GiB through HBM
        a = src[: k * k].view(k, k)
        torch.mm(a, a, out=dst[: k * k].view(k, k))      # MFMA + LDS traffic of a size that changes every iteration
def run(disturbed=False):
    if rng == "reference":
        from gpirt_amd.ops import RStream
        s = Sampler(h, y, th0, rng="reference", rstream=RStream(11), theta_stabilise=True)
    else:
        s = Sampler(h, y, th0, rng="item", seed=11, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
    s.init()
    digs = []
    for it in range(iters):
        if disturbed: disturb(it)
        s.step()
        if it % 10 == 9 or it == iters - 1:
            s.check()
            d = hashlib.sha256()
            for name in ("theta", "f", "beta", "L"):
                a = s.get(name)
                if not np.isfinite(a).all(): raise SystemExit(f"non-finite {name} at iteration {it}")
                d.update(np.ascontiguousarray(a).tobytes())
            digs.append(d.hexdigest())
    return digs
a = run(); b = run(disturbed=bool(load))
bad = [i for i, (x, y_) in enumerate(zip(a, b)) if x != y_]
print(f"rng={rng} n={n} m={m} iters={iters}: {len(a)} checkpoints, mismatches at {bad}" if bad else f"rng={rng} n={n} m={m} iters={iters}: {len(a)} checkpoints bit-identical across two runs" + (" (second run beside foreign load)" if load else ""))
sys.exit(1 if bad else 0)
