"""Repeatability soak for the flag-synchronised kernels: run the same chain twice in one process (fresh samplers) and
require bit-identical state after every iteration -- a lost hand-off or a stale read would show up as a mismatch,
NaN or the panel guard.  usage: python tools/soak.py [n=3000] [m=96] [iters=150]"""
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
y, th0 = make_responses(n, m, seed=7)
h = Handle()
def run():
    s = Sampler(h, y, th0, rng="item", seed=11, theta_stabilise=True, fstar_fused=True, kstar_rank=64)
    s.init()
    digs = []
    for it in range(iters):
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
a = run(); b = run()
bad = [i for i, (x, y_) in enumerate(zip(a, b)) if x != y_]
print(f"n={n} m={m} iters={iters}: {len(a)} checkpoints, mismatches at {bad}" if bad else f"n={n} m={m} iters={iters}: {len(a)} checkpoints bit-identical across two runs")
sys.exit(1 if bad else 0)
