"""sha256 of the Cholesky factor (lower triangle; for the sampler cases the whole ldl x n buffer with the rows of the
bordered factorisation) for a fixed list of sizes: one line per case.  Run under two builds of the library
(GPIRT_HIP_LIBRARY) or two settings of a GPIRT_* switch and compare the output line by line."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gpirt_amd.ops import Handle, to_device
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses

h = Handle()
for n in (257, 1000, 2500, 8192):
    rng = np.random.default_rng(n)
    theta = -5.0 + np.clip(np.rint((rng.standard_normal(n) + 5.0) / 0.01), 0, 1000) * 0.01
    L = torch.tril(h.factor(to_device(theta)))
    torch.cuda.synchronize()
    print("operator", n, hashlib.sha256(L.cpu().numpy().tobytes()).hexdigest(), flush=True)
for n, kw in ((8192, dict(fstar_fused=True, kstar_rank=64)), (4096, dict(fstar_fused=True)), (16384, dict(fstar_fused=True, kstar_rank=64))):
    y, th0 = make_responses(n, 4, seed=3)
    s = Sampler(h, y, th0, rng="item", seed=1, **kw)
    s.init(); s.check()
    buf = s.device_tensor("L")
    print("sampler", n, sorted(kw.items()), hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest(), flush=True)
    s.close()
