"""draw_fstar with the rank-r Chebyshev factorisation of K(theta, theta*) against every grid column solved
(both `fused`): same theta / f / L, compare f* (identical RNG keys), the means and s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gpirt_amd.ops import Handle
from gpirt_amd.sampler import Sampler
from gpirt_amd.synthetic import make_responses
h = Handle()
for (n, m, r) in ((2048, 64, 64), (8192, 256, 64), (8192, 256, 48), (3000, 100, 80)):
    y, th0 = make_responses(n, m, seed=3)
    outs = []
    for rank in (0, r):
        s = Sampler(h, y, th0, rng="item", seed=5, theta_stabilise=True, fstar_fused=True, kstar_rank=rank)
        s.init()
        s.draw_f(); s.draw_fstar(); s.check()
        outs.append((np.array(s.get("fstar")), np.array(s.get("mu_star"))))
        del s
    a, b = outs[0][0], outs[1][0]
    fin = np.isfinite(a) & np.isfinite(b)
    print(f"n={n} m={m} r={r}: max|f*_lowrank - f*_full| = {np.abs(a[fin] - b[fin]).max():.3e}   max|f*| {np.abs(a[fin]).max():.2f}   "
          f"non-finite: {np.count_nonzero(~np.isfinite(a))} vs {np.count_nonzero(~np.isfinite(b))}")
