"""Synthetic 2PL response data of the shape BASELINE.json's configs name.

theta_i ~ N(0,1); intercept a_j ~ U(-2,2), slope b_j ~ U(0.5,3) (mirrors the simulation in the
reference's roxygen example, R/gpirtMCMC.R:49-68); y_ij = +1 w.p. plogis(a_j + b_j theta_i) else -1;
a fraction `na_frac` of entries set to NaN (senate116 has 5.7 %); unanimous columns are re-drawn.
"""
from __future__ import annotations

import numpy as np

CONFIGS = {
    "C2": (1024, 256),
    "C3": (4096, 1024),
    "C4": (8192, 2048),
    "C5": (16384, 4096),
    "M": (8192, 1024),
}


def make_responses(n: int, m: int, seed: int = 20240, na_frac: float = 0.05, snap_theta: bool = True,
                   return_truth: bool = False):
    """Returns (y [n x m, column-major float64 in {-1,+1,NaN}], theta_init [n]) and, with return_truth,
    the generating theta as a third value."""
    rng = np.random.default_rng(seed)
    theta_true = rng.standard_normal(n)
    a = rng.uniform(-2.0, 2.0, m)
    b = rng.uniform(0.5, 3.0, m)
    y = np.empty((n, m), order="F")
    for j in range(m):
        while True:
            p = 1.0 / (1.0 + np.exp(-(a[j] + b[j] * theta_true)))
            col = np.where(rng.random(n) < p, 1.0, -1.0)
            col[rng.random(n) < na_frac] = np.nan
            ok = col[~np.isnan(col)]
            if len(np.unique(ok)) == 2:
                break
            a[j] = rng.uniform(-2.0, 2.0)
        y[:, j] = col
    theta_init = rng.standard_normal(n)
    if snap_theta:
        # steady-state condition of the sampler: every theta lies on the -5:0.01:5 grid (Q6)
        k = np.clip(np.rint((theta_init + 5.0) / 0.01), 0, 1000)
        theta_init = -5.0 + k * 0.01
    if return_truth:
        return y, theta_init, theta_true
    return y, theta_init
