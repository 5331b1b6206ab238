"""cpu_baseline leg of bench.py: the CPU restatement of the reference timed on the
GPU box's host cores, on a bounded sample of the same workload.  TEST / MEASUREMENT INFRASTRUCTURE ONLY --
nothing in the product path imports this.

Two legs, both per MCMC iteration of src/gpirtMCMC.cpp:68-78 at (n, m):
  (i)  reference-shaped, ONE thread: the line-following C restatement (oracle/gpirt_oracle.c) with the unblocked
       Cholesky -- the per-item BLAS-2 structure stock R + reference BLAS executes;
  (ii) the same restatement on ALL host cores, ONE WHOLE ITERATION AT THE FULL SIZE, nothing extrapolated: blocked Cholesky
       with OpenMP (oracle/oracle_fast.c) and the item / grid-column / respondent loops of draw_f, draw_fstar, draw_theta,
       draw_beta spread over a thread pool (oracle/parallel.py: the C calls release the GIL; items are independent under
       the item RNG) -- the driver the metric-size parity test checks every draw of the device against.
Every piece of (i) is timed on a sample and scaled linearly in the sampled dimension (cubically for the unblocked
Cholesky, quadratically for K); the factors are returned so the reader can see how far each number is stretched.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O
from . import parallel as P

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def host_cores() -> int:
    """Cores this process may actually use: cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _p(a):
    return a.ctypes.data_as(_dp)


def _F(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _draw_f(lib, rng, it, f, y, L, mu):
    n, m = f.shape
    out = np.empty((n, m), order="F")
    k = np.zeros(m, dtype=np.int32)
    lib.orc_draw_f(rng.ref, C.c_uint32(it), _p(f), _p(y), _p(L), _p(mu), C.c_int64(n), C.c_int64(m), _p(out),
                   k.ctypes.data_as(_ip))


def _draw_fstar(lib, rng, it, f, theta, ts, L, mu_star):
    n, m = f.shape
    N = len(ts)
    out = np.empty((N, m), order="F")
    s = np.empty(N)
    mean = np.empty((N, m), order="F")
    lib.orc_draw_fstar(rng.ref, C.c_uint32(it), _p(f), _p(theta), _p(ts), _p(L), _p(mu_star), C.c_int64(n), C.c_int64(m),
                       C.c_int64(N), _p(out), _p(s), _p(mean))


def _draw_theta(lib, rng, it, ts, prior, y, fstar):
    n, m = y.shape
    out = np.empty(n)
    lib.orc_draw_theta(rng.ref, C.c_uint32(it), _p(ts), _p(y), _p(prior), _p(fstar), C.c_int64(n), C.c_int64(m),
                       C.c_int64(len(ts)), C.c_int(1), _p(out))


def _draw_beta(lib, rng, it, beta, theta, y, f, pm, ps, st):
    n, m = y.shape
    out = np.empty((2, m), order="F")
    lib.orc_draw_beta(rng.ref, C.c_uint32(it), _p(beta), _p(theta), _p(y), _p(f), _p(pm), _p(ps), _p(st), C.c_int64(n),
                      C.c_int64(m), _p(out))


def _timed(fn):
    t0 = time.perf_counter()
    fn()
    return time.perf_counter() - t0


def _pool_time(pool, jobs):
    """wall time of running all jobs (callables) on the pool"""
    t0 = time.perf_counter()
    list(pool.map(lambda j: j(), jobs))
    return time.perf_counter() - t0


def run(n, m, y, theta, L, f, beta, mu, fstar, nthreads=None):
    """y n x m, theta n, L n x n, f n x m, beta 2 x m, mu n x m, fstar 1001 x m: the sampler's state (host arrays)."""
    lib = O.lib()
    nthreads = nthreads or host_cores()
    N = O.NGRID
    ts = O.theta_star()
    prior = np.array([lib.orc_dnorm_log(t, 0.0, 1.0) for t in ts])
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    L = np.asfortranarray(L)                      # shared, read-only (no copy if already column-major)
    one, allc = {}, {}
    factors = {}

    # ---- K + chol --------------------------------------------------------------------------------
    n_s = min(n, 2048)
    S = np.empty((n_s, n_s), order="F")
    tK = _timed(lambda: lib.orc_se_kernel(_p(theta[:n_s].copy()), C.c_int64(n_s), _p(theta[:n_s].copy()), C.c_int64(n_s), _p(S)))
    S[np.diag_indices_from(S)] += 0.001
    tC = _timed(lambda: lib.orc_potrf_lower(_p(S), C.c_int64(n_s)))                     # unblocked, one thread
    one["K"] = tK * (n / n_s) ** 2
    one["chol"] = tC * (n / n_s) ** 3
    factors["one.chol"] = (n / n_s) ** 3
    Sf = np.empty((n, n), order="F")
    tKf = _timed(lambda: lib.orc_se_kernel(_p(theta), C.c_int64(n), _p(theta), C.c_int64(n), _p(Sf)))
    Sf[np.diag_indices_from(Sf)] += 0.001
    lib.orc_potrf_lower_blocked.restype = C.c_int
    tCf = _timed(lambda: lib.orc_potrf_lower_blocked(_p(Sf), C.c_int64(n), C.c_int(nthreads)))   # full size, all cores
    allc["K"] = tKf                  # the restatement's K() is a scalar loop (src/covariance-function.cpp:3-14): not threaded
    allc["chol"] = tCf
    factors["all.chol"] = 1.0
    del S, Sf

    # ---- draw_f ----------------------------------------------------------------------------------
    mi = min(m, 16)
    fi, yi, mui = _F(f[:, :mi]), _F(y[:, :mi]), _F(mu[:, :mi])
    one["draw_f"] = _timed(lambda: _draw_f(lib, O.ItemStream(1), 1, fi, yi, L, mui)) * (m / mi)
    factors["one.items"] = m / mi
    allc["draw_f"] = _timed(lambda: P.draw_f(1, 1, f, y, L, mu, nthreads))            # all m items, full size
    if True:
        # ---- draw_fstar: t = c * (grid columns) + p * (items); two calls separate c and p ---------
        gs = 16
        mu_star = _F(beta[0][None, :mi] + ts[:gs, None] * beta[1][None, :mi])
        tsg = ts[:gs].copy()
        tA = _timed(lambda: _draw_fstar(lib, O.ItemStream(1), 1, fi, theta, tsg, L, mu_star))
        f1, ms1 = _F(fi[:, :1]), _F(mu_star[:, :1])
        tB = _timed(lambda: _draw_fstar(lib, O.ItemStream(1), 1, f1, theta, tsg, L, ms1))
        p = max(tA - tB, 0.0) / max(mi - 1, 1)
        c = max(tB - p, 0.0) / gs
        one["draw_fstar"] = c * N + p * m
        factors["one.grid_columns"] = N / gs
        mu_star_full = _F(beta[0][None, :] + ts[:, None] * beta[1][None, :])
        allc["draw_fstar"] = _timed(lambda: P.draw_fstar(1, 1, f, theta, L, mu_star_full, nthreads))   # 1001 grid columns + all m items

        # ---- draw_theta ---------------------------------------------------------------------------
        fs = _F(fstar)
        ns = min(n, 128)
        ys = _F(y[:ns, :])
        one["draw_theta"] = _timed(lambda: _draw_theta(lib, O.ItemStream(1), 1, ts, prior, ys, fs)) * (n / ns)
        factors["one.respondents"] = n / ns
        allc["draw_theta"] = _timed(lambda: P.draw_theta(1, 1, y, fs, True, nthreads))     # all n respondents

        # ---- draw_beta ----------------------------------------------------------------------------
        pm, ps, st = np.zeros((2, mi), order="F"), np.full((2, mi), 3.0, order="F"), np.full((2, mi), 0.1, order="F")
        bi = _F(beta[:, :mi])
        one["draw_beta"] = _timed(lambda: _draw_beta(lib, O.ItemStream(1), 1, bi, theta, yi, fi, pm, ps, st)) * (m / mi)
        pmf, psf, stf = np.zeros((2, m), order="F"), np.full((2, m), 3.0, order="F"), np.full((2, m), 0.1, order="F")
        allc["draw_beta"] = _timed(lambda: P.draw_beta(1, 1, beta, theta, y, f, pmf, psf, stf, nthreads))

    t_one, t_all = sum(one.values()), sum(allc.values())
    # The top-level fields are the MEASURED leg (one whole iteration at the full size on all host cores, nothing
    # extrapolated); the reference-shaped single-thread figure, which is stretched from samples, sits beside it.
    return {
        "value": 1.0 / t_all, "unit": "iterations/s", "cores": nthreads, "kind": "port", "extrapolated": False,
        "sample": (f"C restatement of the reference on {nthreads} threads, ONE WHOLE ITERATION AT THE FULL SIZE ({n} x {m}), nothing extrapolated: "
                   f"K (scalar loop, one thread, as src/covariance-function.cpp:3-14), blocked OpenMP potrf, draw_f over all {m} items, "
                   f"draw_fstar over all {N} grid columns and {m} items, draw_theta over all {n} respondents, draw_beta "
                   f"(oracle/parallel.py: the driver tests/test_gpu_metric_oracle.py checks every draw of the device against)"),
        "stage_seconds": {k: round(v, 3) for k, v in allc.items()},
        "host_cores": nthreads,
        "single_thread_reference_shaped": {
            "value": 1.0 / t_one, "unit": "iterations/s", "cores": 1, "kind": "port", "extrapolated": True,
            "sample": (f"C restatement of the reference (oracle/gpirt_oracle.c, unblocked potrf, per-item BLAS-2 structure) on "
                       f"1 thread: K + chol on the leading {n_s} respondents, draw_f / draw_beta on {mi} of {m} items, draw_fstar on "
                       f"{gs} of {N} grid columns + {mi} items, draw_theta on {ns} of {n} respondents; scaled to the full size"),
            "stage_seconds": {k: round(v, 3) for k, v in one.items()},
            "extrapolation_factors": {k: round(v, 1) for k, v in factors.items()},
        },
    }
