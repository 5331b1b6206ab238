"""cpu_baseline leg of bench.py: the CPU restatement of the reference timed on the
GPU box's host cores, on a bounded sample of the same workload.  TEST / MEASUREMENT INFRASTRUCTURE ONLY --
nothing in the product path imports this.

Two legs, both per MCMC iteration of src/gpirtMCMC.cpp:68-78 at (n, m):
  (i)  reference-shaped, ONE thread: the line-following C restatement (oracle/gpirt_oracle.c) with the unblocked
       Cholesky -- the per-item BLAS-2 structure stock R + reference BLAS executes;
  (ii) the same restatement on ALL host cores: blocked Cholesky with OpenMP (oracle/oracle_fast.c, full size, not
       extrapolated) and the item / grid-column / respondent loops of draw_f, draw_fstar, draw_beta, draw_theta
       spread over a thread pool (the C calls release the GIL; items are independent under the item RNG).
Every piece is timed on a sample and scaled linearly in the sampled dimension (cubically for the unblocked
Cholesky, quadratically for K); the factors are returned so the reader can see how far each number is stretched.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O

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
    per_t = 4
    mt = min(m, per_t * nthreads)
    chunks = [slice(a, min(a + per_t, mt)) for a in range(0, mt, per_t)]
    args = [(_F(f[:, c]), _F(y[:, c]), _F(mu[:, c])) for c in chunks]
    with ThreadPoolExecutor(nthreads) as pool:
        allc["draw_f"] = _pool_time(pool, [lambda a=a: _draw_f(lib, O.ItemStream(1), 1, a[0], a[1], L, a[2]) for a in args]) * (m / mt)
        factors["all.items"] = m / mt

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
        g_t = 4                                    # per thread: 4 grid columns, then 4 items
        tsl = [ts[a:a + g_t].copy() for a in range(0, g_t * nthreads, g_t)]
        fa = [(_F(f[:, cc]), _F(beta[0][None, cc] + tsl[i][:, None] * beta[1][None, cc])) for i, cc in enumerate(chunks)]
        fb = [(_F(a[0][:, :1]), _F(a[1][:, :1])) for a in fa]
        tA2 = _pool_time(pool, [lambda i=i: _draw_fstar(lib, O.ItemStream(1), 1, fa[i][0], theta, tsl[i], L, fa[i][1])
                                for i in range(len(fa))])
        tB2 = _pool_time(pool, [lambda i=i: _draw_fstar(lib, O.ItemStream(1), 1, fb[i][0], theta, tsl[i], L, fb[i][1])
                                for i in range(len(fb))])
        p2 = max(tA2 - tB2, 0.0) / max(per_t - 1, 1)         # wall per item per thread, all threads busy
        c2 = max(tB2 - p2, 0.0) / g_t
        allc["draw_fstar"] = c2 * N / nthreads + p2 * m / nthreads
        factors["all.grid_columns"] = N / (g_t * nthreads)

        # ---- draw_theta ---------------------------------------------------------------------------
        fs = _F(fstar)
        ns = min(n, 128)
        ys = _F(y[:ns, :])
        one["draw_theta"] = _timed(lambda: _draw_theta(lib, O.ItemStream(1), 1, ts, prior, ys, fs)) * (n / ns)
        factors["one.respondents"] = n / ns
        r_t = 32
        nt = min(n, r_t * nthreads)
        yb = [_F(y[a:a + r_t, :]) for a in range(0, nt, r_t)]
        allc["draw_theta"] = _pool_time(pool, [lambda b=b: _draw_theta(lib, O.ItemStream(1), 1, ts, prior, b, fs) for b in yb]) * (n / nt)
        factors["all.respondents"] = n / nt

        # ---- draw_beta ----------------------------------------------------------------------------
        pm, ps, st = np.zeros((2, mi), order="F"), np.full((2, mi), 3.0, order="F"), np.full((2, mi), 0.1, order="F")
        bi = _F(beta[:, :mi])
        one["draw_beta"] = _timed(lambda: _draw_beta(lib, O.ItemStream(1), 1, bi, theta, yi, fi, pm, ps, st)) * (m / mi)
        bargs = [(_F(beta[:, cc]), a[1], a[0]) for cc, a in zip(chunks, args)]
        pm4, ps4, st4 = np.zeros((2, per_t), order="F"), np.full((2, per_t), 3.0, order="F"), np.full((2, per_t), 0.1, order="F")
        allc["draw_beta"] = _pool_time(pool, [lambda b=b: _draw_beta(lib, O.ItemStream(1), 1, b[0], theta, b[1], b[2],
                                                                     pm4[:, :b[0].shape[1]], ps4[:, :b[0].shape[1]],
                                                                     st4[:, :b[0].shape[1]]) for b in bargs]) * (m / mt)

    t_one, t_all = sum(one.values()), sum(allc.values())
    return {
        "value": 1.0 / t_one, "unit": "iterations/s", "cores": 1, "kind": "port", "extrapolated": True,
        "sample": (f"C restatement of the reference (oracle/gpirt_oracle.c, unblocked potrf, per-item BLAS-2 structure) on "
                   f"1 thread: K + chol on the leading {n_s} respondents, draw_f / draw_beta on {mi} of {m} items, draw_fstar on "
                   f"{gs} of {N} grid columns + {mi} items, draw_theta on {ns} of {n} respondents; scaled to the full size"),
        "stage_seconds": {k: round(v, 3) for k, v in one.items()},
        "host_cores": nthreads,
        "all_cores": {
            "value": 1.0 / t_all, "unit": "iterations/s", "cores": nthreads, "kind": "port", "extrapolated": True,
            "sample": (f"same restatement on {nthreads} threads: blocked OpenMP potrf at the full n = {n} (not extrapolated), "
                       f"draw_f / draw_beta on {mt} items, draw_fstar on {g_t * nthreads} grid columns + {mt} items, draw_theta on "
                       f"{nt} respondents, {per_t} items / {g_t} columns / {r_t} respondents per thread"),
            "stage_seconds": {k: round(v, 3) for k, v in allc.items()},
        },
        "extrapolation_factors": {k: round(v, 1) for k, v in factors.items()},
    }
