"""The CPU oracle driven on all host cores (TEST / MEASUREMENT INFRASTRUCTURE ONLY: the metric-size parity test and the
all-core leg of bench.py's cpu_baseline; nothing in the product path imports this).

The stage functions of oracle/gpirt_oracle.c are sequential loops over items (draw_f, draw_fstar :23-29, draw_beta),
grid columns (draw_fstar :17-20) or respondents (draw_theta).  Under the item RNG every iteration of those loops has
its own sub-stream keyed by the GLOBAL item / respondent index, so slices of a loop reproduce the whole loop's draws
bit for bit; the C calls release the GIL, so a thread pool runs the slices side by side.  This is what lets the
metric-size parity test (8192 x 1024, tests/test_gpu_metric_oracle.py) compare EVERY draw of an iteration with the
restatement of src/draw-f.cpp:21-73, src/draw-fstar.cpp:10-31, src/draw-theta.cpp:3-37, src/draw-beta.cpp:3-41 in
about a minute.  tests/test_oracle_parallel.py pins the slices against the sequential functions.
"""
from __future__ import annotations

import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import oracle as O

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def host_cores() -> int:
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


def _slices(total, nthreads, per=None):
    per = per or max(1, -(-total // (4 * nthreads)))
    return [slice(a, min(a + per, total)) for a in range(0, total, per)]


def _run(nthreads, jobs):
    with ThreadPoolExecutor(nthreads) as pool:
        return list(pool.map(lambda j: j(), jobs))


def factor(theta, nthreads=None):
    """K + jitter + chol (src/gpirtMCMC.cpp:76-78): scalar K, blocked OpenMP potrf."""
    return O.factor(np.ascontiguousarray(theta, dtype=np.float64), blocked=True, nthreads=nthreads or host_cores())


def draw_f(seed, it, f, y, L, mu, nthreads=None, item0=0):
    """src/draw-f.cpp:64-73 over item slices.  Returns (f_new, rejection counts)."""
    lib = O.lib()
    nthreads = nthreads or host_cores()
    L = np.asfortranarray(L)
    n, m = f.shape
    out = np.empty((n, m), order="F")
    kk = np.zeros(m, dtype=np.int32)

    def job(sl):
        fi, yi, mui = _F(f[:, sl]), _F(y[:, sl]), _F(mu[:, sl])
        oi = np.empty(fi.shape, order="F")
        ki = np.zeros(fi.shape[1], dtype=np.int32)
        rng = O.ItemStream(seed, item_base=item0 + sl.start)
        lib.orc_draw_f(rng.ref, C.c_uint32(it), _p(fi), _p(yi), _p(L), _p(mui), C.c_int64(n), C.c_int64(fi.shape[1]),
                       _p(oi), ki.ctypes.data_as(_ip))
        out[:, sl] = oi
        kk[sl] = ki

    _run(nthreads, [lambda sl=sl: job(sl) for sl in _slices(m, nthreads)])
    return out, kk


def draw_fstar(seed, it, f, theta, L, mu_star, nthreads=None, item0=0):
    """src/draw-fstar.cpp:10-31: :17-20 over grid-column slices, then :23-29 over item slices.
    Returns (fstar, s, mean) with mean INCLUDING mu_star (as the reference's :25)."""
    lib = O.lib()
    nthreads = nthreads or host_cores()
    L = np.asfortranarray(L)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    ts = O.theta_star()
    n, m = f.shape
    N = len(ts)
    kstar = np.empty((n, N), order="F")
    s = np.empty(N)

    def grid(sl):
        tsl = np.ascontiguousarray(ts[sl])
        k = np.empty((n, len(tsl)), order="F")
        si = np.empty(len(tsl))
        lib.orc_fstar_grid(_p(theta), _p(tsl), _p(L), C.c_int64(n), C.c_int64(len(tsl)), _p(k), None, _p(si))
        kstar[:, sl] = k
        s[sl] = si

    _run(nthreads, [lambda sl=sl: grid(sl) for sl in _slices(N, nthreads)])
    out = np.empty((N, m), order="F")
    mean = np.empty((N, m), order="F")

    def items(sl):
        fi, msi = _F(f[:, sl]), _F(mu_star[:, sl])
        oi = np.empty(msi.shape, order="F")
        mi = np.empty(msi.shape, order="F")
        rng = O.ItemStream(seed, item_base=item0 + sl.start)
        lib.orc_fstar_items(rng.ref, C.c_uint32(it), _p(fi), _p(kstar), _p(s), _p(L), _p(msi), C.c_int64(n),
                            C.c_int64(fi.shape[1]), C.c_int64(N), _p(oi), _p(mi))
        out[:, sl] = oi
        mean[:, sl] = mi

    _run(nthreads, [lambda sl=sl: items(sl) for sl in _slices(m, nthreads)])
    return out, s, mean


def draw_theta(seed, it, y, fstar, stabilise=True, nthreads=None):
    """src/draw-theta.cpp:3-37 over respondent blocks.  Returns (theta, degenerate count)."""
    lib = O.lib()
    lib.orc_draw_theta_block.restype = C.c_int
    nthreads = nthreads or host_cores()
    ts = O.theta_star()
    N = len(ts)
    prior = np.array([lib.orc_dnorm_log(t, 0.0, 1.0) for t in ts])
    fs = _F(fstar)
    n, m = y.shape
    out = np.empty(n)
    deg = []

    def job(sl):
        yb = _F(y[sl, :])
        ob = np.empty(yb.shape[0])
        rng = O.ItemStream(seed)
        d = lib.orc_draw_theta_block(rng.ref, C.c_uint32(it), _p(ts), _p(yb), _p(prior), _p(fs), C.c_int64(yb.shape[0]),
                                     C.c_int64(m), C.c_int64(N), C.c_int(int(stabilise)), C.c_int64(sl.start), _p(ob))
        out[sl] = ob
        deg.append(d)

    _run(nthreads, [lambda sl=sl: job(sl) for sl in _slices(n, nthreads)])
    return out, int(sum(deg))


def draw_beta(seed, it, beta, theta, y, f, pm, ps, step, nthreads=None, item0=0):
    """src/draw-beta.cpp:3-41 over item slices."""
    lib = O.lib()
    nthreads = nthreads or host_cores()
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    n, m = y.shape
    out = np.empty((2, m), order="F")

    def job(sl):
        b, yi, fi = _F(beta[:, sl]), _F(y[:, sl]), _F(f[:, sl])
        a, c, d = _F(pm[:, sl]), _F(ps[:, sl]), _F(step[:, sl])
        ob = np.empty(b.shape, order="F")
        rng = O.ItemStream(seed, item_base=item0 + sl.start)
        lib.orc_draw_beta(rng.ref, C.c_uint32(it), _p(b), _p(theta), _p(yi), _p(fi), _p(a), _p(c), _p(d), C.c_int64(n),
                          C.c_int64(b.shape[1]), _p(ob))
        out[:, sl] = ob

    _run(nthreads, [lambda sl=sl: job(sl) for sl in _slices(m, nthreads)])
    return out
