"""ctypes front-end of the CPU oracle (oracle/libgpirt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under gpirt_amd/.  "parity unpinned" by reference golden
vectors (the reference ships none and cannot run here); see oracle/gpirt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgpirt_oracle.so")
NGRID = 1001

ST_INIT_F, ST_INIT_BETA, ST_F_Z, ST_F_ESS, ST_FSTAR, ST_THETA, ST_BETA = 1, 2, 3, 4, 5, 6, 7


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Rng(C.Structure):
    _fields_ = [
        ("kind", C.c_int),
        ("mt", C.c_uint32 * 624),
        ("mti", C.c_int),
        ("n_unif", C.c_uint64),
        ("seed", C.c_uint64),
        ("iter", C.c_uint32),
        ("stage", C.c_uint32),
        ("item", C.c_uint32),
        ("index", C.c_uint32),
        ("item_base", C.c_uint32),
        ("item_local", C.c_uint32),
    ]


class EssTrace(C.Structure):
    _fields_ = [("u", C.c_double), ("log_y", C.c_double), ("eps0", C.c_double),
                ("eps_final", C.c_double), ("k", C.c_int)]


class McmcOpts(C.Structure):
    _fields_ = [("theta_stabilise", C.c_int), ("blocked_potrf", C.c_int),
                ("nthreads", C.c_int), ("fstar_fused", C.c_int)]


_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_unif_rand.restype = C.c_double
        L.orc_norm_rand.restype = C.c_double
        L.orc_rnorm.restype = C.c_double
        L.orc_rnorm.argtypes = [C.POINTER(Rng), C.c_double, C.c_double]
        L.orc_runif.restype = C.c_double
        L.orc_runif.argtypes = [C.POINTER(Rng), C.c_double, C.c_double]
        L.orc_qnorm.restype = C.c_double
        L.orc_qnorm.argtypes = [C.c_double]
        L.orc_dnorm_log.restype = C.c_double
        L.orc_dnorm_log.argtypes = [C.c_double] * 3
        L.orc_plogis.restype = C.c_double
        L.orc_plogis.argtypes = [C.c_double]
        L.orc_item_uniform.restype = C.c_double
        L.orc_item_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_init_rstream.argtypes = [C.POINTER(Rng), C.c_uint32]
        L.orc_rng_init_item.argtypes = [C.POINTER(Rng), C.c_uint64]
        L.orc_rng_substream.argtypes = [C.POINTER(Rng), C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_ll.restype = C.c_double
        L.orc_ll_bar.restype = C.c_double
        L.orc_potrf_lower.restype = C.c_int
        L.orc_potrf_lower_blocked.restype = C.c_int
        L.orc_ess.restype = C.c_int
        L.orc_draw_theta.restype = C.c_int
        L.orc_gpirt_mcmc.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_dp)


def _f(a):
    """column-major float64 copy"""
    return np.asfortranarray(np.array(a, dtype=np.float64))


class RStream:
    """R's Mersenne-Twister + inversion stream; RStream(seed) == set.seed(seed)."""

    def __init__(self, seed: int):
        self.s = Rng()
        lib().orc_rng_init_rstream(C.byref(self.s), C.c_uint32(seed & 0xFFFFFFFF))

    @property
    def ref(self):
        return C.byref(self.s)

    @property
    def n_unif(self):
        return int(self.s.n_unif)

    def runif(self, n, a=0.0, b=1.0):
        return np.array([lib().orc_runif(self.ref, a, b) for _ in range(n)])

    def rnorm(self, n, mu=0.0, sd=1.0):
        return np.array([lib().orc_rnorm(self.ref, mu, sd) for _ in range(n)])

    def mt_state(self):
        return np.array(self.s.mt, dtype=np.uint32), int(self.s.mti)


class ItemStream:
    """Counter-based (Philox4x32-10) per-(iteration, stage, item) sub-streams."""

    def __init__(self, seed: int, item_base: int = 0):
        self.s = Rng()
        lib().orc_rng_init_item(C.byref(self.s), C.c_uint64(seed))
        self.s.item_base = item_base

    @property
    def ref(self):
        return C.byref(self.s)

    @property
    def n_unif(self):
        return int(self.s.n_unif)

    def substream(self, it, stage, item):
        lib().orc_rng_substream(self.ref, it, stage, item)


def item_uniform(seed, it, stage, item, index):
    return lib().orc_item_uniform(seed, it, stage, item, index)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return list(o)


def qnorm(p):
    return lib().orc_qnorm(float(p))


def theta_star():
    g = np.empty(NGRID)
    lib().orc_theta_star_grid(_p(g))
    return g


def se_kernel(x1, x2):
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    x2 = np.ascontiguousarray(x2, dtype=np.float64)
    out = np.empty((len(x1), len(x2)), order="F")
    lib().orc_se_kernel(_p(x1), C.c_int64(len(x1)), _p(x2), C.c_int64(len(x2)), _p(out))
    return out


def potrf_lower(S, blocked=False, nthreads=0):
    A = _f(S)
    n = A.shape[0]
    if blocked:
        info = lib().orc_potrf_lower_blocked(_p(A), C.c_int64(n), C.c_int(nthreads))
    else:
        info = lib().orc_potrf_lower(_p(A), C.c_int64(n))
    return A, info


def factor(theta, jitter=0.001, blocked=False, nthreads=0):
    S = se_kernel(theta, theta)
    S[np.diag_indices_from(S)] += jitter
    return potrf_lower(S, blocked, nthreads)


def rmvnorm(rng, L):
    L = _f(L)
    n = L.shape[0]
    z = np.empty(n)
    out = np.empty(n)
    lib().orc_rmvnorm(rng.ref, _p(L), C.c_int64(n), _p(z), _p(out))
    return out, z


def ll(f, y):
    f = np.ascontiguousarray(f, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().orc_ll(_p(f), _p(y), C.c_int64(len(f)))


def ll_bar(f, y, mu):
    f = np.ascontiguousarray(f, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    return lib().orc_ll_bar(_p(f), _p(y), _p(mu), C.c_int64(len(f)))


def ess(rng, f, y, L, mu, it=0, item=0):
    f, y, mu = (np.ascontiguousarray(a, dtype=np.float64) for a in (f, y, mu))
    L = _f(L)
    n = len(f)
    out = np.empty(n)
    nu = np.empty(n)
    tr = EssTrace()
    rng.s.iter, rng.s.item_local = it, item
    lib().orc_ess(rng.ref, _p(f), _p(y), _p(L), _p(mu), C.c_int64(n), _p(out), _p(nu), C.byref(tr))
    return out, nu, dict(u=tr.u, log_y=tr.log_y, eps0=tr.eps0, eps_final=tr.eps_final, k=tr.k)


def draw_f(rng, f, y, L, mu, it=1):
    f, y, L, mu = _f(f), _f(y), _f(L), _f(mu)
    n, m = f.shape
    out = np.empty((n, m), order="F")
    k = np.zeros(m, dtype=np.int32)
    lib().orc_draw_f(rng.ref, C.c_uint32(it), _p(f), _p(y), _p(L), _p(mu), C.c_int64(n),
                     C.c_int64(m), _p(out), k.ctypes.data_as(_ip))
    return out, k


def trsm_lower(L, B, trans=False):
    L = _f(L)
    B = _f(B)
    if B.ndim == 1:
        B = B.reshape(-1, 1, order="F")
    lib().orc_trsm_lower(_p(L), C.c_int64(L.shape[0]), _p(B), C.c_int64(B.shape[1]), C.c_int(int(trans)))
    return B


def draw_fstar(rng, f, theta, L, mu_star, it=1, tstar=None):
    f, L, mu_star = _f(f), _f(L), _f(mu_star)
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    ts = theta_star() if tstar is None else np.ascontiguousarray(tstar, dtype=np.float64)
    n, m = f.shape
    N = len(ts)
    out = np.empty((N, m), order="F")
    s = np.empty(N)
    mean = np.empty((N, m), order="F")
    lib().orc_draw_fstar(rng.ref, C.c_uint32(it), _p(f), _p(theta), _p(ts), _p(L), _p(mu_star),
                         C.c_int64(n), C.c_int64(m), C.c_int64(N), _p(out), _p(s), _p(mean))
    return out, s, mean


def draw_theta(rng, y, fstar, it=1, stabilise=False):
    y, fstar = _f(y), _f(fstar)
    n, m = y.shape
    ts = theta_star()
    N = len(ts)
    prior = np.array([lib().orc_dnorm_log(t, 0.0, 1.0) for t in ts])
    out = np.empty(n)
    deg = lib().orc_draw_theta(rng.ref, C.c_uint32(it), _p(ts), _p(y), _p(prior), _p(fstar),
                               C.c_int64(n), C.c_int64(m), C.c_int64(N), C.c_int(int(stabilise)), _p(out))
    return out, deg


def draw_beta(rng, beta, theta, y, f, pm, ps, step, it=1):
    beta, y, f, pm, ps, step = (_f(a) for a in (beta, y, f, pm, ps, step))
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    n, m = y.shape
    out = np.empty((2, m), order="F")
    lib().orc_draw_beta(rng.ref, C.c_uint32(it), _p(beta), _p(theta), _p(y), _p(f), _p(pm), _p(ps),
                        _p(step), C.c_int64(n), C.c_int64(m), _p(out))
    return out


def gpirt_mcmc(rng, y, theta0, sample_iterations, burn_iterations, pm=None, ps=None, step=None,
               theta_stabilise=False, blocked_potrf=False, nthreads=0, fstar_fused=False,
               want_state=False):
    """orc_gpirt_mcmc: returns dict(theta, beta, f, IRFs[, L, fstar]) shaped like the R list."""
    y = _f(y)
    n, m = y.shape
    theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
    pm = _f(np.zeros((2, m)) if pm is None else pm)
    ps = _f(np.full((2, m), 3.0) if ps is None else ps)
    step = _f(np.full((2, m), 0.1) if step is None else step)
    S = int(sample_iterations)
    th = np.empty((S + 1, n), order="F")
    be = np.empty((2, m, S + 1), order="F")
    ff = np.empty((n, m, S + 1), order="F")
    irf = np.empty((NGRID, m), order="F")
    Lf = np.empty((n, n), order="F") if want_state else None
    fs = np.empty((NGRID, m), order="F") if want_state else None
    o = McmcOpts(int(theta_stabilise), int(blocked_potrf), int(nthreads), int(fstar_fused))
    info = lib().orc_gpirt_mcmc(rng.ref, _p(y), C.c_int64(n), C.c_int64(m), _p(theta0), C.c_int(S),
                                C.c_int(int(burn_iterations)), _p(pm), _p(ps), _p(step), C.byref(o),
                                _p(th), _p(be), _p(ff), _p(irf),
                                _p(Lf) if want_state else None, _p(fs) if want_state else None)
    if info:
        raise RuntimeError(f"chol(): decomposition failed (leading minor {info})")
    res = dict(theta=th, beta=be, f=ff, IRFs=irf)
    if want_state:
        res["L"] = Lf
        res["fstar"] = fs
    return res
