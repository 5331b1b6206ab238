"""Sampler-level host API: the gpirtMCMC() mirror and the stage-driven Sampler.

`gpirtMCMC()` keeps the reference's R signature and returned list (R/gpirtMCMC.R:85-105,
src/gpirtMCMC.cpp:112-116); it prepares the data exactly like the R wrapper and then crosses the
same boundary the R shim would (.Call -> extern "C" gpirt_mcmc, include/gpirt_hip.h).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import NGRID, RNG_ITEM, RNG_RSTREAM, Options, check
from .response_matrix import as_response_matrix

_dp = C.POINTER(C.c_double)


def _f64(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _options(rng, seed, theta_stabilise, fstar_fused, device, item0=0, m_total=0, kernel_fp32=False,
             kstar_rank=0) -> Options:
    o = _lib.default_options()
    o.rng_kind = RNG_RSTREAM if rng == "reference" else RNG_ITEM
    o.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    o.theta_stabilise = int(bool(theta_stabilise))
    o.fstar_fused = int(bool(fstar_fused))
    o.device = -1 if device is None else int(device)
    o.item0 = int(item0)
    o.m_total = int(m_total)
    o.kernel_fp32 = int(bool(kernel_fp32))
    o.kstar_rank = int(kstar_rank)
    return o


def gpirtMCMC(data, sample_iterations, burn_iterations, vote_codes=None, beta_prior_means=None,
              beta_prior_sds=None, beta_proposal_sds=None, theta_init=None, *, rng="reference",
              seed=1, rstream=None, theta_stabilise=False, fstar_fused=False, kstar_rank=0, device=None,
              progress=False, preset=None):
    """Drop-in for the reference's gpirtMCMC() (R/gpirtMCMC.R:85-105) on one MI355X.

    Positional arguments, defaults and the returned dict (theta (S+1) x n, beta 2 x m x (S+1),
    f n x m x (S+1), IRFs 1001 x m) follow the reference.  Keyword-only extras select the RNG
    contract instead of new required arguments:
      rng="reference": replay R's Mersenne-Twister stream (`rstream`, or RStream(seed), stands in
                       for R's .Random.seed); draw-for-draw comparable with the reference;
      rng="item":      counter-based per-item sub-streams keyed by `seed` (batched, shardable).
    fstar_fused / kstar_rank: algebraically identical, cheaper forms of draw_fstar:
      the mean as (L^-1 k*)^T (L^-1 f), and K(theta, theta*) through its exact rank-r Chebyshev factorisation.
    preset="fast": the library's throughput preset, gpirt_fast_options() (R: options(gpirt.hip.preset = "fast")) --
      item-keyed RNG with this call's `seed`, theta_stabilise, fstar_fused, kstar_rank = 64; what bench.py times.
    """
    from .ops import RStream

    lib = _lib.load()
    data = as_response_matrix(data, vote_codes)                      # R/gpirtMCMC.R:93
    y = _f64(np.asarray(data))
    n, m = y.shape
    # defaults are evaluated AFTER the unanimous items were dropped (R lazy evaluation, quirk Q8)
    pm = _f64(np.zeros((2, m)) if beta_prior_means is None else beta_prior_means)
    ps = _f64(np.full((2, m), 3.0) if beta_prior_sds is None else beta_prior_sds)
    st = _f64(np.full((2, m), 0.1) if beta_proposal_sds is None else beta_proposal_sds)
    for a in (pm, ps, st):
        if a.shape != (2, m):
            raise ValueError("beta prior / proposal matrices must be 2 x ncol(data)")
    if preset == "fast":
        rng = "item"
    elif preset is not None:
        raise ValueError(f"unknown preset {preset!r}")
    rs = None
    if rng == "reference":
        rs = rstream if rstream is not None else RStream(seed)
    if theta_init is None:                                           # R/gpirtMCMC.R:95-97
        if rs is not None:
            theta_init = rs.rnorm(n)
        else:
            theta_init = RStream(seed).rnorm(n)
    theta0 = np.ascontiguousarray(theta_init, dtype=np.float64)
    if theta0.shape != (n,):
        raise ValueError("theta_init must have one value per respondent")
    S, B = int(sample_iterations), int(burn_iterations)
    th = np.empty((S + 1, n), order="F")
    be = np.empty((2, m, S + 1), order="F")
    ff = np.empty((n, m, S + 1), order="F")
    irf = np.empty((NGRID, m), order="F")
    if preset == "fast":
        o = _lib.fast_options()
        o.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        o.device = -1 if device is None else int(device)
    else:
        o = _options(rng, seed, theta_stabilise, fstar_fused, device, kstar_rank=kstar_rank)

    def _tick(ctx, it, total):                                       # src/gpirtMCMC.cpp:64-66
        if progress:
            print("\r%6.3f %% complete" % (100.0 * it / max(total, 1)), end="", flush=True)
        return 0

    cb = _lib.TICK_FN(_tick)
    rc = lib.gpirt_mcmc(_ptr(y), n, m, _ptr(theta0), S, B, _ptr(pm), _ptr(ps), _ptr(st), C.byref(o),
                        rs.ptr if rs is not None else None, cb, None, _ptr(th), _ptr(be), _ptr(ff), _ptr(irf))
    if progress:
        print("\r100.000 % complete")
    if rc > 0:
        raise RuntimeError("chol(): decomposition failed")           # what arma::chol throws
    check(rc)
    return dict(theta=th, beta=be, f=ff, IRFs=irf)


class Sampler:
    """Stage-driven sampler (gpirt_sampler_*): device-resident state, one iteration per step().
    preset="fast": the options are gpirt_fast_options() (what bench.py's headline is timed with) with this call's seed,
    item0 / m_total and kernel_fp32; rng, theta_stabilise, fstar_fused and kstar_rank are then the preset's."""

    def __init__(self, handle, y, theta_init, beta_prior_means=None, beta_prior_sds=None,
                 beta_proposal_sds=None, *, rng="item", seed=1, rstream=None, theta_stabilise=True,
                 fstar_fused=False, item0=0, m_total=0, kernel_fp32=False, kstar_rank=0, preset=None):
        self.lib = _lib.load()
        self.handle = handle
        y = _f64(y)
        self.n, self.m = y.shape
        m = self.m
        pm = _f64(np.zeros((2, m)) if beta_prior_means is None else beta_prior_means)
        ps = _f64(np.full((2, m), 3.0) if beta_prior_sds is None else beta_prior_sds)
        st = _f64(np.full((2, m), 0.1) if beta_proposal_sds is None else beta_proposal_sds)
        theta0 = np.ascontiguousarray(theta_init, dtype=np.float64)
        self.rs = rstream
        if preset == "fast":
            o = _lib.fast_options()
            o.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            o.device = -1 if handle.device is None else int(handle.device)
            o.item0, o.m_total, o.kernel_fp32 = int(item0), int(m_total), int(bool(kernel_fp32))
        elif preset is None:
            o = _options(rng, seed, theta_stabilise, fstar_fused, handle.device, item0, m_total, kernel_fp32, kstar_rank)
        else:
            raise ValueError(f"unknown preset {preset!r}")
        s = C.c_void_p()
        check(self.lib.gpirt_sampler_create(C.byref(s), handle.ptr, _ptr(y), self.n, m, _ptr(theta0), _ptr(pm),
                                            _ptr(ps), _ptr(st), C.byref(o),
                                            rstream.ptr if rstream is not None else None))
        self._s = s
        handle._register(self)

    def close(self):
        """Destroy the device state.  Idempotent; Handle.close() closes the samplers still alive on it first, so no order of
        teardown (fixtures, garbage collection at interpreter exit) can leave a sampler draining a freed handle."""
        if getattr(self, "_s", None):
            self.lib.gpirt_sampler_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name):
        rc = getattr(self.lib, name)(self._s)
        if rc > 0:
            raise RuntimeError("chol(): decomposition failed (leading minor %d)" % rc)
        check(rc)

    def init(self): self._call("gpirt_sampler_init")
    def step(self): self._call("gpirt_sampler_step")
    def draw_f(self): self._call("gpirt_sampler_draw_f")
    def draw_fstar(self): self._call("gpirt_sampler_draw_fstar")
    def theta_partial(self): self._call("gpirt_sampler_theta_partial")
    def theta_finish(self): self._call("gpirt_sampler_theta_finish")
    def theta_block(self): self._call("gpirt_sampler_theta_block")
    def theta_commit(self): self._call("gpirt_sampler_theta_commit")

    def set_theta_block(self, y_block, i0: int, m_total: int):
        """This rank's block of respondents with ALL item columns (item-sharded runs, include/gpirt_hip.h)."""
        yb = _f64(np.asfortranarray(y_block))
        check(self.lib.gpirt_sampler_set_theta_block(self._s, _ptr(yb), int(i0), int(yb.shape[0]), int(m_total)))
    def draw_beta(self): self._call("gpirt_sampler_draw_beta")
    def factor(self): self._call("gpirt_sampler_factor")
    def skip_factor(self): self._call("gpirt_sampler_skip_factor")

    def adopt_factor(self, rows_with_L: bool):
        """L arrived from elsewhere; rows_with_L: the whole ldl x n buffer (the bordered rows too) was received."""
        rc = self.lib.gpirt_sampler_adopt_factor(self._s, int(bool(rows_with_L)))
        check(rc)
    def build_cov(self): self._call("gpirt_sampler_build_cov")

    # -- the factorisation in pieces (distributed hosts, include/gpirt_hip.h "gpirt_potrf_panel_*"), on "L" in place
    @property
    def panel_width(self) -> int:
        return int(self.lib.gpirt_potrf_panel_width())

    @property
    def torch_device(self):
        import torch
        return torch.device("cuda", self.handle.device)

    @property
    def panel_rows(self) -> int:
        """rows of L's column blocks that travel with a panel (n, plus the rows of the bordered factorisation)"""
        r = C.c_int64()
        check(self.lib.gpirt_sampler_panel_rows(self._s, C.byref(r)))
        return r.value

    def panel_factor(self, p: int):
        check(self.lib.gpirt_sampler_panel_factor(self._s, int(p)))

    def panel_update(self, p: int, c: int):
        check(self.lib.gpirt_sampler_panel_update(self._s, int(p), int(c)))

    def panel_copy(self, p: int, buf, to_buf: bool):
        """rows [pW, panel_rows) of outer panel p <-> the dense torch buffer `buf` (what the host broadcasts)"""
        check(self.lib.gpirt_sampler_panel_copy(self._s, int(p), C.c_void_p(buf.data_ptr()), int(bool(to_buf))))

    # ... by halves of an outer panel (half: 0 first sub-panel, 1 the rest, 2 whole; part: 0 needs only the first
    # sub-panel, 1 the rest, 2 all): what the pipelined distributed factorisation drives (gpirt_amd/distributed.py)
    @property
    def subpanel_width(self) -> int:
        return int(self.lib.gpirt_potrf_subpanel_width(int(self.n)))

    def panel_factor_part(self, p: int, half: int):
        check(self.lib.gpirt_sampler_panel_factor_part(self._s, int(p), int(half)))

    def panel_update_part(self, p: int, c: int, part: int):
        check(self.lib.gpirt_sampler_panel_update_part(self._s, int(p), int(c), int(part)))

    def panel_copy_part(self, p: int, half: int, buf, to_buf: bool):
        check(self.lib.gpirt_sampler_panel_copy_part(self._s, int(p), int(half), C.c_void_p(buf.data_ptr()), buf.numel(),
                                                     int(bool(to_buf))))

    def streams_busy(self) -> int:
        """bit mask of the handle's internal streams with work in flight (0: every piece has joined the handle stream)"""
        b = C.c_int()
        check(self.lib.gpirt_debug_streams_busy(self.handle.ptr, C.byref(b)))
        return b.value

    def copy_state_from(self, other: "Sampler"):
        check(self.lib.gpirt_sampler_copy_state(self._s, other._s))

    def accumulate_irf(self): self._call("gpirt_sampler_accumulate_irf")
    def check(self): self._call("gpirt_sampler_check")

    @property
    def iteration(self) -> int:
        it = C.c_int()
        check(self.lib.gpirt_sampler_iteration(self._s, C.byref(it)))
        return it.value

    def set_iteration(self, it: int):
        check(self.lib.gpirt_sampler_set_iteration(self._s, int(it)))

    _SHAPES = {"theta": "n", "f": "nm", "beta": "2m", "mu": "nm", "mu_star": "Nm", "fstar": "Nm", "L": "nn",
               "logpost": "Nn", "irf_sum": "Nm", "s": "N", "mean": "Nm", "nu": "nm", "z": "nm", "y": "nm"}

    def _shape(self, name):
        n, m, N = self.n, self.m, NGRID
        return {"n": (n,), "nm": (n, m), "2m": (2, m), "Nm": (N, m), "nn": (n, n), "Nn": (N, n), "N": (N,)}[
            self._SHAPES[name]]

    def get(self, name: str) -> np.ndarray:
        if name == "ess_k":
            out = np.empty(self.m, dtype=np.int32)
            check(self.lib.gpirt_sampler_get(self._s, b"ess_k", C.c_void_p(out.ctypes.data), self.m))
            return out
        if name == "rs_stats":      # the predicted replay's counters (64-bit words; R-stream samplers only)
            out = np.zeros(8, dtype=np.int64)
            check(self.lib.gpirt_sampler_get(self._s, b"rs_stats", C.c_void_p(out.ctypes.data), 8))
            return out
        shape = self._shape(name)
        out = np.empty(shape, order="F")
        check(self.lib.gpirt_sampler_get(self._s, name.encode(), C.c_void_p(out.ctypes.data), out.size))
        return out

    def set(self, name: str, value):
        v = _f64(value)
        check(self.lib.gpirt_sampler_set(self._s, name.encode(), _ptr(v), v.size))

    def devptr(self, name: str):
        p = C.c_void_p()
        cnt = C.c_int64()
        check(self.lib.gpirt_sampler_devptr(self._s, name.encode(), C.byref(p), C.byref(cnt)))
        return p.value, cnt.value

    def device_tensor(self, name: str):
        """Zero-copy torch view of a state array (used to hand buffers to torch.distributed)."""
        import torch

        ptr, cnt = self.devptr(name)

        class _Wrap:
            pass

        w = _Wrap()
        w.__cuda_array_interface__ = {"shape": (cnt,), "typestr": "<f8", "data": (ptr, False), "version": 2}
        return torch.as_tensor(w, device=f"cuda:{self.handle.device}")

    def finish_irfs(self, sample_iterations: int) -> np.ndarray:
        out = np.empty((NGRID, self.m), order="F")
        check(self.lib.gpirt_sampler_finish_irfs(self._s, int(sample_iterations), _ptr(out)))
        return out

    def enable_timing(self, on=True):
        check(self.lib.gpirt_sampler_enable_timing(self._s, int(on)))

    def stage_times(self) -> dict:
        ms = (C.c_double * 16)()
        k = C.c_int()
        check(self.lib.gpirt_sampler_stage_times(self._s, ms, 16, C.byref(k), None))
        names = ["draw_f", "draw_fstar", "theta_gemm", "theta_sample", "draw_beta", "factor"]
        return {names[i]: ms[i] for i in range(k.value)}
