"""Operator-level host API over the C ABI (device tensors in, device tensors out).

Mirrors the reference's internal operator interface, src/gpirt.h:4-28 -- K(), draw_f(),
draw_fstar(), draw_theta(), draw_beta(), ll_bar() -- plus the LAPACK/BLAS-level pieces the
reference reaches through Armadillo (chol, solve(trimatl/trimatu), cholS * z).  torch is used only
for device memory and the stream; all arithmetic happens in libgpirt_hip.so.

Matrices are COLUMN-MAJOR fp64 (Armadillo / R layout): an n x m matrix is a torch tensor of shape
(n, m) with strides (1, ld).  Use `colmajor()` / `to_device()` to make them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import NGRID, check


def colmajor(n: int, m: int, device="cuda", fill=None) -> torch.Tensor:
    """Uninitialised (or filled) column-major n x m fp64 device matrix."""
    base = torch.empty((m, n), dtype=torch.float64, device=device)
    if fill is not None:
        base.fill_(fill)
    return base.T


def to_device(a, device="cuda") -> torch.Tensor:
    """Host array -> column-major device tensor (vectors stay 1-d)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if a.ndim != 2:
        raise ValueError("expected a vector or a matrix")
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device).T


def to_host(t: torch.Tensor) -> np.ndarray:
    """Device tensor -> Fortran-ordered numpy array."""
    if t.dim() == 2:
        return np.asfortranarray(t.T.contiguous().cpu().numpy().T)
    return t.cpu().numpy()


def _ld(t: torch.Tensor) -> int:
    if t.dtype != torch.float64 or not t.is_cuda:
        raise TypeError("expected a float64 device tensor")
    if t.dim() == 1:
        if t.stride(0) != 1:
            raise ValueError("vector must be contiguous")
        return t.shape[0]
    if t.dim() != 2 or (t.shape[0] > 1 and t.stride(0) != 1):
        raise ValueError("matrix must be column-major (stride(0) == 1)")
    ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[0], 1)
    if ld < t.shape[0]:
        raise ValueError("bad leading dimension")
    return ld


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Handle:
    """gpirt_handle_t bound to a device and (by default) torch's current stream."""

    def __init__(self, device: int | None = None, stream: torch.cuda.Stream | None = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GpirtError(_lib.E_NODEVICE, "no GPU visible: the HIP path has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.device = device
        with torch.cuda.device(device):
            self.stream = stream if stream is not None else torch.cuda.current_stream()
            h = C.c_void_p()
            check(self.lib.gpirt_create(C.byref(h), device, C.c_void_p(self.stream.cuda_stream)))
        self._h = h
        self._samplers = []                # weak references to the samplers created on this handle

    @property
    def ptr(self):
        return self._h

    def _register(self, sampler):
        import weakref
        self._samplers = [r for r in self._samplers if r() is not None]
        self._samplers.append(weakref.ref(sampler))

    def close(self):
        if getattr(self, "_h", None):
            for r in getattr(self, "_samplers", []):      # samplers first: they drain this handle's streams when they go
                s = r()
                if s is not None:
                    s.close()
            self._samplers = []
            self.lib.gpirt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.gpirt_synchronize(self._h))

    # ---------------------------------------------------------------- switches / hang-guard fallback
    def config_get(self, name: str) -> int:
        v = C.c_int()
        check(self.lib.gpirt_config_get(self._h, name.encode(), C.byref(v)))
        return v.value

    def config_set(self, name: str, value: int):
        """Set one of the library's switches (README.md) for this handle; the environment is only read at process start."""
        check(self.lib.gpirt_config_set(self._h, name.encode(), int(value)))

    class _Override:
        def __init__(self, h, name, value):
            self.h, self.name, self.value = h, name, value

        def __enter__(self):
            self.old = self.h.config_get(self.name)
            self.h.config_set(self.name, self.value)
            return self.h

        def __exit__(self, *exc):
            self.h.config_set(self.name, self.old)

    def config(self, name: str, value: int):
        """`with handle.config("GPIRT_TRSM_INV", 2): ...` -- the switch for the duration of the block."""
        return Handle._Override(self, name, value)

    @property
    def guard_fallbacks(self) -> int:
        v = C.c_int()
        check(self.lib.gpirt_guard_fallbacks(self._h, C.byref(v)))
        return v.value

    def debug_trip_guard(self, nth: int = 1):
        """Debug: the nth factorisation from now ends as a hang-guard expiry would leave it (include/gpirt_hip.h)."""
        check(self.lib.gpirt_debug_trip_guard(self._h, int(nth)))

    def calibrate_mfma_f64(self) -> float:
        v = C.c_double()
        check(self.lib.gpirt_calibrate_mfma_f64(self._h, C.byref(v)))
        return v.value

    # ---------------------------------------------------------------- operators
    def se_kernel(self, x1: torch.Tensor, x2: torch.Tensor, jitter: float = 0.0) -> torch.Tensor:
        """K(x1, x2): src/covariance-function.cpp:3-14 (+ jitter on i == j)."""
        out = colmajor(x1.shape[0], x2.shape[0], x1.device)
        check(self.lib.gpirt_se_kernel(self._h, _p(x1), x1.shape[0], _p(x2), x2.shape[0], _p(out), _ld(out), jitter))
        return out

    def potrf_lower(self, S: torch.Tensor) -> torch.Tensor:
        """arma::chol(S, "lower") in place; raises like the reference when S is not PD."""
        n = S.shape[0]
        info = self.lib.gpirt_potrf_lower(self._h, _p(S), n, _ld(S))
        if info > 0:
            raise RuntimeError("chol(): decomposition failed (leading minor %d)" % info)
        check(info)
        return S

    # -- the same factorisation in pieces (include/gpirt_hip.h "gpirt_potrf_panel_*"): what a distributing host calls
    @property
    def panel_width(self) -> int:
        return int(self.lib.gpirt_potrf_panel_width())

    def potrf_begin(self):
        check(self.lib.gpirt_potrf_begin(self._h))

    def potrf_panel_factor(self, A: torch.Tensor, p: int):
        check(self.lib.gpirt_potrf_panel_factor(self._h, _p(A), A.shape[0], _ld(A), int(p)))

    def potrf_panel_update(self, A: torch.Tensor, p: int, c: int):
        check(self.lib.gpirt_potrf_panel_update(self._h, _p(A), A.shape[0], _ld(A), int(p), int(c)))

    def potrf_panel_copy(self, A: torch.Tensor, p: int, buf: torch.Tensor, to_buf: bool):
        check(self.lib.gpirt_potrf_panel_copy(self._h, _p(A), A.shape[0], _ld(A), int(p), _p(buf), int(bool(to_buf))))

    # ... by halves of an outer panel (half: 0 first sub-panel, 1 the rest, 2 whole; part: 0 needs only the first sub-panel)
    def subpanel_width(self, n: int) -> int:
        """first sub-panel of an outer panel of an n x n factorisation (GPIRT_NBP, or by size: csrc/potrf.hip)"""
        return int(self.lib.gpirt_potrf_subpanel_width(int(n)))

    def potrf_panel_factor_part(self, A: torch.Tensor, p: int, half: int):
        check(self.lib.gpirt_potrf_panel_factor_part(self._h, _p(A), A.shape[0], _ld(A), int(p), int(half)))

    def potrf_panel_update_part(self, A: torch.Tensor, p: int, c: int, part: int):
        check(self.lib.gpirt_potrf_panel_update_part(self._h, _p(A), A.shape[0], _ld(A), int(p), int(c), int(part)))

    def potrf_panel_copy_part(self, A: torch.Tensor, p: int, half: int, buf: torch.Tensor, to_buf: bool):
        check(self.lib.gpirt_potrf_panel_copy_part(self._h, _p(A), A.shape[0], _ld(A), int(p), int(half), _p(buf), buf.numel(),
                                                   int(bool(to_buf))))

    def potrf_finish(self):
        info = self.lib.gpirt_potrf_finish(self._h)
        if info > 0:
            raise RuntimeError("chol(): decomposition failed (leading minor %d)" % info)
        check(info)

    def factor(self, theta: torch.Tensor) -> torch.Tensor:
        """K(theta,theta) + 0.001 I -> lower Cholesky factor (src/gpirtMCMC.cpp:15-17)."""
        n = theta.shape[0]
        L = colmajor(n, n, theta.device)
        info = self.lib.gpirt_factor(self._h, _p(theta), n, _p(L), _ld(L))
        if info > 0:
            raise RuntimeError("chol(): decomposition failed (leading minor %d)" % info)
        check(info)
        return L

    def trmm_lz(self, L: torch.Tensor, Z: torch.Tensor) -> torch.Tensor:
        """rmvnorm()'s cholS * res for all columns (src/mvnormal.h:10)."""
        n, m = Z.shape
        out = colmajor(n, m, Z.device)
        check(self.lib.gpirt_trmm_lz(self._h, _p(L), n, _ld(L), _p(Z), m, _ld(Z), _p(out), _ld(out)))
        return out

    def trsm_lower(self, L: torch.Tensor, B: torch.Tensor, trans: bool = False) -> torch.Tensor:
        """solve(trimatl(L), B) / solve(trimatu(L.t()), B) in place (src/draw-fstar.cpp:7,19)."""
        n, nrhs = B.shape
        check(self.lib.gpirt_trsm_lower(self._h, _p(L), n, _ld(L), _p(B), nrhs, _ld(B), int(trans)))
        return B

    def gemm(self, A, B, ta=False, tb=False, alpha=1.0, beta=0.0, C_out=None):
        M = A.shape[1] if ta else A.shape[0]
        K = A.shape[0] if ta else A.shape[1]
        N = B.shape[0] if tb else B.shape[1]
        if C_out is None:
            C_out = colmajor(M, N, A.device, fill=0.0 if beta != 0.0 else None)
        check(self.lib.gpirt_gemm(self._h, int(ta), int(tb), M, N, K, alpha, _p(A), _ld(A), _p(B), _ld(B),
                                  beta, _p(C_out), _ld(C_out)))
        return C_out

    def ll_bar(self, f, y, mu=None) -> torch.Tensor:
        """ll_bar() / ll() per column: src/log-likelihood.cpp:12-37."""
        n, m = f.shape
        out = torch.empty(m, dtype=torch.float64, device=f.device)
        check(self.lib.gpirt_ll_bar(self._h, _p(f), _p(y), _p(mu), n, m, _p(out)))
        return out

    def ll_term(self, a, fast=True, screen=False) -> torch.Tensor:
        """log(1 + exp(-a)) elementwise: the slice kernel's form (csrc/ll_fast.h) or, fast=False, the formula as written;
        screen=True: the single-precision screen of the slice loop's accept test (ll_term_screen, same header)."""
        a = a.contiguous().to(torch.float64)
        out = torch.empty_like(a)
        check(self.lib.gpirt_debug_ll_term(self._h, _p(a), a.numel(), _p(out), 2 if screen else int(bool(fast))))
        return out

    def item_normals(self, seed, it, stage, item0, n_items, n_index) -> torch.Tensor:
        out = colmajor(n_index, n_items)
        check(self.lib.gpirt_item_normals(self._h, seed, it, stage, item0, n_items, n_index, _p(out)))
        return out

    def item_uniforms(self, seed, it, stage, item0, n_items, n_index) -> torch.Tensor:
        out = colmajor(n_index, n_items)
        check(self.lib.gpirt_item_uniforms(self._h, seed, it, stage, item0, n_items, n_index, _p(out)))
        return out

    def draw_f(self, f, y, L, mu, seed: int, it: int):
        """draw_f(): src/draw-f.cpp:64-73 (item RNG).  In place on f; returns (f, rejections)."""
        n, m = f.shape
        k = torch.zeros(m, dtype=torch.int32, device=f.device)
        check(self.lib.gpirt_draw_f(self._h, _p(f), _p(y), _p(L), _ld(L), _p(mu), n, m, seed, it, _p(k)))
        return f, k

    def draw_fstar(self, f, theta, L, mu_star, seed: int, it: int, fused: bool = False):
        """draw_fstar(): src/draw-fstar.cpp:10-31 (item RNG).  Returns (fstar, s, mean)."""
        n, m = f.shape
        out = colmajor(NGRID, m, f.device)
        s = torch.empty(NGRID, dtype=torch.float64, device=f.device)
        mean = colmajor(NGRID, m, f.device)
        check(self.lib.gpirt_draw_fstar(self._h, _p(f), _p(theta), _p(L), _ld(L), _p(mu_star), n, m, seed, it,
                                        int(fused), _p(out), _p(s), _p(mean)))
        return out, s, mean

    def draw_theta(self, y, fstar, seed: int, it: int, stabilise: bool = True):
        """draw_theta(): src/draw-theta.cpp:3-37 (item RNG).  Returns (theta, n_degenerate)."""
        n, m = y.shape
        out = torch.empty(n, dtype=torch.float64, device=y.device)
        deg = torch.zeros(1, dtype=torch.int32, device=y.device)
        check(self.lib.gpirt_draw_theta(self._h, _p(y), _p(fstar), n, m, seed, it, int(stabilise), _p(out), _p(deg)))
        return out, int(deg.item())

    def theta_logpost(self, y, fstar):
        """The log-posterior of draw_theta before the prior (NGRID x n): src/draw-theta.cpp:15-19 as the product the sampler
        forms (csrc/theta_fixed.hip, or the fp64 GEMM under GPIRT_THETA_FIXED=2).  Returns (logpost, fell_back)."""
        n, m = y.shape
        out = colmajor(NGRID, n, y.device)
        fb = C.c_int(0)
        check(self.lib.gpirt_debug_theta_logpost(self._h, _p(y), _p(fstar), n, m, _p(out), C.byref(fb)))
        return out, fb.value

    def draw_beta(self, beta, theta, y, f, pm, ps, step, seed: int, it: int):
        """draw_beta(): src/draw-beta.cpp:3-41 (item RNG).  In place on beta (2 x m)."""
        n, m = y.shape
        check(self.lib.gpirt_draw_beta(self._h, _p(beta), _p(theta), _p(y), _p(f), _p(pm), _p(ps), _p(step),
                                       n, m, seed, it))
        return beta

    # ---------------------------------------------------------------- profiling
    def prof_enable(self, on: bool):
        check(self.lib.gpirt_prof_enable(self._h, int(on)))

    def prof_trailing(self, reset: bool = False):
        ms = C.c_double()
        n = C.c_int64()
        fl = C.c_double()
        check(self.lib.gpirt_prof_trailing(self._h, int(reset), C.byref(ms), C.byref(n), C.byref(fl)))
        return ms.value, n.value, fl.value

    SYRK_CLASSES = ("trailing_128tile", "trailing_64tile", "in_panel_update")

    def prof_syrk(self, reset: bool = False) -> dict:
        """Event-timed syrk launches of the factorisation per class:
        {name: (ms, launches, algorithmic flops, algorithmic bytes)}."""
        out = {}
        for cls, name in enumerate(self.SYRK_CLASSES):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            check(self.lib.gpirt_prof_syrk_bytes(self._h, cls, C.byref(by)))
            check(self.lib.gpirt_prof_syrk(self._h, cls, int(reset), C.byref(ms), C.byref(n), C.byref(fl)))
            out[name] = (ms.value, n.value, fl.value, by.value)
        return out

    OTHER_CLASSES = {3: "draw_f_trmm", 4: "replay_products", 5: "theta_int8_product"}

    def prof_other(self, reset: bool = False) -> dict:
        """The same instrument on two kernels of draw_f and on draw_theta's product (include/gpirt_hip.h, gpirt_prof_syrk
        classes 3, 4 and 5):
        {name: (ms, launches, algorithmic flops, algorithmic bytes)}."""
        out = {}
        for cls, name in self.OTHER_CLASSES.items():
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            check(self.lib.gpirt_prof_syrk_bytes(self._h, cls, C.byref(by)))
            check(self.lib.gpirt_prof_syrk(self._h, cls, int(reset), C.byref(ms), C.byref(n), C.byref(fl)))
            out[name] = (ms.value, n.value, fl.value, by.value)
        return out


class RStream:
    """R's default RNG on the host: RStream(seed) == set.seed(seed); .rnorm(n) == rnorm(n)."""

    def __init__(self, seed: int | None = None, state=None):
        self.lib = _lib.load()
        r = C.c_void_p()
        if state is not None:
            mt, mti = state
            arr = (C.c_uint32 * 624)(*[int(x) for x in mt])
            check(self.lib.gpirt_rstream_from_state(C.byref(r), arr, int(mti)))
        else:
            check(self.lib.gpirt_rstream_create(C.byref(r), C.c_uint32(int(seed) & 0xFFFFFFFF)))
        self._r = r

    @property
    def ptr(self):
        return self._r

    def runif(self, n: int) -> np.ndarray:
        out = np.empty(n)
        check(self.lib.gpirt_rstream_unif(self._r, out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def rnorm(self, n: int) -> np.ndarray:
        out = np.empty(n)
        check(self.lib.gpirt_rstream_norm(self._r, out.ctypes.data_as(C.POINTER(C.c_double)), n))
        return out

    def state(self):
        mt = (C.c_uint32 * 624)()
        mti = C.c_int()
        check(self.lib.gpirt_rstream_get_state(self._r, mt, C.byref(mti)))
        return np.array(mt, dtype=np.uint32), mti.value

    def __del__(self):
        try:
            if self._r:
                self.lib.gpirt_rstream_destroy(self._r)
                self._r = None
        except Exception:
            pass
