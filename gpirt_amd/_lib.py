"""ctypes binding of libgpirt_hip.so -- the C ABI declared in include/gpirt_hip.h.

The product path has NO CPU fallback: if the library is missing, or no gfx950 device is visible,
every compute entry raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GPIRT_HIP_LIBRARY: test hook -- another build of the same library (tests/test_gpu_fences.py loads the fenced variant)
LIB_PATH = os.environ.get("GPIRT_HIP_LIBRARY") or os.path.join(HERE, "libgpirt_hip.so")

NGRID = 1001
RNG_RSTREAM, RNG_ITEM = 0, 1
ST_INIT_F, ST_INIT_BETA, ST_F_Z, ST_F_ESS, ST_FSTAR, ST_THETA, ST_BETA = 1, 2, 3, 4, 5, 6, 7

E_ARG, E_HIP, E_NODEVICE, E_ALLOC, E_RNG, E_INTERRUPT, E_NUMERIC = -1, -2, -3, -4, -5, -6, -7


class GpirtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[gpirt {code}] {msg}")
        self.code = code


class Options(C.Structure):
    _fields_ = [
        ("rng_kind", C.c_int),
        ("seed", C.c_uint64),
        ("theta_stabilise", C.c_int),
        ("fstar_fused", C.c_int),
        ("device", C.c_int),
        ("reserved0", C.c_int),
        ("item0", C.c_int64),
        ("m_total", C.c_int64),
        ("reserved1", C.c_int),
        ("kernel_fp32", C.c_int),
        ("kstar_rank", C.c_int),
        ("reserved", C.c_int * 5),
    ]


TICK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int)

_vp, _i64, _i32, _u64, _u32, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_uint64, C.c_uint32, C.c_double
_dp = C.POINTER(C.c_double)

# name -> (restype, argtypes); every symbol include/gpirt_hip.h declares
SIGNATURES = {
    "gpirt_version": (_i32, []),
    "gpirt_last_error": (C.c_char_p, []),
    "gpirt_device_count": (_i32, [C.POINTER(_i32)]),
    "gpirt_create": (_i32, [C.POINTER(_vp), _i32, _vp]),
    "gpirt_create_own_stream": (_i32, [C.POINTER(_vp), _i32]),
    "gpirt_destroy": (_i32, [_vp]),
    "gpirt_synchronize": (_i32, [_vp]),
    "gpirt_set_stream": (_i32, [_vp, _vp]),
    "gpirt_calibrate_mfma_f64": (_i32, [_vp, _dp]),
    "gpirt_config_get": (_i32, [_vp, C.c_char_p, C.POINTER(_i32)]),
    "gpirt_config_set": (_i32, [_vp, C.c_char_p, _i32]),
    "gpirt_guard_fallbacks": (_i32, [_vp, C.POINTER(_i32)]),
    "gpirt_debug_trip_guard": (_i32, [_vp, _i32]),
    "gpirt_debug_rs_cand_limit": (_i32, [_vp, _i32]),
    "gpirt_debug_rs_mispredict": (_i32, [_vp, _i32]),
    "gpirt_debug_rs_trace": (_i32, [_vp, _i32]),
    "gpirt_debug_last_mcmc_fallbacks": (_i32, []),
    "gpirt_se_kernel": (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _dbl]),
    "gpirt_potrf_lower": (_i32, [_vp, _vp, _i64, _i64]),
    "gpirt_factor": (_i32, [_vp, _vp, _i64, _vp, _i64]),
    "gpirt_potrf_panel_width": (_i64, []),
    "gpirt_potrf_begin": (_i32, [_vp]),
    "gpirt_potrf_panel_factor": (_i32, [_vp, _vp, _i64, _i64, _i64]),
    "gpirt_potrf_panel_update": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64]),
    "gpirt_potrf_panel_copy": (_i32, [_vp, _vp, _i64, _i64, _i64, _vp, _i32]),
    "gpirt_potrf_finish": (_i32, [_vp]),
    "gpirt_potrf_subpanel_width": (_i64, [_i64]),
    "gpirt_potrf_panel_factor_part": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32]),
    "gpirt_potrf_panel_update_part": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, _i32]),
    "gpirt_potrf_panel_copy_part": (_i32, [_vp, _vp, _i64, _i64, _i64, _i32, _vp, _i64, _i32]),
    "gpirt_debug_streams_busy": (_i32, [_vp, C.POINTER(_i32)]),
    "gpirt_debug_ll_term": (_i32, [_vp, _vp, _i64, _vp, _i32]),
    "gpirt_debug_theta_logpost": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _vp]),
    "gpirt_debug_theta_clock": (_i32, [_vp, _vp, _vp, _i64, _i64, _vp, _i64]),
    "gpirt_debug_panel_trace": (_i32, [_vp, _i64, _vp, _i64]),
    "gpirt_trmm_lz": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64]),
    "gpirt_trsm_lower": (_i32, [_vp, _vp, _i64, _i64, _vp, _i64, _i64, _i32]),
    "gpirt_gemm": (_i32, [_vp, _i32, _i32, _i64, _i64, _i64, _dbl, _vp, _i64, _vp, _i64, _dbl, _vp, _i64]),
    "gpirt_ll_bar": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "gpirt_draw_f": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _u64, _u32, _vp]),
    "gpirt_draw_fstar": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _i64, _u64, _u32, _i32, _vp, _vp, _vp]),
    "gpirt_draw_theta": (_i32, [_vp, _vp, _vp, _i64, _i64, _u64, _u32, _i32, _vp, _vp]),
    "gpirt_draw_beta": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _u64, _u32]),
    "gpirt_item_uniforms": (_i32, [_vp, _u64, _u32, _u32, _u32, _i64, _i64, _vp]),
    "gpirt_item_normals": (_i32, [_vp, _u64, _u32, _u32, _u32, _i64, _i64, _vp]),
    "gpirt_rstream_create": (_i32, [C.POINTER(_vp), _u32]),
    "gpirt_rstream_from_state": (_i32, [C.POINTER(_vp), C.POINTER(_u32), _i32]),
    "gpirt_rstream_get_state": (_i32, [_vp, C.POINTER(_u32), C.POINTER(_i32)]),
    "gpirt_rstream_destroy": (_i32, [_vp]),
    "gpirt_rstream_unif": (_i32, [_vp, _dp, _i64]),
    "gpirt_rstream_norm": (_i32, [_vp, _dp, _i64]),
    "gpirt_default_options": (None, [C.POINTER(Options)]),
    "gpirt_fast_options": (None, [C.POINTER(Options)]),
    "gpirt_mcmc": (_i32, [_dp, _i64, _i64, _dp, _i32, _i32, _dp, _dp, _dp, C.POINTER(Options), _vp,
                           TICK_FN, _vp, _dp, _dp, _dp, _dp]),
    "gpirt_sampler_create": (_i32, [C.POINTER(_vp), _vp, _dp, _i64, _i64, _dp, _dp, _dp, _dp,
                                     C.POINTER(Options), _vp]),
    "gpirt_sampler_destroy": (_i32, [_vp]),
    "gpirt_sampler_init": (_i32, [_vp]),
    "gpirt_sampler_step": (_i32, [_vp]),
    "gpirt_sampler_draw_f": (_i32, [_vp]),
    "gpirt_sampler_draw_fstar": (_i32, [_vp]),
    "gpirt_sampler_theta_partial": (_i32, [_vp]),
    "gpirt_sampler_theta_finish": (_i32, [_vp]),
    "gpirt_sampler_set_theta_block": (_i32, [_vp, _vp, _i64, _i64, _i64]),
    "gpirt_sampler_theta_block": (_i32, [_vp]),
    "gpirt_sampler_theta_commit": (_i32, [_vp]),
    "gpirt_sampler_draw_beta": (_i32, [_vp]),
    "gpirt_sampler_factor": (_i32, [_vp]),
    "gpirt_sampler_skip_factor": (_i32, [_vp]),
    "gpirt_sampler_adopt_factor": (_i32, [_vp, _i32]),
    "gpirt_sampler_build_cov": (_i32, [_vp]),
    "gpirt_sampler_panel_factor": (_i32, [_vp, _i64]),
    "gpirt_sampler_panel_update": (_i32, [_vp, _i64, _i64]),
    "gpirt_sampler_panel_copy": (_i32, [_vp, _i64, _vp, _i32]),
    "gpirt_sampler_panel_rows": (_i32, [_vp, C.POINTER(_i64)]),
    "gpirt_sampler_panel_factor_part": (_i32, [_vp, _i64, _i32]),
    "gpirt_sampler_panel_update_part": (_i32, [_vp, _i64, _i64, _i32]),
    "gpirt_sampler_panel_copy_part": (_i32, [_vp, _i64, _i32, _vp, _i64, _i32]),
    "gpirt_sampler_ldl": (_i32, [_vp, C.POINTER(_i64)]),
    "gpirt_sampler_copy_state": (_i32, [_vp, _vp]),
    "gpirt_sampler_accumulate_irf": (_i32, [_vp]),
    "gpirt_sampler_iteration": (_i32, [_vp, C.POINTER(_i32)]),
    "gpirt_sampler_check": (_i32, [_vp]),
    "gpirt_sampler_devptr": (_i32, [_vp, C.c_char_p, C.POINTER(_vp), C.POINTER(_i64)]),
    "gpirt_sampler_get": (_i32, [_vp, C.c_char_p, _vp, _i64]),
    "gpirt_sampler_set": (_i32, [_vp, C.c_char_p, _dp, _i64]),
    "gpirt_sampler_finish_irfs": (_i32, [_vp, _i32, _dp]),
    "gpirt_sampler_enable_timing": (_i32, [_vp, _i32]),
    "gpirt_sampler_stage_times": (_i32, [_vp, _dp, _i32, C.POINTER(_i32), C.POINTER(C.c_char_p)]),
    "gpirt_prof_trailing": (_i32, [_vp, _i32, _dp, C.POINTER(_i64), _dp]),
    "gpirt_prof_enable": (_i32, [_vp, _i32]),
    "gpirt_prof_syrk": (_i32, [_vp, _i32, _i32, _dp, C.POINTER(_i64), _dp]),
    "gpirt_prof_syrk_bytes": (_i32, [_vp, _i32, _dp]),
    "gpirt_sampler_set_iteration": (_i32, [_vp, _i32]),
}

_lib = None


def load():
    """dlopen the in-tree HIP library and attach the signatures (fails loudly if it is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpirtError(E_NODEVICE, f"{LIB_PATH} is missing: build it with "
                                         "`python -m gpirt_amd.build` (hipcc, gfx950). There is no CPU fallback.")
        # Load order matters in a Python process that also uses PyTorch-ROCm: torch bundles its own HIP runtime, and a
        # process that initialises the system runtime first (through this library) and torch's second ends with one of
        # them reporting "no ROCm-capable device".  Importing torch first makes the order the same everywhere
        # (the R host of INTEGRATION.md has no torch and no such issue).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return load().gpirt_last_error().decode("utf-8", "replace")


def check(rc: int) -> int:
    if rc < 0:
        raise GpirtError(rc, last_error())
    return rc


def default_options() -> Options:
    o = Options()
    load().gpirt_default_options(C.byref(o))
    return o


def fast_options() -> Options:
    """The throughput preset of the C ABI (gpirt_fast_options): item-keyed RNG, theta_stabilise, fused + rank-64 draw_fstar."""
    o = Options()
    load().gpirt_fast_options(C.byref(o))
    return o
