"""The C-ABI library on a machine WITHOUT a GPU: it loads, exports every symbol include/gpirt_hip.h
declares, its host-only entries work, and every compute entry fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gpirt_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "gpirt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpirt_[a-z0-9_]+)\s*\(", src)) - {"gpirt_tick_fn"})


def test_exports_every_declared_symbol(lib):
    from gpirt_amd import _lib
    names = _declared()
    assert len(names) >= 45
    for n in names:
        assert hasattr(lib, n), f"libgpirt_hip.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_version_and_options(lib):
    from gpirt_amd import _lib
    assert lib.gpirt_version() >= 101
    o = _lib.default_options()
    # the C defaults are the reference's contract: R-stream replay, draw_theta as written
    assert o.rng_kind == _lib.RNG_RSTREAM and o.device == -1 and o.theta_stabilise == 0 and o.fstar_fused == 0
    assert o.kernel_fp32 == 0 and o.kstar_rank == 0 and o.reserved0 == 0 and o.reserved1 == 0 and not any(o.reserved)
    # the throughput preset: exactly what bench.py times as its headline (gpirt_fast_options, include/gpirt_hip.h)
    f = _lib.fast_options()
    assert f.rng_kind == _lib.RNG_ITEM and f.theta_stabilise == 1 and f.fstar_fused == 1 and f.kstar_rank == 64
    assert f.kernel_fp32 == 0 and f.device == -1 and f.item0 == 0 and f.m_total == 0 and not any(f.reserved)
    # the named fields sit where version 100 had reserved[1] / reserved[2] of an int[8]: same layout, same size
    import ctypes as C
    base = _lib.Options.reserved1.offset
    assert _lib.Options.kernel_fp32.offset == base + 4 and _lib.Options.kstar_rank.offset == base + 8
    assert _lib.Options.reserved.offset == base + 12 and C.sizeof(_lib.Options) == base + 32


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("this test is for the GPU-less container")
    from gpirt_amd import _lib
    h = C.c_void_p()
    assert lib.gpirt_create(C.byref(h), -1, None) == _lib.E_NODEVICE
    assert "no CPU fallback" in _lib.last_error()
    y = np.ones((4, 2), order="F")
    th = np.zeros(4)
    p = np.zeros((2, 2), order="F")
    out = [np.zeros(s, order="F") for s in ((2, 4), (2, 2, 2), (4, 2, 2), (1001, 2))]
    dp = C.POINTER(C.c_double)
    rc = lib.gpirt_mcmc(y.ctypes.data_as(dp), 4, 2, th.ctypes.data_as(dp), 1, 0, p.ctypes.data_as(dp),
                        p.ctypes.data_as(dp), p.ctypes.data_as(dp), None, None, _lib.TICK_FN(0), None,
                        *[o.ctypes.data_as(dp) for o in out])
    assert rc == _lib.E_NODEVICE
    from gpirt_amd import gpirtMCMC
    with pytest.raises(_lib.GpirtError):
        gpirtMCMC(np.array([[1, 0], [0, 1], [1, 1], [0, 0]]), 1, 0, vote_codes=dict(yea=[1], nay=[0], missing=[None]))


def test_argument_errors(lib):
    from gpirt_amd import _lib
    assert lib.gpirt_device_count(None) == _lib.E_ARG
    assert lib.gpirt_rstream_create(None, 1) == _lib.E_ARG
    assert "bad argument" in _lib.last_error()


def test_oracle_is_not_linked_into_the_product():
    """The product must never route through the oracle."""
    import subprocess
    from gpirt_amd import _lib
    out = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpirt_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "gpirt_oracle" not in txt, f
