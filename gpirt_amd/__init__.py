"""gpirt_amd -- MI355X-native (gfx950) implementation of the GP linear-algebra hot path of
duckmayr/gpirt's gpirtMCMC(), behind the reference's own boundary.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/gpirt_hip.h) -> libgpirt_hip.so
  _lib.py          ctypes binding of that C ABI (fails loudly when the library / GPU is missing)
  ops.py           operator-level host API (K, chol, trmm, trsm, draw_f, draw_fstar, ...)
  sampler.py       gpirtMCMC() mirror of the reference's R entry point, and the stage-driven Sampler
  distributed.py   item-column sharding over torch.distributed (RCCL), one process per GPU
  response_matrix.py, synthetic.py   host-side data preparation
"""
from .response_matrix import as_response_matrix, is_response_matrix, response_matrix  # noqa: F401
from .sampler import Sampler, gpirtMCMC  # noqa: F401

__all__ = ["gpirtMCMC", "Sampler", "response_matrix", "as_response_matrix", "is_response_matrix"]
