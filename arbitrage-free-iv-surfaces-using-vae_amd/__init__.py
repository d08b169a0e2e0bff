"""afiv_amd -- MI355X-native surface-generation path (decode -> Heston pricing -> implied vol ->
arbitrage checks) behind the reference project's Python call surface.

Layout
  csrc/      hand-written HIP kernels and the C ABI (include/afiv.h) -> libafiv.so
  _lib.py    ctypes binding of libafiv.so (no fallback: raises if the library is missing)
  api.py     torch-facing wrappers (device memory, streams, shapes)
  dist.py    one-process-per-GPU sharding of parameter rows + final all-gather
  dropin/    modules with the reference's import names (heston_model_ql, cvae_model,
             generate_iv_surface_by_date, generate_regime_surfaces)
"""
from . import _lib                                           # noqa: F401
from .api import (Grid, Decoder, debug_math, ensemble_stats, ENSEMBLE_ONE_SWEEP, VARIANTS, STATUS_ARB_FREE, STATUS_BUTTERFLY_FREE,   # noqa: F401
                  STATUS_VALID, STATUS_SUCCESS_A, STATUS_SUCCESS_B)
from . import dist                                           # noqa: F401

__all__ = ["Grid", "Decoder", "debug_math", "ensemble_stats", "ENSEMBLE_ONE_SWEEP", "dist", "VARIANTS", "STATUS_ARB_FREE", "STATUS_BUTTERFLY_FREE",
           "STATUS_VALID", "STATUS_SUCCESS_A", "STATUS_SUCCESS_B"]
