"""ctypes binding of libafiv.so (the C ABI declared in include/afiv.h).

The library is built in-tree by ``csrc/Makefile`` (``__graft_entry__.build()`` drives it).  There
is no fallback: if the shared object is missing or a call returns an error code, a RuntimeError
is raised.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFIV_LIB_PATH selects another build of the same library (kernel-shape experiments)
LIB_PATH = os.environ.get("AFIV_LIB_PATH") or os.path.join(_HERE, "libafiv.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_i64 = ctypes.c_int64
c_double = ctypes.c_double
_dp = ctypes.POINTER(ctypes.c_double)
_fp = ctypes.POINTER(ctypes.c_float)


class DecoderWeights(ctypes.Structure):
    """struct afiv_decoder_weights"""
    _fields_ = [
        ("lin_w", _fp * 4), ("lin_b", _fp * 4),
        ("bn_w", _fp * 3), ("bn_b", _fp * 3), ("bn_mean", _fp * 3), ("bn_var", _fp * 3),
        ("param_mean", _fp), ("param_std", _fp), ("bn_eps", ctypes.c_float),
    ]


# name -> (restype, argtypes); every symbol include/afiv.h declares
SIGNATURES = {
    "afiv_version": (c_int, []),
    "afiv_error_string": (ctypes.c_char_p, [c_int]),
    "afiv_grid_create": (c_int, [ctypes.POINTER(c_void_p), _dp, c_int, _dp, c_int, c_double, c_double,
                                 c_int, c_i64]),
    "afiv_grid_destroy": (c_int, [c_void_p]),
    "afiv_grid_traits": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "afiv_price": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "afiv_implied_vol": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_int, c_void_p, c_void_p]),
    "afiv_arb_check": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_double, c_int, c_void_p, c_void_p,
                               c_void_p]),
    "afiv_implied_vol_checked": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_int, c_void_p, c_void_p,
                                         c_void_p]),
    "afiv_surface": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_int, c_void_p, c_void_p, c_void_p,
                             c_void_p]),
    "afiv_surface_static": (c_int, [c_void_p, c_void_p, c_i64, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "afiv_decoder_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(DecoderWeights)]),
    "afiv_decoder_destroy": (c_int, [c_void_p]),
    "afiv_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p, c_void_p,
                            c_void_p]),
    "afiv_arbitrage_penalty": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "afiv_stage1_objective": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_double,
                                      c_void_p, c_void_p]),
    "afiv_density_wasserstein": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p,
                                         c_void_p]),
    "afiv_ensemble_stats": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_void_p, c_void_p]),
    "afiv_ensemble_stats_groups": (c_int, [c_void_p, c_void_p, c_i64, c_i64, c_i64, c_void_p, c_void_p]),
    "afiv_debug_math": (c_int, [c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().afiv_error_string(code).decode()
        raise RuntimeError(f"{what} failed: {msg} (code {code})")
