"""CPU tests of the boundary: the shared library loads, exports every symbol include/afiv.h
declares, rejects bad arguments without touching a GPU, and the product never imports the
oracle.  No compute call is made here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "arbitrage-free-iv-surfaces-using-vae_amd")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "afiv.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(afiv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("afiv_version", "afiv_grid_create", "afiv_grid_destroy", "afiv_grid_traits", "afiv_price",
              "afiv_implied_vol", "afiv_arb_check", "afiv_implied_vol_checked", "afiv_surface", "afiv_decoder_create",
              "afiv_decoder_destroy", "afiv_decode", "afiv_arbitrage_penalty",
              "afiv_stage1_objective", "afiv_error_string", "afiv_debug_math", "afiv_surface_static",
              "afiv_density_wasserstein", "afiv_ensemble_stats", "afiv_ensemble_stats_groups"):
        assert s in syms


def test_library_exports_every_declared_symbol(afiv):
    lib = afiv._lib.lib()
    for s in declared_symbols():
        assert hasattr(lib, s), f"libafiv.so does not export {s}"
    assert set(afiv._lib.SIGNATURES) == set(declared_symbols())
    assert lib.afiv_version() == 200


def test_error_strings_and_null_checks_without_gpu(afiv):
    lib = afiv._lib.lib()
    assert lib.afiv_error_string(0) == b"ok"
    assert b"NULL" in lib.afiv_error_string(-1)
    h = ctypes.c_void_p(0)
    dp = ctypes.POINTER(ctypes.c_double)
    x = np.zeros(4)
    # NULL out pointer / NULL arrays
    assert lib.afiv_grid_create(None, x.ctypes.data_as(dp), 4, x.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -1
    assert lib.afiv_grid_create(ctypes.byref(h), None, 4, x.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -1
    # sizes out of range are rejected before any device call
    assert lib.afiv_grid_create(ctypes.byref(h), x.ctypes.data_as(dp), 0, x.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -2
    assert lib.afiv_grid_create(ctypes.byref(h), x.ctypes.data_as(dp), 65, x.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -2
    assert lib.afiv_grid_create(ctypes.byref(h), x.ctypes.data_as(dp), 4, x.ctypes.data_as(dp), 65, 0.0, 0.0, 1, 0) == -2
    # lattice values that are not finite (or a year fraction beyond any calendar) are rejected too,
    # before any device call
    for bad in (np.nan, np.inf, 1e7):
        y = np.array([0.1, 0.2, bad, 0.4])
        assert lib.afiv_grid_create(ctypes.byref(h), x.ctypes.data_as(dp), 4, y.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -4
    y = np.array([0.0, np.nan, 0.1, 0.2])
    assert lib.afiv_grid_create(ctypes.byref(h), y.ctypes.data_as(dp), 4, x.ctypes.data_as(dp), 4, 0.0, 0.0, 1, 0) == -4
    assert lib.afiv_grid_create(ctypes.byref(h), x.ctypes.data_as(dp), 4, x.ctypes.data_as(dp), 4, np.nan, 0.0, 1, 0) == -4
    assert h.value in (None, 0)
    assert lib.afiv_price(None, None, 1, None, None) == -1
    assert lib.afiv_surface_static(None, None, 1, 1.0, 0, None, None, None) == -1
    assert b"scratch" in lib.afiv_error_string(-6)
    assert lib.afiv_decode(None, None, None, 1, 1, None, None, None, None) == -1
    assert lib.afiv_decoder_create(None, None) == -1
    assert lib.afiv_grid_destroy(None) == 0
    assert lib.afiv_decoder_destroy(None) == 0


def test_wrappers_fail_loudly_without_a_gpu(afiv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="GPU"):
        afiv.Grid(np.linspace(-0.2, 0.2, 21), [0.5, 1.0])


def test_missing_library_is_an_error_not_a_fallback(afiv, monkeypatch):
    monkeypatch.setattr(afiv._lib, "_lib", None)
    monkeypatch.setattr(afiv._lib, "LIB_PATH", "/nonexistent/libafiv.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        afiv._lib.lib()


def test_product_code_never_touches_the_oracle():
    offenders = []
    for base, _dirs, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".inc")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\bimport oracle\b|from oracle\b|oracle/|liboracle", text):
                    offenders.append(os.path.join(base, f))
    assert not offenders, offenders


def test_laguerre_tables_are_identical_for_kernel_and_oracle():
    a = open(os.path.join(PKG, "csrc", "laguerre144.inc")).read()
    b = open(os.path.join(ROOT, "oracle", "laguerre144.inc")).read()
    assert a == b
    xs = [float.fromhex(v) for v in re.findall(r"(0x[0-9a-f.]+p[+-]?\d+)", a.split("AFIV_LAG_W_LIST")[0])]
    assert len(xs) == 144 and all(x1 < x2 for x1, x2 in zip(xs, xs[1:]))
    from scipy.special import roots_laguerre
    x0, w0 = roots_laguerre(144)
    assert np.max(np.abs(np.array(xs) - x0) / x0) < 1e-14


def test_shard_bounds_cover_all_rows(afiv):
    for n in (0, 1, 7, 100, 1_000_003):
        for world in (1, 2, 3, 8):
            covered = 0
            prev_hi = 0
            for r in range(world):
                lo, hi = afiv.dist.shard_bounds(n, world, r)
                assert lo == prev_hi and hi >= lo
                covered += hi - lo
                prev_hi = hi
            assert covered == n and prev_hi == n
    with pytest.raises(ValueError):
        afiv.dist.shard_bounds(4, 2, 2)
