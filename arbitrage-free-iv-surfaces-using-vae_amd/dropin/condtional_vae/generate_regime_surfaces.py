"""Drop-in for the library part of the reference's ``generate_regime_surfaces`` script.

The reference file is a script that parses arguments at import time; its reusable pieces are
four functions, kept here under the same names and signatures
(generate_regime_surfaces.py:106-283):

  generate_iv_surface(params, T_grid, logm_grid, spot, r, q) -> (iv, success)     variant B
  check_static_arbitrage(iv_surface, T_grid, logm_grid, spot, r, q) -> [(mat, strike), ...] | None
  check_calendar_arbitrage(...)                                   -> [(mat, strike), ...] | None
  check_all_arbitrage(...) -> {static_violations, calendar_violations, total_violations,
                               is_arbitrage_free} | None

plus ``regime_statistics`` -- the per-regime loop of :296-362 for a whole [N,5] parameter block
in one pass over the GPU.  All of them run the HIP kernels (variant-B Black-Scholes form, i.e.
with log-moneyness entering d1 with the sign the reference uses at :157 / :213 / :251).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
for _p in (_ROOT, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import afiv_bootstrap  # noqa: E402
from heston_model_ql import cached_grid  # noqa: E402

_afiv = afiv_bootstrap.load()


def _surfaces_b(params, T_grid, logm_grid, spot, r, q):
    p = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 5), device="cuda")
    grid = cached_grid(logm_grid, T_grid, r, q)
    iv, flags, _ = grid.surface(p, float(spot), "B")
    return iv, flags


def generate_iv_surface(params, T_grid, logm_grid, spot, r, q):
    """Variant-B surface: success means no NaN anywhere (:180)."""
    try:
        row = np.array([float(x) for x in params], dtype=np.float64).reshape(1, 5)
    except Exception:
        return None, False
    iv, flags = _surfaces_b(row, T_grid, logm_grid, spot, r, q)
    ok = bool(int(flags[0, 3]) & _afiv.STATUS_SUCCESS_B)
    return iv[0].cpu().numpy(), ok


def _check(iv_surface, T_grid, logm_grid, spot, r, q):
    iv = torch.as_tensor(np.asarray(iv_surface, dtype=np.float64)[None], device="cuda").contiguous()
    grid = cached_grid(logm_grid, T_grid, r, q)
    flags, mask = grid.arb_check(iv, None, float(spot), "B", want_mask=True)
    return flags[0].cpu().numpy(), mask[0].cpu().numpy()


def check_static_arbitrage(iv_surface, T_grid, logm_grid, spot, r, q):
    flags, mask = _check(iv_surface, T_grid, logm_grid, spot, r, q)
    if flags[0] < 0:
        return None
    return [(int(m), int(k)) for m, k in np.argwhere(mask & 1)]


def check_calendar_arbitrage(iv_surface, T_grid, logm_grid, spot, r, q):
    flags, mask = _check(iv_surface, T_grid, logm_grid, spot, r, q)
    if flags[1] < 0:
        return None
    # the reference walks strikes in the outer loop (:237-259)
    hits = np.argwhere((mask & 2).T)
    return [(int(m), int(k)) for k, m in hits]


def check_all_arbitrage(iv_surface, T_grid, logm_grid, spot, r, q):
    flags, _ = _check(iv_surface, T_grid, logm_grid, spot, r, q)
    if flags[0] < 0 or flags[1] < 0:
        return None
    return {"static_violations": int(flags[0]), "calendar_violations": int(flags[1]),
            "total_violations": int(flags[0] + flags[1]),
            "is_arbitrage_free": bool(flags[3] & _afiv.STATUS_ARB_FREE)}


def regime_statistics(params, T_grid, logm_grid, spot, r, q):
    """Everything the reference records per regime (:337-362) for one block of samples."""
    params = np.asarray(params, dtype=np.float64).reshape(-1, 5)
    iv, flags = _surfaces_b(params, T_grid, logm_grid, spot, r, q)
    flags_np = flags.cpu().numpy()
    ok = (flags_np[:, 3] & _afiv.STATUS_SUCCESS_B) != 0
    n_total, n_valid = len(params), int(ok.sum())
    if n_valid == 0:
        return None
    s = iv[torch.as_tensor(ok, device=iv.device)].cpu().numpy()
    checked = flags_np[ok]
    atm = len(np.asarray(logm_grid)) // 2
    return {
        "n_total": n_total, "n_valid": n_valid, "success_rate": n_valid / n_total * 100,
        "atm_iv_mean": s[:, :, atm].mean(axis=0), "atm_iv_std": s[:, :, atm].std(axis=0),
        "mean_surface": s.mean(axis=0), "std_surface": s.std(axis=0),
        "median_surface": np.median(s, axis=0),
        "p5_surface": np.percentile(s, 5, axis=0), "p95_surface": np.percentile(s, 95, axis=0),
        "arbitrage_free_count": int(((checked[:, 3] & _afiv.STATUS_ARB_FREE) != 0).sum()),
        "arbitrage_free_pct": float(((checked[:, 3] & _afiv.STATUS_ARB_FREE) != 0).mean() * 100),
        "avg_static_violations": float(checked[:, 0].mean()),
        "avg_calendar_violations": float(checked[:, 1].mean()),
        "surfaces": s,
    }
