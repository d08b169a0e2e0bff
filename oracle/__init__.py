"""CPU oracle for the surface-generation path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The shipped package never does.

It wraps ``oracle/liboracle.so`` (built from ``heston_oracle.c`` by ``oracle/Makefile``), the
plain-C restatement of the reference's per-option algorithm, and adds the few NumPy/SciPy
restatements that are naturally array code (decoder MLP, calibration objectives, ensemble
statistics).  Every function cites the reference lines it follows.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms).  Returns the library path."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "heston_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.afo_price_ratio.restype = ctypes.c_double
        L.afo_price_ratio.argtypes = [_dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int]
        L.afo_price_ratio_grid.restype = None
        L.afo_price_ratio_grid.argtypes = [_dp, ctypes.c_long, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_int, _dp]
        L.afo_ndtr.restype = ctypes.c_double
        L.afo_ndtr.argtypes = [ctypes.c_double]
        L.afo_brentq_poly.restype = ctypes.c_double
        L.afo_brentq_poly.argtypes = [_dp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_double, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.afo_implied_vol.restype = ctypes.c_double
        L.afo_implied_vol.argtypes = [ctypes.c_double] * 6 + [ctypes.c_int,
                                                              ctypes.POINTER(ctypes.c_int)]
        L.afo_implied_vol_grid.restype = None
        L.afo_implied_vol_grid.argtypes = [_dp, ctypes.c_long, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int, _dp]
        L.afo_arb_check.restype = None
        L.afo_arb_check.argtypes = [_dp, _dp, ctypes.c_long, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                    ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_int, _ip]
        L.afo_surface_batch.restype = None
        L.afo_surface_batch.argtypes = [_dp, ctypes.c_long, _dp, ctypes.c_int, _dp, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                        ctypes.c_int, _dp, _dp, _ip]
        L.afo_arbitrage_penalty.restype = None
        L.afo_arbitrage_penalty.argtypes = [_dp, ctypes.c_long, ctypes.c_double, ctypes.c_double,
                                            _dp, ctypes.POINTER(ctypes.c_long)]
        L.afo_set_threads.restype = None
        L.afo_set_threads.argtypes = [ctypes.c_int]
        L.afo_max_threads.restype = ctypes.c_int
        L.afo_max_threads.argtypes = []
        _LIB = L
    return _LIB


def set_threads(n: int) -> int:
    """Set the OpenMP thread count of the batch functions; returns the count in effect."""
    lib().afo_set_threads(int(n))
    return lib().afo_max_threads()


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


# --------------------------------------------------------------------------------------------
# C oracle wrappers
# --------------------------------------------------------------------------------------------
def price_ratio(params, logm, tau, r=0.067, q=0.0, trunc_days=True) -> float:
    """heston_model_ql.py:151-179 for one (log-moneyness, tau)."""
    p, pp = _c(params)
    return lib().afo_price_ratio(pp, r, q, float(logm), float(tau), int(trunc_days))


def price_ratio_grid(params, logm, tau, r=0.067, q=0.0, trunc_days=True) -> np.ndarray:
    """Price ratios [N, nT, nK]: one heston_model_ql.price_ratio call per point."""
    p, pp = _c(np.atleast_2d(params))
    lm, lmp = _c(logm)
    ta, tap = _c(tau)
    out = np.empty((p.shape[0], ta.size, lm.size))
    lib().afo_price_ratio_grid(pp, p.shape[0], lmp, lm.size, tap, ta.size, r, q,
                               int(trunc_days), out.ctypes.data_as(_dp))
    return out


def implied_vol_grid(ratio, logm, tau, spot, r=0.067, q=0.0, variant="A") -> np.ndarray:
    """generate_iv_surface_by_date.py:293-318 (A) / generate_regime_surfaces.py:136-177 (B)."""
    ra, rap = _c(ratio)
    lm, lmp = _c(logm)
    ta, tap = _c(tau)
    n = ra.size // (lm.size * ta.size)
    out = np.empty((n, ta.size, lm.size))
    lib().afo_implied_vol_grid(rap, n, lmp, lm.size, tap, ta.size, spot, r, q,
                               0 if variant == "A" else 1, out.ctypes.data_as(_dp))
    return out


def arb_check(iv, ratio, logm, tau, spot, r=0.067, q=0.0, bs_form="B") -> np.ndarray:
    """Flags [N,4] = (static, calendar, butterfly, status); see heston_oracle.c."""
    v, vp = _c(iv)
    lm, lmp = _c(logm)
    ta, tap = _c(tau)
    n = v.size // (lm.size * ta.size)
    if ratio is None:
        rap = None
    else:
        ra, rap = _c(ratio)
    flags = np.empty((n, 4), dtype=np.int32)
    lib().afo_arb_check(vp, rap, n, lmp, lm.size, tap, ta.size, spot, r, q,
                        0 if bs_form == "A" else 1, flags.ctypes.data_as(_ip))
    return flags


def surface_batch(params, logm, tau, spot, r=0.067, q=0.0, variant="A"):
    """Whole path (price -> IV -> checks), one option at a time.  Returns ratio, iv, flags."""
    p, pp = _c(np.atleast_2d(params))
    lm, lmp = _c(logm)
    ta, tap = _c(tau)
    n = p.shape[0]
    ratio = np.empty((n, ta.size, lm.size))
    iv = np.empty((n, ta.size, lm.size))
    flags = np.empty((n, 4), dtype=np.int32)
    lib().afo_surface_batch(pp, n, lmp, lm.size, tap, ta.size, spot, r, q,
                            0 if variant == "A" else 1, ratio.ctypes.data_as(_dp),
                            iv.ctypes.data_as(_dp), flags.ctypes.data_as(_ip))
    return ratio, iv, flags


def arbitrage_penalty(params_original, r=0.067, q=0.0) -> float:
    """cvae_model.py:245-298 on already de-normalised, transformed parameters [B,5]."""
    p, pp = _c(np.atleast_2d(params_original))
    s = ctypes.c_double(0.0)
    c = ctypes.c_long(0)
    lib().afo_arbitrage_penalty(pp, p.shape[0], r, q, ctypes.byref(s), ctypes.byref(c))
    return s.value / c.value if c.value > 0 else 0.0


def brentq_poly(coef, a, b, xtol=1e-6, rtol=4 * np.finfo(float).eps, maxiter=100):
    c, cp = _c(coef)
    err = ctypes.c_int(0)
    it = ctypes.c_int(0)
    x = lib().afo_brentq_poly(cp, a, b, xtol, rtol, maxiter, ctypes.byref(err), ctypes.byref(it))
    return x, err.value, it.value


# --------------------------------------------------------------------------------------------
# NumPy / SciPy restatements
# --------------------------------------------------------------------------------------------
def decode_params(z, c, w, param_mean, param_std) -> np.ndarray:
    """cvae_model.py:145-160 (decoder stack :84-105, eval mode) followed by
    generate_iv_surface_by_date.py:441-446, in float32.

    ``w`` maps the checkpoint's ``decoder.*`` state-dict names to arrays.  BatchNorm uses the
    running statistics with eps 1e-5; Dropout is the identity in eval mode.
    """
    f = np.float32
    z = np.asarray(z, f)
    c = np.asarray(c, f)
    if c.shape[0] == 1 and z.shape[0] > 1:
        c = np.repeat(c, z.shape[0], axis=0)
    h = np.concatenate([z, c], axis=1)
    for lin, bn in ((0, 1), (4, 5), (8, 9)):
        h = h @ np.asarray(w[f"decoder.{lin}.weight"], f).T + np.asarray(w[f"decoder.{lin}.bias"], f)
        mean = np.asarray(w[f"decoder.{bn}.running_mean"], f)
        var = np.asarray(w[f"decoder.{bn}.running_var"], f)
        h = (h - mean) / np.sqrt(var + f(1e-5)) * np.asarray(w[f"decoder.{bn}.weight"], f) \
            + np.asarray(w[f"decoder.{bn}.bias"], f)
        h = np.maximum(h, f(0))
    out = h @ np.asarray(w["decoder.12.weight"], f).T + np.asarray(w["decoder.12.bias"], f)
    t = out * np.asarray(param_std, f) + np.asarray(param_mean, f)
    p = t.copy()
    for i in (0, 1, 2, 4):
        p[:, i] = np.exp(t[:, i])
    p[:, 3] = np.tanh(t[:, 3])
    return p.astype(f)


def market_price_ratio(iv, logm, tau, r, q):
    """run_single_heston_calibration.py:85-97 (compute_price_ratio)."""
    from scipy.stats import norm
    kf = np.exp(logm - (r - q) * tau)
    st = np.sqrt(tau)
    d1 = (-np.log(kf) + 0.5 * iv ** 2 * tau) / (iv * st)
    d2 = d1 - iv * st
    return norm.cdf(d1) - kf * norm.cdf(d2)


def _market_points(iv_surface, t_grid, logm_grid, r, q):
    """run_single_heston_calibration.py:177-208: flatten the valid market points."""
    all_lm, all_tau, all_pr = [], [], []
    for m in range(len(t_grid)):
        row = iv_surface[m]
        mask = ~np.isnan(row) & (row > 0) & (row < 2.0)
        if mask.sum() < 3:
            continue
        lm = logm_grid[mask]
        pr = market_price_ratio(row[mask], lm, t_grid[m], r, q)
        all_lm.extend(lm)
        all_tau.extend([t_grid[m]] * len(lm))
        all_pr.extend(pr)
    return np.array(all_lm), np.array(all_tau), np.array(all_pr)


def _model_points(params, iv_surface, t_grid, logm_grid, r, q):
    """The model price ratio at every valid market point, in _market_points' order: the same
    per-point function as heston_model_ql.price_ratio, evaluated for the whole lattice in one call
    of the C oracle instead of one call per point."""
    full = price_ratio_grid(params, logm_grid, t_grid, r, q)[0]
    out = []
    for m in range(len(t_grid)):
        row = iv_surface[m]
        mask = ~np.isnan(row) & (row > 0) & (row < 2.0)
        if mask.sum() >= 3:
            out.extend(full[m][mask])
    return np.array(out)


def stage1_objective(params, iv_surface, t_grid, logm_grid, r=0.067, q=0.0,
                     drift_weight=1.0, model_pr=None, guard=True):
    """run_single_heston_calibration.py:252-287.  ``guard=False`` drops the NaN / negative-price
    test of :270-271 (diagnostic: deep out-of-the-money prices of 1e-17 carry a rounding-noise
    sign that differs between arithmetic implementations)."""
    kappa, theta, sigma_v, _rho, _v0 = params
    all_lm, all_tau, all_pr = _market_points(iv_surface, t_grid, logm_grid, r, q)
    if model_pr is None:
        model_pr = _model_points(params, iv_surface, np.asarray(t_grid, float), np.asarray(logm_grid, float), r, q)
    if guard and (np.any(np.isnan(model_pr)) or np.any(model_pr < 0)):
        return 1e10
    rmse = np.sqrt(np.mean((model_pr - all_pr) ** 2))
    atm = np.abs(all_lm) < 0.05
    drift = abs(np.mean(model_pr[atm]) - np.mean(all_pr[atm])) if atm.sum() > 0 else 0.0
    feller = 5.0 * max(0, sigma_v ** 2 - 2 * kappa * theta)
    return rmse + drift_weight * drift + feller


def density_wasserstein(params, lm, tau, market_pr, r, q, n_points=50, guard=True):
    """run_single_heston_calibration.py:100-151."""
    from scipy.stats import wasserstein_distance
    fine = np.linspace(lm.min(), lm.max(), n_points)
    mkt = np.interp(fine, lm, market_pr)
    mdl = price_ratio_grid(params, fine, [tau], r, q)[0, 0]
    if guard and (np.any(np.isnan(mdl)) or np.any(mdl < 0)):
        return np.inf
    dlm = fine[1] - fine[0]
    md = np.zeros(n_points)
    dd = np.zeros(n_points)
    # the reference's loop over i = 1 .. n_points - 2 (:127-131), element by element in one expression
    md[1:-1] = np.exp(r * tau) * ((mkt[2:] - 2 * mkt[1:-1] + mkt[:-2]) / dlm ** 2)
    dd[1:-1] = np.exp(r * tau) * ((mdl[2:] - 2 * mdl[1:-1] + mdl[:-2]) / dlm ** 2)
    md = np.maximum(md, 0)
    dd = np.maximum(dd, 0)
    if md.sum() < 1e-10 or dd.sum() < 1e-10:
        return np.inf
    md /= md.sum()
    dd /= dd.sum()
    return wasserstein_distance(fine, fine, md, dd)


def stage2_objective(params, iv_surface, t_grid, logm_grid, r=0.067, q=0.0,
                     drift_weight=1.0, wasserstein_weight=1.0, n_points=50, guard=True):
    """run_single_heston_calibration.py:342-406."""
    base = stage1_objective(params, iv_surface, t_grid, logm_grid, r, q, drift_weight, guard=guard)
    if base >= 1e10:
        return 1e10
    all_lm, all_tau, all_pr = _market_points(iv_surface, t_grid, logm_grid, r, q)
    dists = []
    for m in range(len(t_grid)):
        mask = np.abs(all_tau - t_grid[m]) < 1e-6
        if mask.sum() < 3:
            continue
        w = density_wasserstein(params, all_lm[mask], t_grid[m], all_pr[mask], r, q, n_points,
                                guard=guard)
        if np.isfinite(w):
            dists.append(w)
    avg = np.mean(dists) if dists else 0.0
    return base + wasserstein_weight * avg


def ensemble_stats(surfaces):
    """generate_iv_surface_by_date.py:474-478: nan-aware mean/median/p5/p95 over axis 0."""
    s = np.asarray(surfaces)
    return (np.nanmean(s, axis=0), np.nanmedian(s, axis=0),
            np.nanpercentile(s, 5, axis=0), np.nanpercentile(s, 95, axis=0))


def scipy_implied_vol_a(price_ratio_value, logm, tau, spot, r, q):
    """generate_iv_surface_by_date.py:296-318 evaluated with SciPy itself (brentq, norm.cdf)."""
    from scipy.optimize import brentq
    from scipy.stats import norm
    try:
        heston_price = price_ratio_value * spot * np.exp(-r * tau)
        K = spot * np.exp(logm)
        F = spot * np.exp((r - q) * tau)
        intrinsic = max(F - K, 0) * np.exp(-r * tau)
        if heston_price < intrinsic * 0.99:
            return np.nan

        def obj(iv):
            d1 = (np.log(F / K) + 0.5 * iv ** 2 * tau) / (iv * np.sqrt(tau))
            d2 = d1 - iv * np.sqrt(tau)
            return np.exp(-r * tau) * (F * norm.cdf(d1) - K * norm.cdf(d2)) - heston_price

        return brentq(obj, 0.001, 3.0, xtol=1e-6)
    except Exception:
        return np.nan


def scipy_implied_vol_b(price_ratio_value, logm, tau, spot, r, q):
    """generate_regime_surfaces.py:138-177 evaluated with SciPy itself."""
    from scipy.optimize import brentq
    from scipy.stats import norm
    try:
        heston_price = price_ratio_value * spot * np.exp(-r * tau)
        if heston_price < 1e-10 or np.isnan(price_ratio_value):
            return np.nan
        K = spot * np.exp(logm)

        def price(sigma):
            if sigma <= 0 or tau <= 0:
                return 1e10
            d1 = (logm + (r - q + 0.5 * sigma ** 2) * tau) / (sigma * np.sqrt(tau))
            d2 = d1 - sigma * np.sqrt(tau)
            return spot * np.exp(-q * tau) * norm.cdf(d1) - K * np.exp(-r * tau) * norm.cdf(d2)

        try:
            iv = brentq(lambda s: price(s) - heston_price, 0.001, 3.0, xtol=1e-6, maxiter=100)
            if np.isnan(iv) or iv <= 0 or iv > 3.0:
                iv = np.nan
            return iv
        except Exception:
            return np.nan
    except Exception:
        return np.nan
