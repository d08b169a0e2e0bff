"""Drop-in for the reference's ``heston_model_ql`` module, backed by the MI355X pricing kernel.

Same names, argument meaning and error behaviour as heston_model_ql.py:13-333, but there is no
QuantLib underneath: ``price_call`` / ``price_ratio`` run ``afiv_price`` (the same quadrature
QuantLib's ``AnalyticHestonEngine(model)`` evaluates, heston_model_ql.py:87) on the current GPU
and copy the few numbers back.  Scalar calls are a convenience path; bulk work should go
through ``afiv_amd.Grid`` directly.  A GPU and libafiv.so are required -- nothing here falls
back to the CPU.
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
import afiv_bootstrap  # noqa: E402

_afiv = afiv_bootstrap.load()

_GRID_CACHE: "OrderedDict[tuple, object]" = OrderedDict()
_GRID_CACHE_MAX = 64


def cached_grid(logm, tau, r, q, trunc_days=True):
    """Grids are immutable device objects; scalar callers hit the same few lattices repeatedly."""
    lm = np.ascontiguousarray(np.asarray(logm, dtype=np.float64).ravel())
    ta = np.ascontiguousarray(np.asarray(tau, dtype=np.float64).ravel())
    key = (lm.tobytes(), ta.tobytes(), float(r), float(q), bool(trunc_days), torch.cuda.current_device())
    g = _GRID_CACHE.get(key)
    if g is None:
        g = _afiv.Grid(lm, ta, r, q, trunc_days)
        _GRID_CACHE[key] = g
        if len(_GRID_CACHE) > _GRID_CACHE_MAX:
            _GRID_CACHE.popitem(last=False)
    else:
        _GRID_CACHE.move_to_end(key)
    return g


class HestonModelQL:
    """Heston model with spot normalised to 1 (heston_model_ql.py:35-87)."""

    def __init__(self, kappa, theta, sigma_v, rho, v0, r=0.067, q=0.0):
        self.kappa = float(kappa)
        self.theta = float(theta)
        self.sigma_v = float(sigma_v)
        self.rho = float(rho)
        self.v0 = float(v0)
        self.r = float(r)
        self.q = float(q)
        self._setup_ql_objects()

    def _setup_ql_objects(self):
        # kept under the reference's name: (re)binds the parameter row the kernel reads
        self.spot = 1.0
        if not torch.cuda.is_available():
            raise RuntimeError("HestonModelQL needs a GPU (no CPU fallback)")
        self._params = torch.tensor([[self.kappa, self.theta, self.sigma_v, self.rho, self.v0]],
                                    dtype=torch.float64, device="cuda")

    def update_parameters(self, kappa, theta, sigma_v, rho, v0, r=None, q=None):
        self.kappa, self.theta, self.sigma_v, self.rho, self.v0 = kappa, theta, sigma_v, rho, v0
        if r is not None:
            self.r = r
        if q is not None:
            self.q = q
        self._setup_ql_objects()

    def check_feller_condition(self):
        return 2 * self.kappa * self.theta > self.sigma_v ** 2

    def _ratios(self, logm_array, tau):
        grid = cached_grid(logm_array, [tau], self.r, self.q)
        return grid.price(self._params).reshape(-1).cpu().numpy()

    def price_call(self, S, K, tau):
        """Call price(s) with spot 1; ``S`` is accepted and ignored exactly as in
        heston_model_ql.py:109-149.  A failing evaluation yields 0.0 (:145-146)."""
        scalar = np.isscalar(K)
        strikes = np.atleast_1d(np.asarray(K, dtype=np.float64))
        prices = np.zeros(strikes.shape, dtype=np.float64)
        # at most AFIV_MAX_STRIKES per lattice
        for lo in range(0, strikes.size, 64):
            blk = strikes[lo:lo + 64]
            try:
                with np.errstate(all="ignore"):
                    ratios = self._ratios(np.log(blk), tau)
                prices[lo:lo + 64] = ratios * np.exp(-self.r * tau)
            except RuntimeError:
                prices[lo:lo + 64] = 0.0
        return prices[0] if scalar else prices

    def price_ratio(self, log_moneyness, tau):
        """c / (S exp(-r tau)) for log-moneyness log(K/S) (heston_model_ql.py:151-179)."""
        scalar = np.isscalar(log_moneyness)
        lm = np.atleast_1d(np.asarray(log_moneyness, dtype=np.float64))
        out = np.zeros(lm.shape, dtype=np.float64)
        for lo in range(0, lm.size, 64):
            try:
                out[lo:lo + 64] = self._ratios(lm[lo:lo + 64], tau)
            except RuntimeError:
                out[lo:lo + 64] = 0.0
        return out[0] if scalar else out

    def implied_volatility(self, S, K, tau, market_price):
        """Black-Scholes-Merton implied vol of one price (heston_model_ql.py:181-249): accuracy
        1e-6 on [0.001, 2.0], then a second attempt on [0.001, 5.0], else 0.2.  No caller in the
        reference uses it; it is host code (SciPy), not part of the GPU path, and its agreement
        with QuantLib's own solver is parity-unpinned."""
        from scipy.optimize import brentq
        from scipy.stats import norm

        def objective(sigma):
            st = sigma * np.sqrt(tau)
            d1 = (np.log(S / K) + (self.r - self.q + 0.5 * sigma ** 2) * tau) / st
            return S * np.exp(-self.q * tau) * norm.cdf(d1) - K * np.exp(-self.r * tau) * norm.cdf(d1 - st) \
                - market_price

        for hi, tol in ((2.0, 1e-6), (5.0, 2e-12)):
            try:
                return brentq(objective, 0.001, hi, xtol=tol, maxiter=100)
            except Exception:
                continue
        return 0.2


class TimeVaryingHestonModelQL:
    """One parameter set per maturity (heston_model_ql.py:252-328)."""

    def __init__(self, taus, params_list, r=0.067, q=0.0):
        self.taus = np.array(taus)
        self.params_list = params_list
        self.r = r
        self.q = q
        self.models = [HestonModelQL(p["kappa"], p["theta"], p["sigma_v"], p["rho"], p["v0"], r=r, q=q)
                       for p in params_list]

    def price_surface(self, log_moneyness_grid):
        grid = np.asarray(log_moneyness_grid, dtype=np.float64)
        out = np.zeros((len(self.taus), grid.shape[1]))
        for i, model in enumerate(self.models):
            out[i, :] = model.price_ratio(grid[i, :], self.taus[i])
        return out

    def check_all_feller_conditions(self):
        return [{"maturity": self.taus[i],
                 "feller_satisfied": m.check_feller_condition(),
                 "condition_value": 2 * m.kappa * m.theta - m.sigma_v ** 2}
                for i, m in enumerate(self.models)]


HestonModel = HestonModelQL
TimeVaryingHestonModel = TimeVaryingHestonModelQL
