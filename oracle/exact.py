"""50-digit evaluation of the reference's pricing quadrature -- TEST INFRASTRUCTURE ONLY.

The same formula as ``heston_oracle.c`` (heston_model_ql.py:151-179 through QuantLib's Gatheral
integrand and 144-node Gauss-Laguerre rule), evaluated with mpmath at 60 significant digits so that
it carries none of the eps / sigma_v^2 rounding noise every double-precision evaluation of that
formula has.  Both branches are evaluated AS WRITTEN: the logarithmic form for sigma_v > 1e-5 and
the first-order series form QuantLib switches to at sigma_v <= 1e-5.

Inputs are taken as the doubles the other implementations see: parameters, year fraction
``float(int(tau*365))/365`` (or tau itself), strike ``exp(logm)`` rounded to double, quadrature
nodes and weights of ``oracle/laguerre144.inc``.
"""
from __future__ import annotations

import os
import re

import mpmath as mp
import numpy as np

DPS = 60
_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES = None


def laguerre_tables():
    """The 144 nodes and exp(x)-scaled weights as the doubles of oracle/laguerre144.inc."""
    global _TABLES
    if _TABLES is None:
        text = open(os.path.join(_HERE, "laguerre144.inc")).read()

        def arr(name):
            start = text.index("#define %s" % name)
            vals = []
            for line in text[start:].splitlines()[1:]:
                vals.extend(float.fromhex(v) for v in re.findall(r"[-+]?0x[0-9a-fA-F.]+p[-+]?\d+", line))
                if not line.rstrip().endswith("\\"):
                    break
            return vals
        x, w = arr("AFIV_LAG_X_LIST"), arr("AFIV_LAG_W_LIST")
        assert len(x) == 144 and len(w) == 144
        _TABLES = (x, w)
    return _TABLES


def _fj(j, phi, kappa, theta, sigma, rho, v0, term, dd, sx, series):
    """Integrand of P_j at abscissa phi (heston_oracle.c:fj_gatheral, same operation for operation)."""
    sigma2 = sigma * sigma
    t1 = mp.mpc(kappa - (rho * sigma if j == 1 else 0), -rho * sigma * phi)
    u = mp.mpc(-phi, 1 if j == 1 else -1)
    d = mp.sqrt(t1 * t1 - sigma2 * phi * u)
    ex = mp.exp(-d * term)
    if not series:
        p = (t1 - d) / (t1 + d)
        g = mp.log((1 - p * ex) / (1 - p))
        arg = v0 * (t1 - d) * (1 - ex) / (sigma2 * (1 - ex * p)) \
            + kappa * theta / sigma2 * ((t1 - d) * term - 2 * g)
    else:
        td = phi / (2 * t1) * u
        p = td * sigma2 / (t1 + d)
        g = p * (1 - ex)
        arg = v0 * td * (1 - ex) / (1 - p * ex) + kappa * theta * (td * term - 2 * g / sigma2)
    arg = arg + mp.mpc(0, phi * (dd - sx))
    return mp.im(mp.exp(arg)) / phi


def price_ratio(params, logm, tau, r=0.067, q=0.0, trunc_days=True):
    """C / exp(-r tau) as an mpf; NaN (float) where the formula as written divides 0 by 0
    (sigma_v = 0 in the series form) or the option has expired (0 days -> NPV 0)."""
    X, W = laguerre_tables()
    with mp.workdps(DPS):
        kappa, theta, sigma, rho, v0 = (mp.mpf(float(v)) for v in params)
        tau = float(tau)
        term_f = float(int(tau * 365.0)) / 365.0 if trunc_days else tau
        if not term_f > 0.0:
            return mp.mpf(0)
        series = not (float(params[2]) > 1e-5)
        if series and float(params[2]) == 0.0:
            return float("nan")                 # 2 g / sigma^2 = 0 / 0
        term = mp.mpf(term_f)
        strike = mp.mpf(float(np.exp(float(logm))))
        rr, qq = mp.mpf(float(r)), mp.mpf(float(q))
        Dr, Dq = mp.exp(-rr * term), mp.exp(-qq * term)
        dd = -mp.log(Dr / Dq)
        sx = mp.log(strike)
        s1 = mp.mpf(0)
        s2 = mp.mpf(0)
        for x, w in zip(X, W):
            s1 += mp.mpf(w) * _fj(1, mp.mpf(x), kappa, theta, sigma, rho, v0, term, dd, sx, series)
            s2 += mp.mpf(w) * _fj(2, mp.mpf(x), kappa, theta, sigma, rho, v0, term, dd, sx, series)
        c = Dq * (s1 / mp.pi + mp.mpf(1) / 2) - strike * Dr * (s2 / mp.pi + mp.mpf(1) / 2)
        return c / mp.exp(-rr * mp.mpf(tau))


def price_ratio_points(params, logm, tau, points, r=0.067, q=0.0, trunc_days=True) -> np.ndarray:
    """The exact ratio, rounded to double (inf where it exceeds the double range), at the listed
    (row, maturity index, strike index) triples."""
    out = np.empty(len(points))
    for n, (i, t, k) in enumerate(points):
        v = price_ratio(params[i], logm[k], tau[t], r, q, trunc_days)
        try:
            out[n] = float(v)
        except OverflowError:
            out[n] = np.inf if v > 0 else -np.inf
    return out


def _fj_strike_free(j, phi, kappa, theta, sigma, rho, v0, term, series):
    """exp of the integrand's exponent without its strike term i phi (dd - sx): the part of _fj that
    every strike of a maturity shares."""
    sigma2 = sigma * sigma
    t1 = mp.mpc(kappa - (rho * sigma if j == 1 else 0), -rho * sigma * phi)
    u = mp.mpc(-phi, 1 if j == 1 else -1)
    d = mp.sqrt(t1 * t1 - sigma2 * phi * u)
    ex = mp.exp(-d * term)
    if not series:
        p = (t1 - d) / (t1 + d)
        g = mp.log((1 - p * ex) / (1 - p))
        arg = v0 * (t1 - d) * (1 - ex) / (sigma2 * (1 - ex * p)) \
            + kappa * theta / sigma2 * ((t1 - d) * term - 2 * g)
    else:
        td = phi / (2 * t1) * u
        p = td * sigma2 / (t1 + d)
        g = p * (1 - ex)
        arg = v0 * td * (1 - ex) / (1 - p * ex) + kappa * theta * (td * term - 2 * g / sigma2)
    return mp.exp(arg)


def price_ratio_grid(params, logm, tau, r=0.067, q=0.0, trunc_days=True) -> np.ndarray:
    """[N, nT, nK] doubles of the exact ratio.  Same quadrature as ``price_ratio``, with the
    strike-independent factor of every integrand value computed once per (parameter set, maturity,
    node) and multiplied by exp(i phi (dd - sx)) per strike: at 60 digits the re-association is
    immaterial (the two agree to 1e-55, tests/test_oracle.py), and a lattice costs a twentieth."""
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    X, W = laguerre_tables()
    out = np.empty((len(params), len(tau), len(logm)))
    with mp.workdps(DPS):
        for i, prm in enumerate(params):
            kappa, theta, sigma, rho, v0 = (mp.mpf(float(v)) for v in prm)
            series = not (float(prm[2]) > 1e-5)
            for t, tt in enumerate(tau):
                tt = float(tt)
                term_f = float(int(tt * 365.0)) / 365.0 if trunc_days else tt
                if not term_f > 0.0:
                    out[i, t, :] = 0.0
                    continue
                if series and float(prm[2]) == 0.0:
                    out[i, t, :] = np.nan
                    continue
                term = mp.mpf(term_f)
                rr, qq = mp.mpf(float(r)), mp.mpf(float(q))
                Dr, Dq = mp.exp(-rr * term), mp.exp(-qq * term)
                dd = -mp.log(Dr / Dq)
                e1 = [_fj_strike_free(1, mp.mpf(x), kappa, theta, sigma, rho, v0, term, series) for x in X]
                e2 = [_fj_strike_free(2, mp.mpf(x), kappa, theta, sigma, rho, v0, term, series) for x in X]
                for k, lm in enumerate(logm):
                    strike = mp.mpf(float(np.exp(float(lm))))
                    sx = mp.log(strike)
                    s1 = mp.mpf(0)
                    s2 = mp.mpf(0)
                    for x, w, a1, a2 in zip(X, W, e1, e2):
                        ph = mp.expj(mp.mpf(x) * (dd - sx))
                        s1 += mp.mpf(w) * mp.im(a1 * ph) / mp.mpf(x)
                        s2 += mp.mpf(w) * mp.im(a2 * ph) / mp.mpf(x)
                    c = Dq * (s1 / mp.pi + mp.mpf(1) / 2) - strike * Dr * (s2 / mp.pi + mp.mpf(1) / 2)
                    v = c / mp.exp(-rr * mp.mpf(tt))
                    try:
                        out[i, t, k] = float(v)
                    except OverflowError:
                        out[i, t, k] = np.inf if v > 0 else -np.inf
    return out
