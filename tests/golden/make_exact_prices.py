#!/usr/bin/env python3
"""50-digit evaluation of the reference's pricing quadrature (heston_model_ql.py:151-179 through
QuantLib's Gatheral integrand and 144-node Gauss-Laguerre rule) for parameter sets with a small
vol-of-vol, where every double-precision evaluation of that formula carries an absolute noise of
about eps / sigma_v^2.  Writes tests/golden/exact_prices.npz; tests compare the oracle and the
GPU kernel with these values to show on which side of the noise each one sits.

Inputs are taken as the doubles the other implementations see (parameters, year fraction, strike
= exp(logm) rounded to double, quadrature nodes and weights of oracle/laguerre144.inc)."""
import os
import re

import mpmath as mp
import numpy as np

mp.mp.dps = 50
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def laguerre_tables():
    text = open(os.path.join(ROOT, "oracle", "laguerre144.inc")).read()

    def arr(name):
        start = text.index("#define %s" % name)
        vals = []
        for line in text[start:].splitlines()[1:]:
            found = re.findall(r"[-+]?0x[0-9a-fA-F.]+p[-+]?\d+", line)
            vals.extend(float.fromhex(v) for v in found)
            if not line.rstrip().endswith("\\"):
                break
        return vals
    x, w = arr("AFIV_LAG_X_LIST"), arr("AFIV_LAG_W_LIST")
    assert len(x) == 144 and len(w) == 144
    return x, w


def fj(j, phi, kappa, theta, sigma, rho, v0, term, dd, sx):
    sigma2 = sigma * sigma
    t1 = mp.mpc(kappa - (rho * sigma if j == 1 else 0), -rho * sigma * phi)
    u = mp.mpc(-phi, 1 if j == 1 else -1)
    d = mp.sqrt(t1 * t1 - sigma2 * phi * u)
    ex = mp.exp(-d * term)
    p = (t1 - d) / (t1 + d)
    g = mp.log((1 - p * ex) / (1 - p))
    arg = v0 * (t1 - d) * (1 - ex) / (sigma2 * (1 - ex * p)) + kappa * theta / sigma2 * ((t1 - d) * term - 2 * g) \
        + mp.mpc(0, phi * (dd - sx))
    return mp.im(mp.exp(arg)) / phi


def price_ratio(p, r, q, logm, tau, X, W):
    kappa, theta, sigma, rho, v0 = (mp.mpf(float(v)) for v in p)
    term = mp.mpf(float(int(tau * 365.0)) / 365.0)
    strike = mp.mpf(float(np.exp(logm)))
    r, q = mp.mpf(r), mp.mpf(q)
    Dr, Dq = mp.exp(-r * term), mp.exp(-q * term)
    dd = -mp.log(Dr / Dq)
    sx = mp.log(strike)
    s1 = sum(mp.mpf(w) * fj(1, mp.mpf(x), kappa, theta, sigma, rho, v0, term, dd, sx) for x, w in zip(X, W))
    s2 = sum(mp.mpf(w) * fj(2, mp.mpf(x), kappa, theta, sigma, rho, v0, term, dd, sx) for x, w in zip(X, W))
    c = Dq * (s1 / mp.pi + mp.mpf(1) / 2) - strike * Dr * (s2 / mp.pi + mp.mpf(1) / 2)
    return c / mp.exp(-r * mp.mpf(float(tau)))


def small_vol_of_vol():
    X, W = laguerre_tables()
    rng = np.random.default_rng(77)
    sets = []
    for sigma in (1e-3, 2e-3, 5e-3, 1e-2, 3e-2, 1e-1):
        for _ in range(4):
            sets.append([float(np.exp(rng.uniform(np.log(0.05), np.log(8.0)))),      # kappa
                         float(np.exp(rng.uniform(np.log(0.005), np.log(0.2)))),     # theta
                         sigma,
                         float(rng.uniform(-0.95, 0.5)),                             # rho
                         float(np.exp(rng.uniform(np.log(0.005), np.log(0.2))))])    # v0
    params = np.array(sets)
    tau = np.array([0.08333333, 0.5, 2.0])
    logm = np.array([-0.2, -0.1, 0.0, 0.1, 0.2])
    r, q = 0.067, 0.0
    out = np.zeros((len(params), len(tau), len(logm)))
    for i, p in enumerate(params):
        for t, ta in enumerate(tau):
            for k, lm in enumerate(logm):
                out[i, t, k] = float(price_ratio(p, r, q, float(lm), float(ta), X, W))
        print(i, params[i, 2], out[i, 1, 2], flush=True)
    np.savez(os.path.join(HERE, "exact_prices.npz"), params=params, tau=tau, logm=logm, r=r, q=q, ratio=out)


def corners():
    """exact_prices_corners.npz: (a) the corner rows of the calibration box where the
    double-precision oracle is noisiest (large kappa theta T / sigma_v^2), with the day-count
    truncation on and off; (b) the series branch the reference takes at sigma_v <= 1e-5, evaluated
    as written, sigma_v = 0 included (0/0: NaN)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import exact
    r, q = 0.067, 0.0
    tau = np.array([0.02, 0.9, 5.0])            # 7.3, 328.5 and 1825 days
    logm = np.array([-0.2, -0.1, 0.0, 0.1, 0.2])
    main_rows = [[kappa, theta, sigma, rho, v0]
                 for kappa in (2.0, 20.0) for theta in (0.25, 2.0) for sigma in (1e-4, 1e-3)
                 for rho in (-0.999, -0.5, 0.5, 0.999) for v0 in (0.01, 0.25)]
    series_rows = [[kappa, theta, sigma, rho, v0]
                   for sigma in (1e-5, 5e-6, 0.0) for kappa, theta in ((2.0, 0.04), (0.5, 0.09), (20.0, 2.0))
                   for rho, v0 in ((-0.5, 0.04), (0.3, 0.02), (-0.999, 0.25), (0.999, 0.01))]
    params = np.array(main_rows + series_rows)
    out = np.zeros((2, len(params), len(tau), len(logm)))
    for i, p in enumerate(params):
        for ti, trunc in enumerate((True, False)):
            out[ti, i] = exact.price_ratio_grid(p[None], logm, tau, r, q, trunc)[0]
        print(i, p.tolist(), out[0, i, 1, 2], out[1, i, 2, 4], flush=True)
    np.savez(os.path.join(HERE, "exact_prices_corners.npz"), params=params, tau=tau, logm=logm, r=r, q=q,
             ratio_trunc=out[0], ratio_exact_tau=out[1], n_main=len(main_rows))


if __name__ == "__main__":
    import sys
    if "corners" in sys.argv[1:]:
        corners()
    else:
        small_vol_of_vol()
