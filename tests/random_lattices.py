"""Random lattices and parameter rows for the parity sweep (tests/test_gpu_random_lattices.py,
tools/gpu_fuzz.py): 1..64 strikes and maturities, symmetric / asymmetric / unsorted / repeated
strikes, whole-day / irregular / unsorted / sub-day maturities, day truncation on and off, rates
of either sign, spots over five decades, both Black-Scholes forms, ragged batch sizes,
benchmark-sampler parameters mixed with corner values.  ``run_case`` compares one trial with the
CPU oracle by the rules of tests/test_gpu_parity.py."""
import os

import numpy as np
import torch

import oracle
from test_gpu_parity import sample_params, price_tol, assert_iv_close, dev, oracle_variant

CORNER = dict(kappa=[1e-3, 0.01, 2.0, 20.0], theta=[1e-3, 0.04, 0.25, 2.0], sigma=[1e-5, 1e-4, 1e-3, 1.0, 3.0, 5.0],
              rho=[-0.999, -0.99, -0.95, 0.0, 0.95, 0.999], v0=[1e-3, 0.04, 0.25, 2.0])


def make_case(seed):
    rng = np.random.default_rng(seed)
    nK = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 21, 31, 32, 33, 50, 63, 64, int(rng.integers(1, 65))]))
    nT = int(rng.choice([1, 2, 3, 7, 8, 19, 20, 32, 33, 63, 64, int(rng.integers(1, 65))]))
    kind = rng.integers(0, 5)
    a = float(rng.uniform(0.02, 0.6))
    if kind == 0: logm = np.linspace(-a, a, nK)
    elif kind == 1: logm = np.linspace(-a, float(rng.uniform(0.02, 0.6)), nK)
    elif kind == 2: logm = np.sort(rng.uniform(-a, a, nK))
    elif kind == 3: logm = rng.uniform(-a, a, nK)
    else: logm = np.round(rng.uniform(-a, a, nK), 1)            # repeated strikes
    if nK == 1: logm = np.array([float(rng.uniform(-a, a))])
    tk = rng.integers(0, 5)
    if tk == 0: tau = np.round(np.linspace(rng.integers(5, 60), rng.integers(100, 1500), nT)) / 365.0
    elif tk == 1: tau = np.sort(rng.uniform(0.004, 5.0, nT))
    elif tk == 2: tau = rng.uniform(0.004, 3.0, nT)
    elif tk == 3: tau = (rng.integers(7, 40) + rng.integers(20, 40) * np.arange(nT)) / 365.0   # one increment
    else: tau = np.sort(np.concatenate([rng.uniform(0.0005, 0.01, min(2, nT)), rng.uniform(0.02, 2.0, max(nT - 2, 0))]))[:nT]
    r = float(rng.choice([0.067, 0.0, -0.01, 0.03, 0.12]))
    q = float(rng.choice([0.0, 0.0, 0.02, -0.005, 0.1]))
    spot = float(rng.choice([1.0, 100.0, 7000.0, 21000.0, 1e5]))
    variant = "A" if rng.random() < 0.6 else "B"
    if variant == "A" and seed % 4 == 3:
        variant = "A_BRENT"                                   # the forward form on brentq's own path
    trunc = bool(rng.random() < 0.75)
    per = nT * nK
    N = int(rng.choice([1, 2, 6, 7, 8, 13, 14, 15, 29])) if per > 600 else int(rng.integers(1, 60))
    p = sample_params(N, 1000 + seed)
    for i in range(N):
        if rng.random() < 0.25:
            p[i] = [rng.choice(CORNER[k]) for k in ("kappa", "theta", "sigma", "rho", "v0")]
    return dict(logm=logm, tau=tau, r=r, q=q, spot=spot, variant=variant, trunc=trunc, p=p,
                desc=f"nK={nK} kind={kind} nT={nT} tk={tk} r={r} q={q} spot={spot} {variant} trunc={trunc} N={N}")


def oracle_noise(p, tau):
    """Rounding noise of ANY double-precision evaluation of the reference formula as written (the
    oracle's included): it divides two quantities that are each within sigma_v^2 of a cancellation
    by sigma_v^2, which costs eps / sigma_v^2, times kappa theta T for the drift term
    (tests/test_oracle.py::test_oracle_noise_against_60_digit_corner_rows measures it)."""
    return 32 * np.finfo(float).eps / np.maximum(p[:, 2, None, None], 1e-300) ** 2 * \
        np.maximum(1.0, (p[:, 0] * p[:, 1])[:, None, None] * np.abs(tau)[None, :, None])


def price_bad(got, ref, p, tau, logm, r, q, trunc, max_exact_rows=3):
    """Points whose price leaves the allowance.  The kernel is held to the PLAIN rule
    1e-8 max(|C|, 1e-4) everywhere.  Where the oracle can arbitrate -- its own rounding noise
    (oracle_noise) is below a tenth of that -- it does.  Elsewhere (small sigma_v with large
    kappa theta T, and the series form the reference takes at sigma_v <= 1e-5) the oracle is the
    noisy side: the kernel is compared with a 60-digit evaluation of the same quadrature
    (oracle/exact.py) on up to ``max_exact_rows`` whole strike rows holding such points (a row --
    one parameter set, one maturity -- costs the 60-digit evaluation little more than one of its
    points, since its strikes share the integrand's strike-free factor), still by the plain rule,
    and with the oracle only by the oracle's own noise bound.  Where the formula's value leaves the
    double range both sides must agree on that.  Where it leaves the range of prices -- beyond 1e3
    in spot-1 units, a thousand times anything a call can be worth, which only the series form
    reaches: there the value is the exponential of a sum whose terms reach 1e7, so its relative
    accuracy is the ABSOLUTE accuracy of that sum, eight digits in any double-precision evaluation
    (seed 30081: exact -2.744667e32, kernel 2.6e-8 off, oracle 2.9e-7 off) -- both sides must
    agree that it did, and to 1e-6 relative.  The same holds for every strike of such a row (one
    parameter set, one maturity): all strikes share the integrand, so where some of them come out at
    1e3 .. 1e4 the small values between are remainders of terms that size (seed 90000: a two-day
    maturity whose ratios run from -896 to 6394; the kernel is within 1e-9 of the exact values
    relative to that scale, the oracle within 8e-6) -- they are compared to 1e-6 of the row's
    largest magnitude."""
    from oracle import exact
    tol = price_tol(ref)
    noise = oracle_noise(p, tau) + 1e-4 * np.abs(ref) * (p[:, 2] <= 1e-5)[:, None, None]
    wild = lambda a: ~np.isfinite(a) | (np.abs(a) > 1e300)     # overflowed, or one doubling away from it
    arbitrates = (noise <= 0.1 * tol) & ~wild(ref)
    with np.errstate(invalid="ignore"):
        bad = arbitrates & ~(np.abs(got - ref) <= tol)
        # the noisy remainder: a sanity bound against the oracle ...
        bad |= ~arbitrates & ~(np.abs(got - ref) <= tol + noise) & ~(wild(got) & wild(ref))
    # ... and the plain rule against the exact value at a sample of those points
    idx = np.argwhere(~arbitrates)
    with np.errstate(invalid="ignore"):
        rowmax = np.max(np.where(np.isfinite(ref), np.abs(ref), 0.0), axis=2)      # [N, nT]
    if len(idx):
        rows = np.unique(idx[:, :2], axis=0)
        rows = rows[np.random.default_rng(len(idx)).permutation(len(rows))[:max_exact_rows]]
        pick, ex = [], []
        for i, t in rows:
            exact_row = exact.price_ratio_grid(p[i:i + 1], logm, tau[t:t + 1], r, q, trunc)[0, 0]
            for k in np.flatnonzero(~arbitrates[i, t]):
                pick.append((i, t, k))
                ex.append(exact_row[k])
        for (i, t, k), e in zip(pick, ex):
            g = got[i, t, k]
            if wild(np.array(e)):
                ok = bool(wild(np.array(g)))
            elif abs(e) > 1e3:
                ok = abs(g) > 1e3 and abs(g - e) <= 1e-6 * abs(e)
            elif rowmax[i, t] > 1e3:
                ok = abs(g - e) <= 1e-6 * rowmax[i, t]
            else:
                ok = abs(g - e) <= 1e-8 * max(abs(e), 1e-4)
            if not ok:
                bad[i, t, k] = True
    return bad


def run_case(afiv, c):
    logm, tau, r, q, spot, v, p = c["logm"], c["tau"], c["r"], c["q"], c["spot"], c["variant"], c["p"]
    g = afiv.Grid(logm, tau, r, q, trunc_days=c["trunc"])
    ref = oracle.price_ratio_grid(p, logm, tau, r, q, c["trunc"])
    got = g.price(dev(p)).cpu().numpy()
    bad = price_bad(got, ref, p, tau, logm, r, q, c["trunc"])
    assert not bad.any(), f"price: {bad.sum()} points, worst {np.nanmax(np.abs(got - ref)[bad]):.3g}"
    iv, fl = g.implied_vol_checked(dev(ref), spot, v)
    iv, fl = iv.cpu().numpy(), fl.cpu().numpy()
    iv_o = oracle.implied_vol_grid(ref, logm, tau, spot, r, q, oracle_variant(v))
    if v != "B":
        assert_iv_close(iv, iv_o, logm, tau, spot, r, q, same_path=v == "A_BRENT")
    else:
        # the spot form is solved on brentq's own path: over the 91 spot-form trials of the fixed seed
        # range (860 000 roots) no NaN mask differs and no root is further than 1e-9 from the oracle's
        mism = np.isnan(iv) != np.isnan(iv_o)
        assert not mism.any(), f"variant B NaN mask differs at {mism.sum()} points"
        m = ~np.isnan(iv) & ~np.isnan(iv_o)
        d = np.abs(iv[m] - iv_o[m])
        if os.environ.get("AFIV_IV_RULE_REPORT"):
            with open(os.environ["AFIV_IV_RULE_REPORT"], "a") as fh:
                fh.write(f"random_lattices.py:B desc={c['desc'].replace(' ', '_')[:60]} points={iv.size} mism={int(mism.sum())} roots={d.size} "
                         f"beyond_1e-6={int(np.sum(d > 1e-6))} beyond_1e-9={int(np.sum(d > 1e-9))} "
                         f"max={float(d.max()) if d.size else 0.0:.3g}\n")
        assert d.size == 0 or d.max() <= 1e-8, f"variant B: {np.sum(d > 1e-8)} of {d.size} roots differ, worst {d.max():.3g}"
    fl_o = oracle.arb_check(iv, ref, logm, tau, spot, r, q, oracle_variant(v))
    assert np.array_equal(fl, fl_o), f"flags differ in {np.sum(np.any(fl != fl_o, axis=1))} surfaces"
    iv2, fl2, ra2 = g.surface(dev(p), spot, v)
    iv3, fl3 = g.implied_vol_checked(ra2, spot, v)
    assert torch.equal(torch.nan_to_num(iv2, nan=-1.0), torch.nan_to_num(iv3, nan=-1.0)) and torch.equal(fl2, fl3), \
        "afiv_surface differs from price + implied_vol_checked"
    g.close()
