"""CPU tests of the oracle itself (the checker must be trusted before it checks anything).

* pricing restatement vs the QuantLib-evaluated objective values the reference stored
  (NIFTY_heston_single_params.pickle -> tests/golden/calibration_pin.npz): THE parity pin;
* restated Brent solver vs scipy.optimize.brentq, step counts included;
* C implied vols vs SciPy-evaluated goldens (both variants);
* C arbitrage checks vs the outputs of the reference's own check_* functions;
* NumPy decoder restatement vs torch.nn layers loaded with the checkpoint's weights.
"""
import numpy as np
import pytest

import oracle

T_GRID = np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
LOGM_GRID = np.linspace(-0.2, 0.2, 21)


@pytest.fixture(scope="module")
def pin(golden_dir):
    return np.load(f"{golden_dir}/calibration_pin.npz")


def test_pricing_restatement_reproduces_stored_quantlib_objectives(pin):
    """fit_error = Stage-2 objective (run_single_heston_calibration.py:342-406) or, on fallback,
    the Stage-1 objective (:428-435), evaluated by QuantLib when the reference was run.  All 500
    stored days, split by what each day can show:

    * Stage-1 days -- the stored value is reproduced by the Stage-1 objective (168 prices, RMSE +
      drift + Feller term, no density term) and not by Stage 2: at least 25 days, every one of them
      to 1e-9.  A 1e-9 coincidence on a number of size 1e-2 does not happen by chance, so these
      days pin the price restatement with no amplification in between.
    * Stage-2 days whose `< 0` guard (:270-271) does not trip here: 55 % to 1e-12, 75 % to 1e-9,
      92 % to 1e-6; of those with a calibrated vol-of-vol below 0.01 (a third of them) 99 % to
      1e-9.  The remainder differs by up to 5e-3 of the density-Wasserstein term, which divides
      second differences of prices by dlm^2 = 1/1.5e4 and clips them at zero; those days sit at
      |rho| near the optimiser's bound and are shown not to be a pricing difference by
      test_unreproduced_days_are_not_a_pricing_difference below.
    * days whose guard trips here but did not in QuantLib (deep out-of-the-money prices of 1e-17
      carry a rounding-noise sign; 55 days) and the 5 days that carry the previous day's
      parameters (:537-549) are counted, not judged."""
    r, q = float(pin["r"]), float(pin["q"])
    e1, e2, trip = [], [], []
    for i in range(500):
        args = (pin["params"][i], pin["iv_surfaces"][i], pin["T_grid"], pin["logm_grid"], r, q)
        e1.append(abs(oracle.stage1_objective(*args, guard=False) - pin["fit_errors"][i]))
        e2.append(abs(oracle.stage2_objective(*args, guard=False) - pin["fit_errors"][i]))
        trip.append(oracle.stage1_objective(*args) >= 1e10)
    e1, e2, trip = np.array(e1), np.array(e2), np.array(trip)
    ok = ~pin["failed"]
    stage1_day = ok & (e1 <= 1e-9) & (e2 > 1e-9)
    assert stage1_day.sum() >= 25
    stage2_day = ok & ~stage1_day & ~trip
    assert stage2_day.sum() >= 400
    e = e2[stage2_day]
    assert np.mean(e <= 1e-12) >= 0.55 and np.mean(e <= 1e-9) >= 0.75 and np.mean(e <= 1e-6) >= 0.92
    assert np.median(e) <= 1e-12
    calm = stage2_day & (pin["params"][:, 2] < 0.01)
    assert calm.sum() >= 150 and np.mean(e2[calm] <= 1e-9) >= 0.99
    assert 40 <= (ok & trip).sum() <= 90 and (~ok).sum() == 5


@pytest.mark.parametrize("day,gap", [(18, -2.75e-6), (50, -3.92e-9), (56, 1.32e-5), (102, 1.20e-3), (103, 3.16e-5),
                                     (110, 6.00e-5), (122, 2.79e-4), (268, 8.02e-4), (305, 7.12e-4), (374, -1.22e-5),
                                     (390, 1.24e-3), (400, 1.58e-5), (415, 7.33e-4), (485, -8.77e-6)])
def test_unreproduced_days_are_not_a_pricing_difference(pin, monkeypatch, day, gap):
    """Stage-2 days whose stored fit_error the restatement does not reproduce: fourteen of them,
    from every contiguous run of such days (17-60, 94-128, 266-272, 304-319, 355-408, 413-424,
    475-499), the three largest gaps (days 102, 390, 268) and both signs included.  Day 103, for
    instance, stores 0.0222060 where the restatement gives 0.0221743.  The same objective evaluated from 60-digit prices of the same quadrature equals
    the double-precision oracle's to 1e-13, so neither rounding nor cancellation in the price
    restatement accounts for the gap: the stored value was not taken at the stored parameters
    (run_single_heston_calibration.py:414-424 keeps L-BFGS-B's `fun` and `x`)."""
    from oracle import exact
    i = day
    r, q = float(pin["r"]), float(pin["q"])
    args = (pin["params"][i], pin["iv_surfaces"][i], pin["T_grid"], pin["logm_grid"], r, q)
    double = oracle.stage2_objective(*args, guard=False)
    assert abs((pin["fit_errors"][i] - double) / gap - 1.0) < 0.02
    # two maturities of the density term keep the 60-digit evaluation to a few seconds
    taus = pin["T_grid"][[0, 5]]
    lm, tau, pr = oracle._market_points(pin["iv_surfaces"][i], pin["T_grid"], pin["logm_grid"], r, q)

    def dists():
        return [oracle.density_wasserstein(pin["params"][i], lm[np.abs(tau - t) < 1e-6], t,
                                           pr[np.abs(tau - t) < 1e-6], r, q, guard=False) for t in taus]
    base_d = oracle.stage1_objective(*args, guard=False)
    w_d = dists()
    monkeypatch.setattr(oracle, "price_ratio_grid", exact.price_ratio_grid)
    base_x = oracle.stage1_objective(*args, guard=False)
    w_x = dists()
    assert abs(base_x - base_d) < 1e-13
    assert np.max(np.abs(np.array(w_x) - np.array(w_d))) < 1e-11


def test_factored_60_digit_lattice_equals_the_per_option_evaluation():
    """oracle.exact.price_ratio_grid shares the strike-independent factor of the integrand between
    the strikes of a maturity; at 60 digits that changes nothing a double can see."""
    from oracle import exact
    p = np.array([[2.0, 0.09, 0.3, -0.7, 0.09], [0.5, 0.02, 1e-3, 0.95, 0.01], [2.0, 0.04, 1e-5, -0.5, 0.04]])
    lm, tau = np.array([-0.2, 0.0, 0.13]), [0.08333333, 1.0]
    pts = [(i, t, k) for i in range(3) for t in range(2) for k in range(3)]
    assert np.array_equal(exact.price_ratio_grid(p, lm, tau),
                          exact.price_ratio_points(p, lm, tau, pts).reshape(3, 2, 3))


def test_guarded_objective_matches_on_days_without_noise_sign_flips(pin):
    r, q = float(pin["r"]), float(pin["q"])
    hits = 0
    for i in range(0, 500, 25):
        args = (pin["params"][i], pin["iv_surfaces"][i], pin["T_grid"], pin["logm_grid"], r, q)
        s2 = oracle.stage2_objective(*args)
        if abs(s2 - pin["fit_errors"][i]) <= 1e-10:
            hits += 1
    assert hits >= 6


def test_day_count_truncation_is_reproduced():
    """heston_model_ql.py:131 prices at floor(365 tau)/365 but :176 discounts with tau."""
    p = [2.0, 0.09, 0.3, -0.7, 0.09]
    days = [int(t * 365) for t in T_GRID]
    assert days == [30, 60, 91, 182, 273, 365, 547, 730]
    a = oracle.price_ratio(p, 0.0, 0.08333333)
    b = oracle.price_ratio(p, 0.0, 30 / 365.0, trunc_days=False) \
        * np.exp(-0.067 * 30 / 365.0) / np.exp(-0.067 * 0.08333333)
    assert abs(a - b) < 1e-15


def test_put_call_sanity_and_monotonicity():
    p = [1.5, 0.04, 0.4, -0.6, 0.05]
    pr = oracle.price_ratio_grid(p, LOGM_GRID, T_GRID)[0]
    assert np.all(np.diff(pr, axis=1) < 0)          # calls decrease in strike
    assert np.all(pr > 0) and np.all(pr < np.exp(0.067 * 2.0) + 1e-9)
    # deep ITM at an exactly representable day count: C ~ S - K exp(-rT)
    lm = -0.2
    approx = (1.0 - np.exp(lm) * np.exp(-0.067 * 1.0)) / np.exp(-0.067 * 1.0)
    assert approx - 1e-12 <= pr[5, 0] <= approx + 0.05      # no-arbitrage lower bound, small time value


def test_restated_brent_matches_scipy_step_for_step():
    from scipy.optimize import brentq
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(300):
        c = rng.standard_normal(5)
        c[4] *= 0.1
        f = lambda x: c[0] + x * (c[1] + x * (c[2] + x * c[3])) + c[4] * np.exp(x)   # noqa: E731
        a, b = -1.0 - rng.random(), 1.0 + rng.random()
        x, err, it = oracle.brentq_poly(c, a, b)
        if f(a) * f(b) > 0:
            assert err == -1
            continue
        root, info = brentq(f, a, b, xtol=1e-6, full_output=True)
        assert err == 0
        assert abs(x - root) <= 1e-13 * max(1, abs(root))
        assert it == info.iterations
        checked += 1
    assert checked > 50


def test_ndtr_matches_scipy():
    from scipy.special import ndtr
    x = np.concatenate([np.linspace(-36, 9, 2001), [-0.5, 0.5, 0.0, 1e-300, -1e-300]])
    got = np.array([oracle.lib().afo_ndtr(float(v)) for v in x])
    ref = ndtr(x)
    # the C library's erf/erfc and SciPy's Cephes agree to an ulp of 1.0 absolutely and, in the
    # tails (where the argument's own rounding is amplified by x^2), to 1e-12 relatively
    assert np.all(np.abs(got - ref) <= 2.3e-16)          # one ulp of 1.0 in absolute terms
    assert np.all(np.abs(got - ref) <= 1e-12 * ref)


@pytest.mark.parametrize("variant", ["A", "B"])
def test_c_implied_vol_matches_scipy_goldens(golden_dir, variant):
    g = np.load(f"{golden_dir}/golden_iv.npz")
    ref = g["iv_a"] if variant == "A" else g["iv_b"]
    got = oracle.implied_vol_grid(g["ratio"], g["logm"], g["tau"], float(g["spot"]),
                                  float(g["r"]), float(g["q"]), variant)
    mism = np.isnan(got) != np.isnan(ref)
    m = ~np.isnan(got) & ~np.isnan(ref)
    d = np.abs(got - ref)[m]
    if variant == "B":
        assert not mism.any()
        assert d.max() <= 1e-12
    else:
        # glibc erf/erfc vs SciPy's Cephes differ by an ulp; where vega is ~1e-6 that flips
        # Brent's path (up to 1.5e-6) and one bracket test.  See test_gpu_parity's rule.
        assert mism.sum() <= 2
        assert np.mean(d <= 1e-9) >= 0.995
        assert d.max() <= 5e-6


def test_c_checks_match_reference_function_outputs(golden_dir):
    g = np.load(f"{golden_dir}/golden_flags.npz")
    flags = oracle.arb_check(g["iv"], None, g["logm"], g["tau"], float(g["spot"]),
                             float(g["r"]), float(g["q"]), "B")
    ref = g["result"]
    assert np.array_equal(flags[:, 0], ref[:, 0])
    assert np.array_equal(flags[:, 1], ref[:, 1])
    assert np.array_equal(flags[:, 3] & 1, ref[:, 3])
    assert (ref[:, 0] < 0).sum() > 0 and (ref[:, 0] > 0).sum() > 0


def test_flat_surface_fails_form_b_static_check_with_50_violations():
    """The reference's spot-form d1 uses +log(K/S); a flat 20 % surface then 'violates'."""
    iv = np.full((1, 8, 21), 0.2)
    fb = oracle.arb_check(iv, None, LOGM_GRID, T_GRID, 21000.0, 0.067, 0.0, "B")
    fa = oracle.arb_check(iv, None, LOGM_GRID, T_GRID, 21000.0, 0.067, 0.0, "A")
    assert fb[0, 0] == 50
    assert fa[0, 0] == 0 and fa[0, 1] == 0 and (fa[0, 3] & 1)


def test_numpy_decoder_matches_torch_layers(golden_dir):
    d = np.load(f"{golden_dir}/decoder.npz")
    g = np.load(f"{golden_dir}/golden_decode.npz")
    sd = {k: d[k] for k in d.files if k.startswith("decoder.")}
    got = oracle.decode_params(g["z"], g["c"], sd, d["param_mean"], d["param_std"])
    ref = g["params"]
    for col in (0, 1, 2, 4):
        assert np.all(np.abs(got[:, col] - ref[:, col]) <= 5e-5 * np.abs(ref[:, col]))
    assert np.all(np.abs(got[:, 3] - ref[:, 3]) <= 5e-6)


def test_arbitrage_penalty_is_zero_for_a_regular_model_and_positive_for_noise():
    good = np.array([[2.0, 0.04, 0.3, -0.7, 0.04]])
    assert oracle.arbitrage_penalty(good) == 0.0


def test_ensemble_statistics_ignore_nans():
    s = np.array([[[0.1, np.nan]], [[0.3, 0.2]], [[0.2, 0.4]]])
    mean, med, p5, p95 = oracle.ensemble_stats(s)
    assert np.allclose(mean, [[0.2, 0.3]])
    assert np.allclose(med, [[0.2, 0.3]])
    assert np.allclose(p5, [[0.11, 0.21]])


def test_oracle_noise_against_50_digit_quadrature(golden_dir):
    """The reference formula divides two cancelling quantities by sigma_v^2; evaluated in doubles
    as written (which is what the oracle restates) it is eps / sigma_v^2 away from the exactly
    evaluated quadrature.  This is the conditioning term of the GPU tests' price tolerance."""
    g = np.load(f"{golden_dir}/exact_prices.npz")
    ref = g["ratio"]
    got = oracle.price_ratio_grid(g["params"], g["logm"], g["tau"], float(g["r"]), float(g["q"]))
    sig = g["params"][:, 2][:, None, None]
    eps = np.finfo(float).eps
    assert np.all(np.abs(got - ref) <= 1e-8 * np.maximum(np.abs(ref), 1e-4) + 16 * eps / sig ** 2)
    small = g["params"][:, 2] <= 1e-3
    assert np.abs(got - ref)[small].max() > 1e-12           # the plain bound alone would not hold
    wide = g["params"][:, 2] >= 0.03
    assert np.abs(got - ref)[wide].max() < 1e-12


def test_oracle_noise_against_60_digit_corner_rows(golden_dir):
    """At the corners of the calibration box the same conditioning grows with the drift term:
    the oracle (the reference formula evaluated in doubles as written) is within
    32 eps max(1, kappa theta T) / sigma_v^2 of the exactly evaluated quadrature -- up to 4e-5 at
    sigma_v = 1e-4 -- which is the allowance tests/random_lattices.py gives the ORACLE there; the GPU
    kernel is held to the plain 1e-8 rule against the exact values
    (tests/test_gpu_parity.py::test_price_against_60_digit_quadrature_at_the_corners_and_on_the_series_branch).
    The series form the reference takes at sigma_v <= 1e-5 has no such cancellation: the oracle
    reproduces its exact value to 1e-11, and returns NaN with it at sigma_v = 0 (0 / 0 as written)."""
    g = np.load(f"{golden_dir}/exact_prices_corners.npz")
    p, tau, n_main = g["params"], g["tau"], int(g["n_main"])
    eps = np.finfo(float).eps
    for trunc, key in ((True, "ratio_trunc"), (False, "ratio_exact_tau")):
        ref = g[key]
        got = oracle.price_ratio_grid(p, g["logm"], tau, float(g["r"]), float(g["q"]), trunc)
        main = slice(0, n_main)
        noise = 32 * eps / p[main, 2, None, None] ** 2 * np.maximum(1.0, (p[main, 0] * p[main, 1])[:, None, None] * tau[None, :, None])
        err = np.abs(got - ref)[main]
        assert np.all(err <= 1e-8 * np.maximum(np.abs(ref[main]), 1e-4) + noise)
        assert err.max() > 1e-6                                  # far outside the plain rule: the oracle cannot arbitrate here
        ser = slice(n_main, None)
        fin = np.isfinite(ref[ser])
        assert np.array_equal(np.isfinite(got[ser]), fin)
        assert np.all(np.abs(got[ser] - ref[ser])[fin] <= 1e-11)
        assert not fin[p[ser, 2] == 0.0].any() and fin[p[ser, 2] > 0.0].all()
