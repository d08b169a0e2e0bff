/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the reference's surface-generation
 * path.  Nothing in the shipped package links, loads or calls this file; only tests/, the
 * smoke check and the benchmark's cpu_baseline leg do, as the checker.
 *
 * What is restated, one option at a time exactly as the reference evaluates it:
 *   - heston_model_ql.py:109-149 (price_call) and :151-179 (price_ratio): European call under
 *     Heston with spot 1, strike exp(logm), maturity int(tau*365) days under Actual/365F,
 *     flat continuously-compounded r and q, priced by QuantLib's AnalyticHestonEngine(model)
 *     (heston_model_ql.py:87).  QuantLib (requirements.txt:12, "quantlib>=1.28", unpinned) is a
 *     third-party dependency whose source is not part of the reference tree; its published
 *     algorithm for that constructor is: Gatheral's complex-logarithm formulation of the
 *     characteristic function (Fj_Helper), integrated with a 144-node Gauss-Laguerre rule whose
 *     weights carry exp(x), value = S*Dq*(P1+1/2) - K*Dr*(P2+1/2).
 *   - generate_iv_surface_by_date.py:267-322 (implied-vol inversion, "variant A") and
 *     generate_regime_surfaces.py:106-182 ("variant B"), both solved by SciPy's brentq
 *     (restated below from its published algorithm, scipy/optimize/Zeros/brentq.c) on
 *     [0.001, 3.0] with xtol=1e-6, rtol=4*eps, maxiter=100.
 *   - generate_regime_surfaces.py:188-283 (static / calendar arbitrage checks) and
 *     cvae_model.py:272-285 (static + butterfly rule on price ratios).
 *
 * Parity pin: the price restatement reproduces the QuantLib-evaluated calibration objective
 * values stored by the reference (NIFTY_heston_single_params.pickle, field fit_errors); see
 * tests/test_oracle.py.  Point-wise QuantLib prices are not stored anywhere in the
 * reference, so that aggregate is the only full-precision pin.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "laguerre144.inc"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * Pricing
 * ---------------------------------------------------------------------------------------- */

/* Integrand of P_j at quadrature abscissa phi (QuantLib Fj_Helper::operator(), Gatheral
 * branch, no add-on term).  term = year fraction, dd = log(spot) - log(Dr/Dq), sx = log(K). */
static double fj_gatheral(int j, double phi, double kappa, double theta, double sigma,
                          double rho, double v0, double term, double dd, double sx)
{
    const double sigma2 = sigma * sigma;
    const double rsigma = rho * sigma;
    const double t0 = kappa - ((j == 1) ? rho * sigma : 0.0);
    const double rpsig = rsigma * phi;
    const double complex t1 = t0 + (-rpsig) * I;
    const double complex u = -phi + ((j == 1) ? 1.0 : -1.0) * I;
    const double complex d = csqrt(t1 * t1 - sigma2 * phi * u);
    const double complex ex = cexp(-d * term);
    double complex arg;
    if (sigma > 1e-5) {
        const double complex p = (t1 - d) / (t1 + d);
        const double complex g = clog((1.0 - p * ex) / (1.0 - p));
        arg = v0 * (t1 - d) * (1.0 - ex) / (sigma2 * (1.0 - ex * p))
            + (kappa * theta) / sigma2 * ((t1 - d) * term - 2.0 * g)
            + (phi * (dd - sx)) * I;
    } else {
        const double complex td = phi / (2.0 * t1) * u;
        const double complex p = td * sigma2 / (t1 + d);
        const double complex g = p * (1.0 - ex);
        arg = v0 * td * (1.0 - ex) / (1.0 - p * ex)
            + (kappa * theta) * (td * term - 2.0 * g / sigma2)
            + (phi * (dd - sx)) * I;
    }
    return cimag(cexp(arg)) / phi;
}

/* Call value with spot 1 (heston_model_ql.py:57,75) at strike K, year fraction term. */
static double heston_call_spot1(const double p[5], double r, double q, double strike, double term)
{
    const double kappa = p[0], theta = p[1], sigma = p[2], rho = p[3], v0 = p[4];
    const double spot = 1.0;
    const double Dr = 1.0 / exp(r * term);   /* FlatForward, continuous compounding */
    const double Dq = 1.0 / exp(q * term);
    const double ratio = Dr / Dq;
    const double dd = log(spot) - log(ratio);
    const double sx = log(strike);
    double s1 = 0.0, s2 = 0.0;
    /* GaussianQuadrature::operator() accumulates from the last node down to the first. */
    for (int i = AFIV_NQ - 1; i >= 0; --i) {
        s1 += AFIV_LAG_W[i] * fj_gatheral(1, AFIV_LAG_X[i], kappa, theta, sigma, rho, v0, term, dd, sx);
        s2 += AFIV_LAG_W[i] * fj_gatheral(2, AFIV_LAG_X[i], kappa, theta, sigma, rho, v0, term, dd, sx);
    }
    const double p1 = s1 / M_PI, p2 = s2 / M_PI;
    return spot * Dq * (p1 + 0.5) - strike * Dr * (p2 + 0.5);
}

/* heston_model_ql.py:131: maturity = evaluation date + int(tau*365) days, Actual/365 Fixed.
 * trunc_days=0 prices at tau itself (not what the reference does; kept for diagnostics). */
static double year_fraction(double tau, int trunc_days)
{
    if (!trunc_days) return tau;
    return (double)((long)(tau * 365.0)) / 365.0;
}

/* heston_model_ql.py:151-179.  An already expired option (0 days) has NPV 0 in QuantLib. */
double afo_price_ratio(const double p[5], double r, double q, double logm, double tau, int trunc_days)
{
    const double term = year_fraction(tau, trunc_days);
    const double strike = 1.0 * exp(logm);
    double c = 0.0;
    if (term > 0.0) c = heston_call_spot1(p, r, q, strike, term);
    return c / (1.0 * exp(-r * tau));
}

void afo_price_ratio_grid(const double *params, long n, const double *logm, int nk,
                          const double *tau, int nt, double r, double q, int trunc_days,
                          double *out /* [n][nt][nk] */)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (long s = 0; s < n; ++s)
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < nk; ++k)
                out[(s * nt + t) * nk + k] =
                    afo_price_ratio(params + 5 * s, r, q, logm[k], tau[t], trunc_days);
}

/* ------------------------------------------------------------------------------------------
 * Normal CDF as scipy.special.ndtr evaluates it (Cephes ndtr.c branch structure)
 * ---------------------------------------------------------------------------------------- */
static double ndtr(double a)
{
    if (isnan(a)) return NAN;
    const double sqrth = 7.07106781186547524401E-1;
    const double x = a * sqrth;
    const double z = fabs(x);
    double y;
    if (z < sqrth) {
        y = 0.5 + 0.5 * erf(x);
    } else {
        y = 0.5 * erfc(z);
        if (x > 0) y = 1.0 - y;
    }
    return y;
}
double afo_ndtr(double a) { return ndtr(a); }

/* ------------------------------------------------------------------------------------------
 * Brent's root finder as SciPy's brentq runs it
 * ---------------------------------------------------------------------------------------- */
typedef double (*afo_fn)(double, void *);
enum { AFO_OK = 0, AFO_SIGNERR = -1, AFO_CONVERR = -2, AFO_NANERR = -3 };

static double brentq(afo_fn f, void *ctx, double xa, double xb, double xtol, double rtol,
                     int maxiter, int *err, int *iters)
{
    double xpre = xa, xcur = xb, xblk = 0.0, fpre, fcur, fblk = 0.0, spre = 0.0, scur = 0.0;
    double sbis, delta, stry, dpre, dblk;
    *err = AFO_OK;
    if (iters) *iters = 0;
    fpre = f(xpre, ctx);
    if (isnan(fpre)) { *err = AFO_NANERR; return NAN; }   /* Python wrapper raises on NaN */
    fcur = f(xcur, ctx);
    if (isnan(fcur)) { *err = AFO_NANERR; return NAN; }
    if (fpre == 0) return xpre;
    if (fcur == 0) return xcur;
    if (signbit(fpre) == signbit(fcur)) { *err = AFO_SIGNERR; return NAN; }
    for (int i = 0; i < maxiter; ++i) {
        if (iters) *iters = i + 1;
        if (fpre != 0 && fcur != 0 && (signbit(fpre) != signbit(fcur))) {
            xblk = xpre; fblk = fpre; spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        delta = (xtol + rtol * fabs(xcur)) / 2;
        sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);           /* secant */
            } else {
                dpre = (fpre - fcur) / (xpre - xcur);                    /* inverse quadratic */
                dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            double lim = fmin(fabs(spre), 3 * fabs(sbis) - delta);
            if (2 * fabs(stry) < lim) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis; scur = sbis;
        }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = f(xcur, ctx);
        if (isnan(fcur)) { *err = AFO_NANERR; return NAN; }
    }
    *err = AFO_CONVERR;
    return NAN;
}

/* Test hook: brentq on f(x) = c0 + c1 x + c2 x^2 + c3 x^3 + c4*exp(x), so the restated solver
 * itself can be compared step for step with scipy.optimize.brentq. */
static double poly_fn(double x, void *ctx)
{
    const double *c = (const double *)ctx;
    return c[0] + x * (c[1] + x * (c[2] + x * c[3])) + c[4] * exp(x);
}
double afo_brentq_poly(const double c[5], double a, double b, double xtol, double rtol,
                       int maxiter, int *err, int *iters)
{
    return brentq(poly_fn, (void *)c, a, b, xtol, rtol, maxiter, err, iters);
}

/* ------------------------------------------------------------------------------------------
 * Implied volatility, variants A and B
 * ---------------------------------------------------------------------------------------- */
typedef struct { double F, K, tau, r, q, spot, logm, target; } bs_ctx;

/* generate_iv_surface_by_date.py:308-312 */
static double bs_obj_a(double iv, void *vctx)
{
    const bs_ctx *c = (const bs_ctx *)vctx;
    const double d1 = (log(c->F / c->K) + 0.5 * (iv * iv) * c->tau) / (iv * sqrt(c->tau));
    const double d2 = d1 - iv * sqrt(c->tau);
    const double bs = exp(-c->r * c->tau) * (c->F * ndtr(d1) - c->K * ndtr(d2));
    return bs - c->target;
}

/* generate_regime_surfaces.py:154-162 (log-moneyness enters with the sign the reference uses) */
static double bs_price_b(double sigma, const bs_ctx *c)
{
    if (sigma <= 0 || c->tau <= 0) return 1e10;
    const double d1 = (c->logm + (c->r - c->q + 0.5 * (sigma * sigma)) * c->tau) / (sigma * sqrt(c->tau));
    const double d2 = d1 - sigma * sqrt(c->tau);
    return c->spot * exp(-c->q * c->tau) * ndtr(d1) - c->K * exp(-c->r * c->tau) * ndtr(d2);
}
static double bs_obj_b(double sigma, void *vctx)
{
    const bs_ctx *c = (const bs_ctx *)vctx;
    return bs_price_b(sigma, c) - c->target;
}

/* Forward-form price used when variant-A IVs are re-priced for the arbitrage checks. */
static double bs_price_a(double iv, const bs_ctx *c)
{
    const double d1 = (log(c->F / c->K) + 0.5 * (iv * iv) * c->tau) / (iv * sqrt(c->tau));
    const double d2 = d1 - iv * sqrt(c->tau);
    return exp(-c->r * c->tau) * (c->F * ndtr(d1) - c->K * ndtr(d2));
}

/* One grid point of generate_iv_surface.  variant: 0 = A, 1 = B.  Returns NaN on failure. */
double afo_implied_vol(double price_ratio, double logm, double tau, double spot, double r,
                       double q, int variant, int *iters)
{
    bs_ctx c;
    int err = 0;
    c.tau = tau; c.r = r; c.q = q; c.spot = spot; c.logm = logm;
    const double heston_price = price_ratio * spot * exp(-r * tau);
    c.K = spot * exp(logm);
    c.target = heston_price;
    if (iters) *iters = 0;
    if (variant == 0) {
        c.F = spot * exp((r - q) * tau);
        const double intrinsic = fmax(c.F - c.K, 0.0) * exp(-r * tau);
        if (heston_price < intrinsic * 0.99) return NAN;
        return brentq(bs_obj_a, &c, 0.001, 3.0, 1e-6, 4 * 2.220446049250313e-16, 100, &err, iters);
    } else {
        c.F = 0.0;
        if (heston_price < 1e-10 || isnan(price_ratio)) return NAN;
        double iv = brentq(bs_obj_b, &c, 0.001, 3.0, 1e-6, 4 * 2.220446049250313e-16, 100, &err, iters);
        if (isnan(iv) || iv <= 0 || iv > 3.0) iv = NAN;
        return iv;
    }
}

void afo_implied_vol_grid(const double *ratio, long n, const double *logm, int nk,
                          const double *tau, int nt, double spot, double r, double q,
                          int variant, double *iv /* [n][nt][nk] */)
{
#pragma omp parallel for schedule(dynamic, 1)
    for (long s = 0; s < n; ++s)
        for (int t = 0; t < nt; ++t)
            for (int k = 0; k < nk; ++k) {
                long i = (s * nt + t) * nk + k;
                iv[i] = afo_implied_vol(ratio[i], logm[k], tau[t], spot, r, q, variant, 0);
            }
}

/* ------------------------------------------------------------------------------------------
 * Arbitrage checks
 * flags[4] = { static_violations, calendar_violations, butterfly_violations, status }
 *   static   : generate_regime_surfaces.py:188-223  (price[i] < price[i+1] - 1e-6), -1 if the
 *              surface holds a NaN or non-positive IV (the reference returns None)
 *   calendar : generate_regime_surfaces.py:226-261  (price[i] > price[i+1] + 1e-6), -1 likewise
 *   butterfly: cvae_model.py:272-285 rule on PRICE RATIOS over every consecutive strike triple
 *              of every maturity; a maturity with a NaN or negative ratio is skipped
 *   status   : bit0 is_arbitrage_free as generate_regime_surfaces.py:282 (static==0 &&
 *              calendar==0, surface valid); bit1 butterfly count == 0; bit2 surface valid (no
 *              NaN / <=0 IV); bit3 success as generate_iv_surface_by_date.py:321
 *              (nan_count / points < 0.1); bit4 success as generate_regime_surfaces.py:180
 *              (no NaN at all)
 * bs_form 0 re-prices with the forward form of variant A, 1 with the variant-B form that the
 * reference's own check functions use (generate_regime_surfaces.py:213-215, :251-253).
 * ---------------------------------------------------------------------------------------- */
void afo_arb_check(const double *iv, const double *ratio, long n, const double *logm, int nk,
                   const double *tau, int nt, double spot, double r, double q, int bs_form,
                   int32_t *flags /* [n][4] */)
{
#pragma omp parallel for schedule(static)
    for (long s = 0; s < n; ++s) {
        const double *v = iv + s * nt * nk;
        const double *pr = ratio ? ratio + s * nt * nk : 0;
        int valid = 1, nan_count = 0;
        for (int i = 0; i < nt * nk; ++i) {
            if (isnan(v[i])) { nan_count++; valid = 0; }
            else if (v[i] <= 0) valid = 0;
        }
        int stat = -1, cal = -1, fly = 0;
        if (valid) {
            stat = 0; cal = 0;
            double price[nt * nk];
            for (int t = 0; t < nt; ++t)
                for (int k = 0; k < nk; ++k) {
                    bs_ctx c;
                    c.tau = tau[t]; c.r = r; c.q = q; c.spot = spot; c.logm = logm[k];
                    c.K = spot * exp(logm[k]);
                    c.F = spot * exp((r - q) * tau[t]);
                    c.target = 0;
                    price[t * nk + k] = bs_form == 0 ? bs_price_a(v[t * nk + k], &c)
                                                     : bs_price_b(v[t * nk + k], &c);
                }
            for (int t = 0; t < nt; ++t)
                for (int k = 0; k + 1 < nk; ++k)
                    if (price[t * nk + k] < price[t * nk + k + 1] - 1e-6) stat++;
            for (int k = 0; k < nk; ++k)
                for (int t = 0; t + 1 < nt; ++t)
                    if (price[t * nk + k] > price[(t + 1) * nk + k] + 1e-6) cal++;
        }
        if (pr) {
            for (int t = 0; t < nt; ++t) {
                int skip = 0;
                for (int k = 0; k < nk; ++k)
                    if (isnan(pr[t * nk + k]) || pr[t * nk + k] < 0) skip = 1;
                if (skip) continue;
                for (int k = 0; k + 2 < nk; ++k) {
                    double b = pr[t * nk + k] - 2 * pr[t * nk + k + 1] + pr[t * nk + k + 2];
                    if (b < 0) fly++;
                }
            }
        }
        int status = 0;
        if (valid && stat == 0 && cal == 0) status |= 1;
        if (fly == 0) status |= 2;
        if (valid) status |= 4;
        if ((double)nan_count / (double)(nt * nk) < 0.1) status |= 8;
        if (nan_count == 0) status |= 16;
        flags[4 * s + 0] = stat;
        flags[4 * s + 1] = cal;
        flags[4 * s + 2] = fly;
        flags[4 * s + 3] = status;
    }
}

/* ------------------------------------------------------------------------------------------
 * Whole path for a batch of parameter sets: price -> IV -> checks, one option at a time, the
 * way the reference's Python loops evaluate it.  Used as the benchmark's CPU baseline.
 * ---------------------------------------------------------------------------------------- */
void afo_surface_batch(const double *params, long n, const double *logm, int nk,
                       const double *tau, int nt, double spot, double r, double q, int variant,
                       double *ratio /* [n][nt][nk] */, double *iv /* [n][nt][nk] */,
                       int32_t *flags /* [n][4] */)
{
    afo_price_ratio_grid(params, n, logm, nk, tau, nt, r, q, 1, ratio);
    afo_implied_vol_grid(ratio, n, logm, nk, tau, nt, spot, r, q, variant, iv);
    afo_arb_check(iv, ratio, n, logm, nk, tau, nt, spot, r, q, variant, flags);
}

/* cvae_model.py:245-298: static + butterfly penalties on the 5x2 price-ratio grid.
 * params are the de-normalised Heston parameters (already exp/tanh-transformed, widened to
 * double).  Accumulates sum and count of the collected penalties; the caller takes the mean. */
void afo_arbitrage_penalty(const double *params, long n, double r, double q,
                           double *sum_out, long *count_out)
{
    static const double lm[5] = {-0.1, -0.05, 0.0, 0.05, 0.1};
    static const double taus[2] = {0.5, 1.0};
    double sum = 0.0;
    long count = 0;
    for (long s = 0; s < n; ++s) {
        for (int t = 0; t < 2; ++t) {
            double pr[5];
            int skip = 0;
            for (int k = 0; k < 5; ++k) {
                pr[k] = afo_price_ratio(params + 5 * s, r, q, lm[k], taus[t], 1);
                if (isnan(pr[k]) || pr[k] < 0) skip = 1;
            }
            if (skip) continue;
            for (int j = 0; j < 4; ++j)
                if (pr[j + 1] > pr[j]) { sum += (pr[j + 1] - pr[j]) * 100.0; count++; }
            for (int j = 0; j < 3; ++j) {
                double b = pr[j] - 2 * pr[j + 1] + pr[j + 2];
                if (b < 0) { sum += -b * 100.0; count++; }
            }
        }
    }
    *sum_out = sum;
    *count_out = count;
}

int afo_num_nodes(void) { return AFIV_NQ; }

/* Thread control for the timed CPU baseline (OpenMP over surfaces). */
#ifdef _OPENMP
#include <omp.h>
void afo_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }
int afo_max_threads(void) { return omp_get_max_threads(); }
#else
void afo_set_threads(int n) { (void)n; }
int afo_max_threads(void) { return 1; }
#endif
