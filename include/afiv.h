/*
 * afiv -- MI355X (gfx950) kernels for arbitrage-checked Heston implied-volatility surfaces.
 *
 * C ABI of libafiv.so.  The reference project is pure Python with no FFI of its own, so each
 * entry point below names the Python routine whose arithmetic it replaces; the Python drop-in
 * modules bind these symbols with ctypes.
 *
 * Conventions
 *   - "host" pointers are ordinary CPU memory and are only read during the call.
 *   - "device" pointers are HIP device memory owned by the caller (for example the data_ptr()
 *     of a torch tensor on the current device).  Execution entry points enqueue work on
 *     `stream` (a hipStream_t passed as void*, NULL = the default stream) and return without
 *     synchronising; they allocate nothing.
 *   - Return value 0 = success, negative = AFIV_E_* below.  Per-point numerical failures are
 *     data (NaN implied vols, flag words), never errors, as in the reference's try/except -> NaN
 *     (generate_iv_surface_by_date.py:316-318).
 *   - All arrays are dense, row-major, in the shapes written next to each argument.
 *   - Concurrency: grid and decoder objects are immutable after creation and the library keeps no
 *     other state, so calls may be issued from any thread on any stream, on shared objects too.
 *     Two calls use scratch memory that is not the caller's and must therefore not overlap with
 *     another call using the same scratch: afiv_surface with ratio_out == NULL (the grid's
 *     workspace) and afiv_surface_static (one array per GPU).
 */
#ifndef AFIV_H
#define AFIV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFIV_VERSION 200

enum {
    AFIV_OK = 0,
    AFIV_E_NULL = -1,        /* required pointer is NULL */
    AFIV_E_SHAPE = -2,       /* a size is out of the supported range */
    AFIV_E_HIP = -3,         /* a HIP runtime call, kernel launch or host allocation failed */
    AFIV_E_ARG = -4,         /* bad enum / flag value, lattice value that is not finite, object of another device */
    AFIV_E_WORKSPACE = -5,   /* the call needs a workspace the grid was created without */
    AFIV_E_SCRATCH = -6      /* the batch exceeds what one afiv_surface_static call passes through its scratch */
};

/* Which Black-Scholes form is inverted / re-priced, and for the forward form how its root is
 * found.  The forward form's objective increases with the vol, so every convergent method lands
 * within the reference's xtol = 1e-6 of scipy.optimize.brentq's answer: AFIV_VARIANT_A uses a
 * closed-form first guess and Householder steps (one or two objective evaluations per point),
 * AFIV_VARIANT_A_BRENT walks brentq's own iteration path (about ten).  Both apply brentq's rules at
 * the bracket ends with the same arithmetic, so their NaN masks are identical.  The spot form's
 * objective is not monotone (the reference's sign convention), so it is always solved on brentq's
 * path.  Only AFIV_VARIANT_A and AFIV_VARIANT_B are valid as a `bs_form`. */
enum {
    AFIV_VARIANT_A = 0,        /* forward form, generate_iv_surface_by_date.py:308-312 */
    AFIV_VARIANT_B = 1,        /* spot form as written in generate_regime_surfaces.py:154-159 */
    AFIV_VARIANT_A_BRENT = 2   /* forward form on brentq's iteration path (same decisions up to last-bit rounding) */
};

/* Limits of this build. */
#define AFIV_MAX_STRIKES 64      /* strikes per grid */
#define AFIV_MAX_MATURITIES 64   /* maturities per grid */
#define AFIV_ENSEMBLE_ONE_SWEEP 16384  /* afiv_ensemble_stats: up to this many surfaces a column is sorted in LDS
                                        * after one sweep; larger ensembles take three or four sweeps */
#define AFIV_STATIC_SCRATCH_DOUBLES (16 * 1024 * 1024)   /* afiv_surface_static: the array the ratios pass through (128 MiB) */
#define AFIV_STATIC_MAX_PIECES 1024                     /* ... and how many array-fulls one call may enqueue */

typedef struct afiv_grid afiv_grid;
typedef struct afiv_decoder afiv_decoder;

int afiv_version(void);
const char *afiv_error_string(int code);

/* ---- grid: the (maturity, strike) lattice and everything that depends only on it ----------
 * Replaces the per-call constants of heston_model_ql.py:109-179: strike = exp(logm) (:171),
 * maturity date = evaluation date + int(tau*365) days under Actual/365F (:131; pass
 * trunc_days = 1 for that behaviour, 0 to price at tau itself), flat-forward discount factors
 * for r and q (:60-69) and the untruncated discount exp(-r*tau) of :176.  Also holds the
 * strike-dependent cos/sin tables of the 144-node Gauss-Laguerre rule.
 * logm_host[nK], tau_host[nT] are host arrays.  workspace_surfaces > 0 reserves device memory
 * for that many surfaces' price ratios so afiv_surface may be called with ratio_out == NULL.
 * Must be called with the target device current; the grid is bound to that device. */
int afiv_grid_create(afiv_grid **out, const double *logm_host, int nK, const double *tau_host,
                     int nT, double r, double q, int trunc_days, int64_t workspace_surfaces);
int afiv_grid_destroy(afiv_grid *g);

/* Which of the pricing kernel's shortcuts the lattice qualifies for (no counterpart in the
 * reference; for tests and for callers who want to know why one grid prices faster than another).
 * half_strike = 1: the strikes are symmetric about their midpoint (any linspace(a, b, n), n >= 4),
 * so half of them carry the matrix work.  stepped = 0, 1, 2, 3: whole-day maturities a constant
 * number of days apart (1), or a base step plus 0..3 days (2), advance exp(-d T) by complex
 * multiplications instead of an exp and a sincos per maturity; with trunc_days = 0, evenly spaced
 * year fractions (a linspace) do the same (1); 3: whole-day maturities of which at least a quarter
 * are sums of two earlier ones (the reference's own 30, 60, 91, 182, 273, 365, 547, 730 days), whose
 * exponentials follow from those by one complex product; 0: none of these.  Modes 1 and 2 take 4 to
 * 32 maturities.
 * Either pointer may be NULL.  Results do not depend on these beyond rounding. */
int afiv_grid_traits(const afiv_grid *g, int *half_strike, int *stepped);

/* ---- pricing: replaces HestonModelQL.price_ratio (heston_model_ql.py:151-179) evaluated over
 * the whole grid, i.e. QuantLib AnalyticHestonEngine(model) (:87) per option.
 * params: device [N,5] doubles (kappa, theta, sigma_v, rho, v0); ratio_out: device [N,nT,nK]. */
int afiv_price(const afiv_grid *g, const double *params, int64_t N, double *ratio_out,
               void *stream);

/* ---- implied volatility: replaces the inner loop body of generate_iv_surface
 * (generate_iv_surface_by_date.py:296-318 for AFIV_VARIANT_A, generate_regime_surfaces.py:138-177
 * for AFIV_VARIANT_B): the reference's guard, then the root of the Black-Scholes objective on
 * [0.001, 3.0] under scipy.optimize.brentq's rules -- f(0.001) == 0 returns 0.001, equal signs at
 * the two ends or a NaN objective give NaN -- so NaN where the reference yields NaN.  Inside a
 * valid bracket AFIV_VARIANT_B and AFIV_VARIANT_A_BRENT make brentq's decisions one for one
 * (xtol 1e-6, rtol 4 eps, 100 iterations; the arithmetic of a step differs from brentq.c's in the
 * last bits: one quotient for the inverse-quadratic step, fused multiply-adds in the tolerance),
 * which lands on brentq's own root to 1e-10 on 99.9 % of points; AFIV_VARIANT_A converges to the
 * root itself and is within 5e-7 of brentq's answer wherever the vega is above 1e-3 of the spot.
 * ratio: device [N,nT,nK]; iv_out: device [N,nT,nK]. */
int afiv_implied_vol(const afiv_grid *g, const double *ratio, int64_t N, double spot,
                     int variant, double *iv_out, void *stream);

/* ---- arbitrage checks: replaces check_static_arbitrage / check_calendar_arbitrage /
 * check_all_arbitrage (generate_regime_surfaces.py:188-283) and the static + butterfly rule of
 * compute_arbitrage_penalty (cvae_model.py:272-285) applied to every strike triple.
 * iv: device [N,nT,nK]; ratio: device [N,nT,nK] or NULL (then butterfly count = 0);
 * flags_out: device [N,4] int32 = { static_violations, calendar_violations,
 * butterfly_violations, status }.  static/calendar are -1 when the surface holds a NaN or
 * non-positive IV (the reference returns None).  status bits: 1 = arbitrage-free as
 * generate_regime_surfaces.py:282; 2 = no butterfly violation; 4 = surface valid;
 * 8 = success as generate_iv_surface_by_date.py:321 (NaN share < 10 %); 16 = no NaN at all
 * (generate_regime_surfaces.py:180).  bs_form selects the re-pricing formula (AFIV_VARIANT_*).
 * mask_out: NULL, or device [N,nT,nK] bytes receiving where the violations sit, i.e. the
 * (mat_idx, strike_idx) lists the reference functions return: bit 1 = static violation between
 * strikes k and k+1, bit 2 = calendar violation between maturities t and t+1, bit 4 = negative
 * butterfly on strikes k..k+2.  Static/calendar bits are only set on valid surfaces. */
int afiv_arb_check(const afiv_grid *g, const double *iv, const double *ratio, int64_t N,
                   double spot, int bs_form, int32_t *flags_out, uint8_t *mask_out, void *stream);

/* ---- implied vol and checks in one pass over the lattice: afiv_implied_vol followed by
 * afiv_arb_check(bs_form = variant, no mask), but each vol's re-priced Black-Scholes value is
 * taken from the root finder instead of being recomputed: its last objective evaluation, or, where
 * AFIV_VARIANT_A returns the vol a small final step lands on, the target price itself (within
 * vega x 1e-9 of that vol's value).  The vols are identical to afiv_implied_vol's; the flag words are
 * identical to the two-call sequence's except where a price difference sits that close to its 1e-6
 * threshold (rounding, 1e-11, on the Brent-path variants: the two evaluate the same formula with
 * different roundings -- one shared exponential and a reciprocal-based quotient here, separate
 * exponentials and an IEEE quotient there).  Replaces the body of generate_iv_surface plus check_all_arbitrage
 * (generate_regime_surfaces.py:320-335 calls them back to back). */
int afiv_implied_vol_checked(const afiv_grid *g, const double *ratio, int64_t N, double spot,
                             int variant, double *iv_out, int32_t *flags_out, void *stream);

/* ---- whole surface: price -> implied vol -> checks for N parameter sets, i.e. the body of the
 * per-sample loop generate_iv_surface_by_date.py:464-468 plus the checks above.
 * ratio_out may be NULL if the grid has a workspace (the batch is then processed in
 * workspace-sized chunks); iv_out and flags_out are required. */
int afiv_surface(const afiv_grid *g, const double *params, int64_t N, double spot, int variant,
                 double *ratio_out, double *iv_out, int32_t *flags_out, void *stream);

/* ---- whole surface, for callers that cannot allocate: afiv_surface with ratio_out == NULL, but
 * the price ratios pass through a fixed 128 MiB array inside the library instead of a workspace the
 * grid would have had to allocate, so neither the grid nor the caller provides memory for them and
 * the call itself calls no hipMalloc / hipFree or allocating routine.  No counterpart in the
 * reference: it exists for code under HIP graph capture, tight serving loops and harnesses that
 * forbid allocation.  Same results as afiv_surface, bit for bit.
 * A batch whose ratios fit the array (N * nT * nK <= AFIV_STATIC_SCRATCH_DOUBLES: 16777 surfaces of
 * 20 x 50, 99864 of 8 x 21) is one pair of launches; a larger one passes through the array in
 * pieces -- a pair of launches each, in stream order, so a million surfaces of 20 x 50 need 8.4 GB
 * for their vols and nothing for their ratios -- up to AFIV_STATIC_MAX_PIECES pieces per call,
 * beyond which the call enqueues nothing and returns AFIV_E_SCRATCH.  Pieces cost about 3 %
 * of the one-launch rate at 20 x 50 (each ends with a pipeline drain and a launch gap; a 32 MiB array
 * cost 12 %), which is what afiv_surface with a workspace of 131 072 surfaces costs as well.  Every call on a GPU
 * shares that GPU's array (each GPU holds its own copy until the process ends): calls that overlap
 * on one GPU -- different streams or threads -- race on it, so issue them in one stream's order.
 * Nothing in the array has to be cleared between calls.  The array's device address is looked up
 * when a grid is created (afiv_grid_create, with that GPU current), so a first call
 * issued inside a stream capture finds it resolved. */
int afiv_surface_static(const afiv_grid *g, const double *params, int64_t N, double spot, int variant,
                        double *iv_out, int32_t *flags_out, void *stream);

/* ---- decoder: replaces ConditionalVAE_SingleHeston.decode (cvae_model.py:145-160, decoder
 * stack :84-105 in eval mode) followed by the de-normalisation and exp/tanh transforms of
 * generate_iv_surface_by_date.py:441-446.  Host arrays follow the checkpoint's state-dict
 * layout: linear weights [out,in] row-major, BatchNorm weight/bias/running_mean/running_var. */
typedef struct afiv_decoder_weights {
    const float *lin_w[4];   /* decoder.0/4/8/12.weight : [32,12] [64,32] [128,64] [5,128] */
    const float *lin_b[4];   /* decoder.0/4/8/12.bias */
    const float *bn_w[3];    /* decoder.1/5/9.weight */
    const float *bn_b[3];    /* decoder.1/5/9.bias */
    const float *bn_mean[3]; /* decoder.1/5/9.running_mean */
    const float *bn_var[3];  /* decoder.1/5/9.running_var */
    const float *param_mean; /* [5] */
    const float *param_std;  /* [5] */
    float bn_eps;            /* 1e-5 */
} afiv_decoder_weights;

int afiv_decoder_create(afiv_decoder **out, const afiv_decoder_weights *w_host);
int afiv_decoder_destroy(afiv_decoder *d);

/* z: device [N,4] float; c: device [c_rows,8] float with c_rows == N or 1 (one row broadcast,
 * as cvae_model.py:365-366).  Outputs (either may be NULL): normalized_out [N,5] float = raw
 * decoder output; params32_out [N,5] float = transformed parameters; params64_out [N,5] double =
 * the same values widened (what generate_iv_surface receives, :286). */
int afiv_decode(const afiv_decoder *d, const float *z, const float *c, int64_t N, int64_t c_rows,
                float *normalized_out, float *params32_out, double *params64_out, void *stream);

/* ---- training-time arbitrage penalty: replaces compute_arbitrage_penalty
 * (cvae_model.py:208-298).  params: device [B,5] doubles, already de-normalised and
 * transformed; g must be the 5-strike x 2-maturity grid of cvae_model.py:245-246 (any grid
 * works: the rule is applied per maturity).  scratch: device [B*nT*nK] doubles;
 * acc_out: device [2] doubles = { sum of penalties, number of penalties }. */
int afiv_arbitrage_penalty(const afiv_grid *g, const double *params, int64_t B, double *scratch,
                           double *acc_out, void *stream);

/* ---- calibration Stage-1 objective: replaces objective_stage1
 * (run_single_heston_calibration.py:252-287) for C candidate parameter sets against D days.
 * model_ratio: device [C,nT,nK] (from afiv_price); market_ratio: device [D,nT,nK] (NaN marks
 * points the reference drops at :185); params: device [C,5]; loss_out: device [C,D]. */
int afiv_stage1_objective(const afiv_grid *g, const double *model_ratio, const double *params,
                          int64_t C, const double *market_ratio, int64_t D, double drift_weight,
                          double *loss_out, void *stream);

/* ---- calibration Stage-2 density term: replaces compute_density_wasserstein
 * (run_single_heston_calibration.py:100-151) for C candidates against D days, per maturity.
 * g is the FINE grid: its strikes are the n_points log-moneyness values of :107 (at least 3,
 * equally spaced), its maturities those of the surface.  model_ratio: device [C,nT,nK] from
 * afiv_price on g; market_ratio: device [D,nT,nK] = the day's market ratios interpolated to the
 * fine strikes (:110), a maturity the reference skips (< 3 market points, :375-377) filled with
 * NaN.  scratch: device [(C+D)*nT*(nK+1)] doubles.  dist_out: device [C,D,nT] = the Wasserstein
 * distance, +inf where the reference returns inf (:116-119, :139-140), NaN for skipped
 * maturities; objective_stage2 (:342-406) averages the finite entries of a row. */
int afiv_density_wasserstein(const afiv_grid *g, const double *model_ratio, int64_t C,
                             const double *market_ratio, int64_t D, double *scratch,
                             double *dist_out, void *stream);

/* ---- ensemble statistics: replaces generate_iv_surface_by_date.py:474-478, i.e.
 * np.nanmean, np.nanmedian, np.nanpercentile(5) and np.nanpercentile(95) over axis 0 of the
 * stacked valid surfaces (NumPy's default "linear" percentile method, bit for bit).
 * iv: device [N,points] with points = nT*nK; select: NULL (use every surface) or device [N]
 * bytes, nonzero = use (the reference keeps the surfaces whose success flag is set, :466-467);
 * stats_out: device [4,points] = mean, median, 5th, 95th percentile; a grid point without any
 * non-NaN value gives NaN.  1 <= N <= 2^31, points >= 1; no scratch memory whatever N is (large
 * ensembles are read several times instead: see AFIV_ENSEMBLE_ONE_SWEEP).  Needs no grid object;
 * runs on the current device. */
int afiv_ensemble_stats(const double *iv, const uint8_t *select, int64_t N, int64_t points,
                        double *stats_out, void *stream);

/* The same block for `groups` consecutive ensembles of N surfaces each in one call -- the per-date
 * statistics of a batch over dates, where the reference runs its script (and that block) once per
 * date: iv device [groups*N, points], select NULL or device [groups*N], stats_out device
 * [groups,4,points].  Every group's statistics are those afiv_ensemble_stats gives for it alone. */
int afiv_ensemble_stats_groups(const double *iv, const uint8_t *select, int64_t groups, int64_t N,
                               int64_t points, double *stats_out, void *stream);

/* ---- diagnostics: the kernels' own FP64 elementary functions, element-wise, so that unit tests
 * can check them against high-precision references.  No counterpart in the reference (which
 * calls libm / SciPy).  x, y, out0, out1: device [n]; y only for AFIV_MATH_ATAN2 (atan2(x, y));
 * out1 only for AFIV_MATH_SINCOS (out0 = sin, out1 = cos), AFIV_MATH_EXPM1, AFIV_MATH_COSM1
 * and AFIV_MATH_EXPM1_TAB; may be NULL. */
enum {
    AFIV_MATH_EXP = 0, AFIV_MATH_SINCOS = 1, AFIV_MATH_LOG1P = 2, AFIV_MATH_ATAN2 = 3,
    AFIV_MATH_RCP = 4, AFIV_MATH_NDTR = 5,
    AFIV_MATH_EXPM1 = 6,  /* out0 = exp(x) - 1, out1 = exp(x) */
    AFIV_MATH_COSM1 = 7,  /* out0 = cos(x) - 1, out1 = cos(x) */
    AFIV_MATH_EXP_TAB = 8,   /* the pricing kernel's table-driven exponential */
    AFIV_MATH_EXPM1_TAB = 9  /* its exp(x) - 1 companion: out0 = exp(x) - 1, out1 = exp(x) */
};
int afiv_debug_math(int op, const double *x, const double *y, int64_t n, double *out0, double *out1,
                    void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AFIV_H */
