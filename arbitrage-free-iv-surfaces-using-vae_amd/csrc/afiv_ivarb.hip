// Implied-volatility inversion and arbitrage checks on gfx950.
//
// Compiled with -ffp-contract=off: these kernels reproduce NumPy/SciPy evaluation order so that
// guards, bracket signs and violation counts agree with the Python reference bit for bit on
// identical inputs (mul and add are never fused here).
//
//   implied vol : generate_iv_surface_by_date.py:296-318 (variant A) and
//                 generate_regime_surfaces.py:138-177 (variant B); the root finder follows
//                 scipy.optimize.brentq step for step (one grid point per lane at a time), because
//                 variant B's objective is not monotone and only the same iteration path
//                 lands on the same root.
//   checks      : generate_regime_surfaces.py:188-283 and cvae_model.py:272-285.
#include "afiv_internal.h"
#include "afiv_math.h"
#include "erfcx_table.inc"

namespace {

// Normal CDF through one scaled complementary error function: N(a) = erfc(-a/sqrt 2)/2 with
// erfc(u) = exp(-u^2) erfcx(u), erfcx from a 64-segment degree-8 table in y = 1/(1+u) (see
// tools/gen_erfcx_table.py; 1.5 ulp).  One code path for every lane -- the C library's erf/erfc
// pick one of five ranges per argument, which a wave executes back to back when its lanes
// disagree.  The table lives in LDS (ERFCX_LDS doubles, filled by load_erfcx_table).
__constant__ double c_erfcx[AFIV_ERFCX_NSEG * AFIV_ERFCX_NCOEF] = {AFIV_ERFCX_TABLE};
constexpr int ERFCX_ROW = 10;                       // 9 coefficients padded to 80 B rows
constexpr int ERFCX_LDS = AFIV_ERFCX_NSEG * ERFCX_ROW;
static_assert(AFIV_ERFCX_NCOEF == 9 && AFIV_ERFCX_NSEG == 64, "erfcx table shape");

__device__ __forceinline__ void load_erfcx_table(double *lds_tab)
{
    for (int i = threadIdx.x; i < ERFCX_LDS; i += blockDim.x) {
        const int seg = i / ERFCX_ROW, j = i - seg * ERFCX_ROW;
        lds_tab[i] = j < AFIV_ERFCX_NCOEF ? c_erfcx[seg * AFIV_ERFCX_NCOEF + j] : 0.0;
    }
}

template <bool NANSAFE = true>
__device__ __forceinline__ double ndtr_dev(double a, const double *__restrict__ tab)
{
    const double sqrth = 7.07106781186547524401E-1;
    const double x = a * sqrth;
    const double u = fmin(fabs(x), 40.0);           // erfc(40) = 0 in double; also maps NaN to a finite dummy
    const double t = 64.0 * afivm::rcp_f64(1.0 + u);
    const double tc = fmin(t, 0x1.fffffffffffffp5);      // t = 64 (u = 0) belongs to the last segment
    const int seg = (int)tc;
    const double f = __builtin_amdgcn_fract(tc) - 0.5;
    const double *c = tab + seg * ERFCX_ROW;
    double p = c[8];
    p = fma(p, f, c[7]);
    p = fma(p, f, c[6]);
    p = fma(p, f, c[5]);
    p = fma(p, f, c[4]);
    p = fma(p, f, c[3]);
    p = fma(p, f, c[2]);
    p = fma(p, f, c[1]);
    p = fma(p, f, c[0]);
    const double s = u * u;
    const double lo = fma(u, u, -s);                // exact low part of the square
    double e = NANSAFE ? afivm::exp_f64<false>(-s) : afivm::exp_core_f64<false>(-s);   // s <= 1600
    e = fma(-e, lo, e);
    const double h = 0.5 * e * p;                   // erfc(|x|)/2
    const double r = x > 0 ? 1.0 - h : h;
    return NANSAFE && x != x ? x : r;               // NaN in, NaN out (callers that never pass NaN opt out)
}

// Per-point constants of the Black-Scholes objective.
struct BsPoint {
    double lnFK;      // log(F/K)                       (variant A)
    double F, K;      // spot*exp((r-q)tau), spot*exp(logm)
    double disc;      // exp(-r tau)
    double sdq;       // spot*exp(-q tau)               (variant B)
    double kdr;       // K*exp(-r tau)                  (variant B)
    double logm, rq;  // logm, r - q                    (variant B)
    double tau, sqrt_tau;
    double target;    // heston price
    double ratio;     // exp(-d2^2/2) / exp(-d1^2/2): F/K (variant A), exp(logm + (r-q) tau) (variant B)
};

// n / d through the hardware reciprocal, two Newton steps and one residual correction: the
// correctly rounded quotient in all but rare last-bit cases, at half the cost of the IEEE
// expansion.  d = 0 gives NaN where IEEE gives inf or NaN; every use below only compares the
// result with "<", which is false for both.
__device__ __forceinline__ double fdivq(double n, double d)
{
    const double r = afivm::rcp_f64(d);
    const double q = n * r;
    const double e = __builtin_fma(-d, q, n);
    return __builtin_fma(e, r, q);
}

typedef double d2v __attribute__((ext_vector_type(2)));

// Segment and local coordinate of erfcx's table for t = 64/(1 + u) (see ndtr_dev).
__device__ __forceinline__ const d2v *erfcx_row(double t, const double *__restrict__ tab, double *f)
{
    const double tc = fmin(t, 0x1.fffffffffffffp5);      // t = 64 (u = 0) belongs to the last segment
    const int seg = (int)tc;
    *f = __builtin_amdgcn_fract(tc) - 0.5;
    // 24-bit multiply: full rate, where a 32-bit v_mul_lo_u32 runs at a quarter of it
    return reinterpret_cast<const d2v *>(tab + __mul24(seg, ERFCX_ROW));   // rows are 80 bytes: 16-byte aligned
}

__device__ __forceinline__ double erfcx_horner(d2v q0, d2v q1, d2v q2, d2v q3, double q4, double f)
{
    double p = q4;
    p = fma(p, f, q3.y);
    p = fma(p, f, q3.x);
    p = fma(p, f, q2.y);
    p = fma(p, f, q2.x);
    p = fma(p, f, q1.y);
    p = fma(p, f, q1.x);
    p = fma(p, f, q0.y);
    return fma(p, f, q0.x);
}

// The root finder's objective (without the target): the Black-Scholes value at a vol from the
// bracket (finite, > 0: no NaN handling, no IEEE special cases in the quotient).  Both normal
// CDFs share ONE exponential: exp(-d2^2/2) = exp(-d1^2/2) c.ratio, the identity
// F phi(d1) = K phi(d2) of the formula itself, so N(d2) costs its erfcx polynomial only.  The
// ten table reads of the two polynomials are issued together, ahead of the exponential, so that
// their LDS latency is paid once and under that arithmetic (left to the compiler they are sunk
// next to their uses, one wait per coefficient pair).
template <int VARIANT>
__device__ __forceinline__ double bs_objective(const BsPoint &c, double iv, const double *__restrict__ tab,
                                               const afivm::ExpCoef &ec)
{
    const double sqrth = 7.07106781186547524401E-1;
    double num;
    if (VARIANT == AFIV_VARIANT_A) num = c.lnFK + 0.5 * (iv * iv) * c.tau;
    else num = c.logm + (c.rq + 0.5 * (iv * iv)) * c.tau;
    const double den = iv * c.sqrt_tau;
    const double d1 = fdivq(num, den);
    const double d2 = d1 - den;
    const double x1 = d1 * sqrth, x2 = d2 * sqrth;
    const double u1 = fabs(x1), u2 = fabs(x2);
    // both table coordinates from one reciprocal: 64/(1+u1) = 64 (1+u2) / ((1+u1)(1+u2))
    const double w1 = 1.0 + u1, w2 = 1.0 + u2;
    const double r64 = 64.0 * afivm::rcp_f64(w1 * w2);
    double f1, f2;
    const d2v *r1 = erfcx_row(r64 * w2, tab, &f1);
    const d2v *r2 = erfcx_row(r64 * w1, tab, &f2);
    const d2v a0 = r1[0], a1 = r1[1], a2 = r1[2], a3 = r1[3];
    const double a4 = reinterpret_cast<const double *>(r1)[8];
    const d2v b0 = r2[0], b1 = r2[1], b2 = r2[2], b3 = r2[3];
    const double b4 = reinterpret_cast<const double *>(r2)[8];
    __builtin_amdgcn_sched_barrier(0);
    const double s = u1 * u1;
    const double lo = fma(u1, u1, -s);              // exact low part of the square
    double e = afivm::exp_core_f64(-s, ec);
    e = 0.5 * fma(-e, lo, e);                       // exp(-u1^2) / 2
    const double p1 = erfcx_horner(a0, a1, a2, a3, a4, f1);
    const double p2 = erfcx_horner(b0, b1, b2, b3, b4, f2);
    const double h1 = e * p1, h2 = (e * c.ratio) * p2;
    const double n1 = x1 > 0 ? 1.0 - h1 : h1;
    const double n2 = x2 > 0 ? 1.0 - h2 : h2;
    // The two products are rounded separately, as NumPy rounds them (this unit is compiled without
    // FMA contraction): deep in the money the option's time value is a 1e-10 remainder of that
    // difference, and a fused multiply-add there moves the root by up to 1e-2 in vol.
    if (VARIANT == AFIV_VARIANT_A) return c.disc * (c.F * n1 - c.K * n2);
    return c.sdq * n1 - c.kdr * n2;
}

// FAST: vols come from a root finder's bracket (finite, > 0), so the normal CDF needs no NaN
// passthrough and the quotient no IEEE special cases.
template <int VARIANT, bool FAST = false>
__device__ __forceinline__ double bs_price(const BsPoint &c, double iv, const double *__restrict__ tab)
{
    if (VARIANT == AFIV_VARIANT_A) {
        const double num = c.lnFK + 0.5 * (iv * iv) * c.tau, den = iv * c.sqrt_tau;
        const double d1 = FAST ? fdivq(num, den) : num / den;
        const double d2 = d1 - iv * c.sqrt_tau;
        return c.disc * (c.F * ndtr_dev<!FAST>(d1, tab) - c.K * ndtr_dev<!FAST>(d2, tab));
    } else {
        if (iv <= 0 || c.tau <= 0) return 1e10;
        const double num = c.logm + (c.rq + 0.5 * (iv * iv)) * c.tau, den = iv * c.sqrt_tau;
        const double d1 = FAST ? fdivq(num, den) : num / den;
        const double d2 = d1 - iv * c.sqrt_tau;
        return c.sdq * ndtr_dev<!FAST>(d1, tab) - c.kdr * ndtr_dev<!FAST>(d2, tab);
    }
}

__device__ __forceinline__ BsPoint make_point(const GridView &g, int t, int k, double spot, int variant)
{
    BsPoint c;
    c.tau = g.tau[t];
    c.sqrt_tau = sqrt(c.tau);
    c.disc = g.disc_tau[t];
    c.K = spot * g.Kmult[k];
    c.F = spot * g.fwd_mult[t];
    c.lnFK = log(c.F / c.K);
    c.sdq = spot * g.dq_tau[t];
    c.kdr = c.K * c.disc;
    c.logm = g.logm[k];
    c.rq = g.r - g.q;
    c.target = 0.0;
    c.ratio = variant == AFIV_VARIANT_A ? c.F / c.K : g.Kmult[k] * g.fwd_mult[t];
    return c;
}

// One workgroup per surface.
constexpr int ARB_THREADS = 256;
#ifndef AFIV_IV_REFILL_LANES
#define AFIV_IV_REFILL_LANES 8            // idle lanes of a wave that trigger a refill
#endif
#ifndef AFIV_IV_THREADS
#define AFIV_IV_THREADS 512               // fused kernel, wave-per-surface mode: 8 surfaces in flight per workgroup
#endif
constexpr int MAX_PTS = AFIV_MAX_STRIKES * AFIV_MAX_MATURITIES;

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int BSFORM>
__global__ __launch_bounds__(ARB_THREADS) void arb_kernel(GridView g, const double *__restrict__ iv,
                                                          const double *__restrict__ ratio, int64_t N,
                                                          double spot, int32_t *__restrict__ flags,
                                                          uint8_t *__restrict__ mask)
{
    __shared__ double price[MAX_PTS];
    __shared__ int skip_t[AFIV_MAX_MATURITIES];
    __shared__ int counters[5];      // nan, invalid, static, calendar, butterfly
    __shared__ double tab[ERFCX_LDS];
    load_erfcx_table(tab);
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    const int tid = threadIdx.x;
    for (int64_t s = blockIdx.x; s < N; s += gridDim.x) {
        if (tid < 5) counters[tid] = 0;
        if (tid < AFIV_MAX_MATURITIES) skip_t[tid] = 0;
        __syncthreads();
        const double *v = iv + s * per;
        const double *pr = ratio ? ratio + s * per : nullptr;
        int n_nan = 0, n_bad = 0;
        for (int i = tid; i < per; i += ARB_THREADS) {
            const int t = i / nK, k = i - t * nK;
            const double x = v[i];
            if (isnan(x)) { n_nan++; n_bad++; }
            else if (x <= 0) n_bad++;
            BsPoint c = make_point(g, t, k, spot, BSFORM);
            price[i] = bs_price<BSFORM>(c, x, tab);
            if (pr) {
                const double y = pr[i];
                if (isnan(y) || y < 0) skip_t[t] = 1;      // benign race: all writers store 1
            }
        }
        n_nan = wave_sum(n_nan);
        n_bad = wave_sum(n_bad);
        if ((tid & 63) == 0) {
            if (n_nan) atomicAdd(&counters[0], n_nan);
            if (n_bad) atomicAdd(&counters[1], n_bad);
        }
        __syncthreads();
        int n_stat = 0, n_cal = 0, n_fly = 0;
        const bool valid = counters[1] == 0;
        for (int i = tid; i < per; i += ARB_THREADS) {
            const int t = i / nK, k = i - t * nK;
            unsigned m = 0;
            if (valid) {
                if (k + 1 < nK && price[i] < price[i + 1] - 1e-6) { n_stat++; m |= 1u; }
                if (t + 1 < nT && price[i] > price[i + nK] + 1e-6) { n_cal++; m |= 2u; }
            }
            if (pr && k + 2 < nK && !skip_t[t]) {
                const double b = pr[i] - 2 * pr[i + 1] + pr[i + 2];
                if (b < 0) { n_fly++; m |= 4u; }
            }
            if (mask) mask[s * per + i] = (uint8_t)m;
        }
        n_stat = wave_sum(n_stat);
        n_cal = wave_sum(n_cal);
        n_fly = wave_sum(n_fly);
        if ((tid & 63) == 0) {
            if (n_stat) atomicAdd(&counters[2], n_stat);
            if (n_cal) atomicAdd(&counters[3], n_cal);
            if (n_fly) atomicAdd(&counters[4], n_fly);
        }
        __syncthreads();
        if (tid == 0) {
            const int nn = counters[0];
            const int st = valid ? counters[2] : -1;
            const int ca = valid ? counters[3] : -1;
            const int fl = counters[4];
            int status = 0;
            if (valid && st == 0 && ca == 0) status |= 1;
            if (fl == 0) status |= 2;
            if (valid) status |= 4;
            if ((double)nn / (double)per < 0.1) status |= 8;
            if (nn == 0) status |= 16;
            flags[4 * s + 0] = st;
            flags[4 * s + 1] = ca;
            flags[4 * s + 2] = fl;
            flags[4 * s + 3] = status;
        }
        __syncthreads();
    }
}

// Fused implied vol + checks.
//
// A group of GT threads owns one surface at a time: one wave (GT = 64, no workgroup barriers in
// the main loop) when there are surfaces enough to fill the chip, the whole workgroup otherwise
// (small batches, and lattices whose LDS footprint allows one surface per workgroup only).
//
// Brent's iteration count varies from 5 to 37 over a surface (7 at the money, 15 deep in the
// money on average), so lanes do not march through a fixed set of grid points.  The root finder is
// a per-lane state machine that makes ONE objective evaluation per trip of the wave's loop; a lane
// whose point has converged takes the next point from the surface's queue (strike-major order,
// deep in-the-money strikes first) inside the same trip, between the two halves of Brent's loop
// body, so no lane ever idles through an evaluation while the queue has points.  The objective
// values at the two bracket ends depend on the grid point and the spot only (minus the target):
// a small kernel tabulates them once per call, which saves two of the ~12.7 evaluations per point.
// A wave therefore spends sum(iterations)/64 trips on a surface plus a short tail, 1.7 x fewer
// evaluations than lanes bound to fixed points on the benchmark lattice.
//
// The surface's price ratios are staged in LDS (coalesced read); the butterfly check runs on them
// at once, and each entry is then overwritten by the Black-Scholes price of the solved vol,
// taken from Brent's last objective value (objective + target, exact at a root), so the static
// and calendar checks cost no further normal CDFs.  Vols go straight to global memory.
struct MatRow { double tau, sqrt_tau, disc, F, sdq; };
struct __attribute__((aligned(32))) PointRow { double lnfk, fa, fb, ratio; };

constexpr double BRENT_XA = 0.001, BRENT_XB = 3.0, BRENT_XTOL = 1e-6, BRENT_RTOL = 4 * 2.220446049250313e-16;
constexpr int BRENT_MAXITER = 100;

// Black-Scholes values at the bracket ends and log(F/K) for every grid point (one thread each).
template <int VARIANT>
__global__ __launch_bounds__(256) void point_table_kernel(GridView g, double spot)
{
    __shared__ double tab[ERFCX_LDS];
    load_erfcx_table(tab);
    __syncthreads();
    const int per = g.nT * g.nK;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= per) return;
    const int t = i / g.nK, k = i - t * g.nK;
    const BsPoint c = make_point(g, t, k, spot, VARIANT);
    PointRow row;
    row.lnfk = c.lnFK;
    const afivm::ExpCoef ec = afivm::load_exp_coef();
    row.fa = bs_objective<VARIANT>(c, BRENT_XA, tab, ec);
    row.fb = bs_objective<VARIANT>(c, BRENT_XB, tab, ec);
    row.ratio = c.ratio;
    reinterpret_cast<PointRow *>(g.point)[i] = row;
}

template <int GT>
__device__ __forceinline__ void group_sync()
{
    if (GT == 64) {
        // one wave: its LDS operations execute in order; only the compiler must not reorder
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// j / n for 0 <= j < 4096, 1 <= n <= 64 with inv = 1.0f / n (exact: (j + 1/2)/n is at least
// 1/(2n) away from an integer, far more than the float rounding)
__device__ __forceinline__ int small_div(int j, float inv) { return (int)(((float)j + 0.5f) * inv); }

__device__ __forceinline__ bool opposite_sign(double a, double b)
{
    return ((__double2hiint(a) ^ __double2hiint(b)) < 0);
}

enum { PH_IDLE = 0, PH_ITER = 1, PH_DONE = 2 };

template <int VARIANT, bool CHECKS, int GT, int THREADS>
__global__ __launch_bounds__(THREADS) void surface_kernel(GridView g, const double *__restrict__ ratio,
                                                          int64_t N, double spot, double *__restrict__ iv_out,
                                                          int32_t *__restrict__ flags)
{
    constexpr int NG = THREADS / GT;
    extern __shared__ double dyn[];
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    double *tab = dyn;                                   // ERFCX_LDS
    MatRow *mat = reinterpret_cast<MatRow *>(tab + ERFCX_LDS);                // [64]
    double *strike = reinterpret_cast<double *>(mat + AFIV_MAX_MATURITIES);   // [64]
    double *logm = strike + AFIV_MAX_STRIKES;            // [64]
    double *price_all = logm + AFIV_MAX_STRIKES;         // [NG][per]
    unsigned short *order = reinterpret_cast<unsigned short *>(price_all + (size_t)NG * per);   // [per] queue position -> lattice point
    __shared__ int skip_all[NG][AFIV_MAX_MATURITIES];
    __shared__ int ctr_all[NG][8];                       // queue head, nan, bad, static, calendar, butterfly
    const int tid = threadIdx.x, lane = tid & 63, gid = tid / GT, lt = tid % GT;
    double *price = price_all + (size_t)gid * per;
    int *skip_t = skip_all[gid];
    int *ctr = ctr_all[gid];
    const PointRow *__restrict__ point = reinterpret_cast<const PointRow *>(g.point);

    load_erfcx_table(tab);
    for (int t = tid; t < nT; t += THREADS) {
        MatRow m;
        m.tau = g.tau[t];
        m.sqrt_tau = sqrt(m.tau);
        m.disc = g.disc_tau[t];
        m.F = spot * g.fwd_mult[t];
        m.sdq = spot * g.dq_tau[t];
        mat[t] = m;
    }
    for (int k = tid; k < nK; k += THREADS) {
        strike[k] = spot * g.Kmult[k];
        logm[k] = g.logm[k];
    }
    for (int i = tid; i < per; i += THREADS) order[i] = g.order[i];
    __syncthreads();

    const float inv_nK = 1.0f / (float)nK;
    const double nan = __builtin_nan("");
    const double rq = g.r - g.q;
    const afivm::ExpCoef ec = afivm::load_exp_coef();
    const double xa = BRENT_XA, xb = BRENT_XB, xtol = BRENT_XTOL, rtol = BRENT_RTOL;

    for (int64_t s = (int64_t)blockIdx.x * NG + gid; s < N; s += (int64_t)gridDim.x * NG) {
        group_sync<GT>();                                // the previous surface's counters have been read
        if (lt < 8) ctr[lt] = 0;
        if (lt < AFIV_MAX_MATURITIES) skip_t[lt] = 0;
        group_sync<GT>();
        const double *src = ratio + s * per;
        for (int i = lt; i < per; i += GT) {
            const double y = src[i];
            price[i] = y;
            if (CHECKS && (isnan(y) || y < 0)) skip_t[small_div(i, inv_nK)] = 1;   // benign race: all writers store 1
        }
        group_sync<GT>();
        int n_fly = 0;
        if (CHECKS) {
            for (int i = lt; i < per; i += GT) {
                const int t = small_div(i, inv_nK), k = i - __mul24(t, nK);
                if (k + 2 < nK && !skip_t[t]) {
                    const double b = price[i] - 2 * price[i + 1] + price[i + 2];
                    if (b < 0) n_fly++;
                }
            }
            group_sync<GT>();                            // ratios are overwritten from here on
        }

        // ---- Brent: per trip the second half of brentq.c's loop body (choose the step), one
        //      objective evaluation, the first half (re-bracket, convergence test), and the
        //      hand-over of converged lanes to new points ----------------------------------------
        int n_nan = 0, n_bad = 0;
        int next = 0;                                    // queue head when the group is one wave
        int phase = PH_IDLE, iter = 0, i_cur = 0;
        bool exhausted = false;
        BsPoint c = {};
        double res = nan, fr = 0.0;
        double xpre = xa, xcur = xb, xblk = xa, fpre = 1, fcur = -1, fblk = 1, spre = 0, scur = 0;
        double delta = 1, sbis = 1;
        for (;;) {
            // -- finished lanes store; idle lanes take the next point of the queue
            if (phase == PH_DONE) {
                if (VARIANT != AFIV_VARIANT_A && (isnan(res) || res <= 0 || res > 3.0)) res = nan;
                iv_out[s * per + i_cur] = res;
                if (CHECKS) {
                    price[i_cur] = fr + c.target;
                    if (isnan(res)) { n_nan++; n_bad++; }
                    else if (res <= 0) n_bad++;
                }
                phase = PH_IDLE;
            }
            // Handing out points costs about a third of a trip's instructions whether one lane or
            // sixty take one, and an idle lane wastes 1/64 of a trip: refill once enough lanes
            // wait (or nothing else is in flight).
            const bool want = phase == PH_IDLE && !exhausted;
            const unsigned long long wm = __ballot(want);
            if (wm && (__popcll(wm) >= AFIV_IV_REFILL_LANES || __ballot(phase != PH_IDLE) == 0)) {
                const int cnt = __popcll(wm);
                int base;
                if (GT == 64) {
                    base = next;
                    next += cnt;
                } else {
                    const int leader = __ffsll((long long)wm) - 1;
                    int b = 0;
                    if (lane == leader) b = atomicAdd(&ctr[0], cnt);
                    base = __shfl(b, leader, 64);
                }
                const int j = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32),
                                                                   __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u));
                if (want) {
                    if (j >= per) {
                        exhausted = true;
                    } else {
                        i_cur = order[j];
                        const int t = small_div(i_cur, inv_nK), k = i_cur - __mul24(t, nK);
                        const PointRow pt = point[i_cur];
                        const MatRow m = mat[t];
                        c.tau = m.tau;
                        c.sqrt_tau = m.sqrt_tau;
                        c.disc = m.disc;
                        c.F = m.F;
                        c.sdq = m.sdq;
                        c.K = strike[k];
                        c.kdr = c.K * c.disc;
                        c.logm = logm[k];
                        c.rq = rq;
                        c.lnFK = pt.lnfk;
                        c.ratio = pt.ratio;
                        const double y = price[i_cur];
                        const double hp = y * spot * c.disc;
                        c.target = hp;
                        bool ok;
                        if (VARIANT == AFIV_VARIANT_A) {
                            const double intrinsic = fmax(c.F - c.K, 0.0) * c.disc;
                            ok = !(hp < intrinsic * 0.99);
                        } else {
                            ok = !(hp < 1e-10 || isnan(y));
                        }
                        // brentq.c up to its loop, then the first half of the loop body for i = 0
                        const double fa = pt.fa - hp, fb = pt.fb - hp;
                        res = nan;
                        fr = 0.0;
                        phase = PH_DONE;                 // rejected, or decided without iterating
                        if (ok && !isnan(fa) && !isnan(fb)) {
                            if (fa == 0) res = xa;
                            else if (fb == 0) res = xb;
                            else if (opposite_sign(fa, fb)) {
                                // the bracket is valid: xblk = xa, fblk = fa, spre = scur = xb - xa,
                                // swapped so that |fcur| <= |fblk|
                                const bool sw = fabs(fa) < fabs(fb);
                                xpre = sw ? xb : xa;  fpre = sw ? fb : fa;
                                xcur = sw ? xa : xb;  fcur = sw ? fa : fb;
                                xblk = xpre;          fblk = fpre;
                                spre = scur = xb - xa;
                                delta = __builtin_fma(0.5 * rtol, fabs(xcur), 0.5 * xtol);   // (xtol + rtol |xcur|)/2 in one operation
                                sbis = (xblk - xcur) / 2;
                                iter = 0;
                                phase = PH_ITER;         // |sbis| = 1.4995 > delta: never done here
                            }
                        }
                    }
                }
            }
            if (__ballot(phase != PH_IDLE) == 0) break;  // nothing in flight and the queue is empty

            // -- second half of the loop body: interpolate or bisect
            {
                const bool interp = fabs(spre) > delta && fabs(fcur) < fabs(fpre);
                const bool sec = xpre == xblk;
                const double a = fpre - fcur, b = xpre - xcur;
                // secant:  -fcur (xcur - xpre) / (fcur - fpre) = (-fcur b) / a.
                // inverse quadratic, brentq.c's -fcur (fblk dblk - fpre dpre) / (dblk dpre (fblk - fpre))
                // with dpre = a / b, dblk = A / B multiplied through by b B: one quotient instead
                // of three, -fcur (fblk A b - fpre a B) / (A a (fblk - fpre)); the same step to a
                // few ulp (the iteration paths stay those of the oracle, see the tests).
                const double A = fblk - fcur, B = xblk - xcur;
                const double num_q = -fcur * (fblk * A * b - fpre * a * B);
                const double den_q = A * a * (fblk - fpre);
                const double stry = fdivq(sec ? -fcur * b : num_q, sec ? a : den_q);
                const double lim = fmin(fabs(spre), __builtin_fma(3.0, fabs(sbis), -delta));
                const bool take = interp && 2 * fabs(stry) < lim;
                spre = take ? scur : sbis;
                scur = take ? stry : sbis;
                xpre = xcur;
                fpre = fcur;
                const double sd = sbis > 0 ? delta : -delta;
                xcur += fabs(scur) > delta ? scur : sd;
            }

            double f = bs_objective<VARIANT>(c, xcur, tab, ec) - c.target;
            // pin the evaluation here: its only reader is the conditional block below, and sunk
            // into it the table reads would no longer be issued ahead of the exponential
            asm volatile("" : "+v"(f));

            // -- first half of the next pass: re-bracket, order, convergence test
            if (phase == PH_ITER) {
                fcur = f;
                iter++;
                const bool sgn = fpre != 0 && fcur != 0 && opposite_sign(fpre, fcur);
                const double dx = xcur - xpre;
                xblk = sgn ? xpre : xblk;
                fblk = sgn ? fpre : fblk;
                spre = sgn ? dx : spre;
                scur = sgn ? dx : scur;
                const bool sw = fabs(fblk) < fabs(fcur);
                const double xc = xcur, fc = fcur;
                xpre = sw ? xc : xpre;
                xcur = sw ? xblk : xc;
                xblk = sw ? xc : xblk;
                fpre = sw ? fc : fpre;
                fcur = sw ? fblk : fc;
                fblk = sw ? fc : fblk;
                delta = __builtin_fma(0.5 * rtol, fabs(xcur), 0.5 * xtol);   // (xtol + rtol |xcur|)/2 in one operation
                sbis = (xblk - xcur) / 2;
                const bool failed = isnan(f) || iter == BRENT_MAXITER;
                const bool done = fcur == 0 || fabs(sbis) < delta;
                if (failed || done) {
                    res = failed ? nan : xcur;
                    fr = failed ? 0.0 : fcur;
                    phase = PH_DONE;
                }
            }
        }

        if (CHECKS) {
            n_nan = wave_sum(n_nan);
            n_bad = wave_sum(n_bad);
            n_fly = wave_sum(n_fly);
            if (lane == 0) {
                if (n_nan) atomicAdd(&ctr[1], n_nan);
                if (n_bad) atomicAdd(&ctr[2], n_bad);
                if (n_fly) atomicAdd(&ctr[5], n_fly);
            }
            group_sync<GT>();
            const bool valid = ctr[2] == 0;
            int n_stat = 0, n_cal = 0;
            if (valid) {
                for (int i = lt; i < per; i += GT) {
                    const int t = small_div(i, inv_nK), k = i - __mul24(t, nK);
                    if (k + 1 < nK && price[i] < price[i + 1] - 1e-6) n_stat++;
                    if (t + 1 < nT && price[i] > price[i + nK] + 1e-6) n_cal++;
                }
            }
            n_stat = wave_sum(n_stat);
            n_cal = wave_sum(n_cal);
            if (lane == 0) {
                if (n_stat) atomicAdd(&ctr[3], n_stat);
                if (n_cal) atomicAdd(&ctr[4], n_cal);
            }
            group_sync<GT>();
            if (lt == 0) {
                const int nn = ctr[1];
                const int st = valid ? ctr[3] : -1;
                const int ca = valid ? ctr[4] : -1;
                const int fl = ctr[5];
                int status = 0;
                if (valid && st == 0 && ca == 0) status |= 1;
                if (fl == 0) status |= 2;
                if (valid) status |= 4;
                if ((double)nn / (double)per < 0.1) status |= 8;
                if (nn == 0) status |= 16;
                flags[4 * s + 0] = st;
                flags[4 * s + 1] = ca;
                flags[4 * s + 2] = fl;
                flags[4 * s + 3] = status;
            }
        }
    }
}

// cvae_model.py:272-285 over a batch: sum and count of the collected penalties.
// Single workgroup, fixed summation order (deterministic).
__global__ __launch_bounds__(256) void penalty_kernel(GridView g, const double *__restrict__ ratio,
                                                      int64_t B, double *__restrict__ acc_out)
{
    __shared__ double ssum[256];
    __shared__ double scnt[256];
    const int nT = g.nT, nK = g.nK;
    double sum = 0.0, cnt = 0.0;
    for (int64_t item = threadIdx.x; item < B * nT; item += 256) {
        const double *pr = ratio + item * nK;
        bool skip = false;
        for (int k = 0; k < nK; ++k)
            if (isnan(pr[k]) || pr[k] < 0) skip = true;
        if (skip) continue;
        for (int j = 0; j + 1 < nK; ++j)
            if (pr[j + 1] > pr[j]) { sum += (pr[j + 1] - pr[j]) * 100.0; cnt += 1.0; }
        for (int j = 0; j + 2 < nK; ++j) {
            const double b = pr[j] - 2 * pr[j + 1] + pr[j + 2];
            if (b < 0) { sum += -b * 100.0; cnt += 1.0; }
        }
    }
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            ssum[threadIdx.x] += ssum[threadIdx.x + off];
            scnt[threadIdx.x] += scnt[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { acc_out[0] = ssum[0]; acc_out[1] = scnt[0]; }
}

// run_single_heston_calibration.py:252-287 for (candidate c, day d).  One workgroup per
// candidate keeps that candidate's model ratios in LDS; threads walk the days.
__global__ __launch_bounds__(256) void stage1_kernel(GridView g, const double *__restrict__ model_ratio,
                                                     const double *__restrict__ params, int64_t C,
                                                     const double *__restrict__ market_ratio, int64_t D,
                                                     double drift_weight, double *__restrict__ loss_out)
{
    __shared__ double model[MAX_PTS];
    __shared__ int atm[MAX_PTS / AFIV_MAX_MATURITIES];
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += 256) model[i] = model_ratio[c * per + i];
        for (int k = threadIdx.x; k < nK; k += 256) atm[k] = fabs(g.logm[k]) < 0.05 ? 1 : 0;
        __syncthreads();
        const double kappa = params[5 * c + 0], theta = params[5 * c + 1], sigma = params[5 * c + 2];
        const double feller = 5.0 * fmax(0.0, sigma * sigma - 2 * kappa * theta);
        for (int64_t d = threadIdx.x; d < D; d += 256) {
            const double *mk = market_ratio + d * per;
            double se = 0.0, sm_model = 0.0, sm_mkt = 0.0;
            int n = 0, n_atm = 0;
            bool bad = false;
            for (int i = 0; i < per; ++i) {
                const double y = mk[i];
                if (isnan(y)) continue;
                const double m = model[i];
                if (isnan(m) || m < 0) bad = true;
                const double e = m - y;
                se += e * e;
                n++;
                if (atm[i % nK]) { sm_model += m; sm_mkt += y; n_atm++; }
            }
            double loss;
            if (n < 10) loss = __builtin_nan("");          // the reference skips such a day (:202-203)
            else if (bad) loss = 1e10;
            else {
                const double rmse = sqrt(se / n);
                const double drift = n_atm > 0 ? fabs(sm_model / n_atm - sm_mkt / n_atm) : 0.0;
                loss = rmse + drift_weight * drift + feller;
            }
            loss_out[c * D + d] = loss;
        }
    }
}

// Unit-test hook: the normal CDF used by the kernels above, element-wise.
__global__ void debug_ndtr_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ out)
{
    __shared__ double tab[ERFCX_LDS];
    load_erfcx_table(tab);
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = ndtr_dev(x[i], tab);
}

}  // namespace

int afiv_launch_debug_ndtr(const double *x, int64_t n, double *out, hipStream_t s)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(debug_ndtr_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, n, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

namespace {
size_t surface_lds_bytes(const GridView &g, int groups)
{
    return sizeof(double) * ((size_t)ERFCX_LDS + 5 * AFIV_MAX_MATURITIES + 2 * AFIV_MAX_STRIKES +
                             (size_t)groups * g.nT * g.nK) +
           ((size_t)g.nT * g.nK * sizeof(unsigned short) + 7) / 8 * 8;
}

template <int VARIANT, bool CHECKS, int GT, int THREADS>
int launch_surface_kernel(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                          int32_t *flags, hipStream_t s)
{
    constexpr int NG = THREADS / GT;
    const size_t lds = surface_lds_bytes(g, NG);
    if (lds > 48 * 1024) {
        // raising the dynamic-LDS limit is a per-function attribute, not an allocation
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&surface_kernel<VARIANT, CHECKS, GT, THREADS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
    }
    const int per = g.nT * g.nK;
    hipLaunchKernelGGL(point_table_kernel<VARIANT>, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, g, spot);
    int64_t blocks = (N + NG - 1) / NG;
    if (blocks > 256 * 24) blocks = 256 * 24;
    hipLaunchKernelGGL((surface_kernel<VARIANT, CHECKS, GT, THREADS>), dim3((unsigned)blocks), dim3(THREADS), lds,
                       s, g, ratio, N, spot, iv_out, flags);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

template <int VARIANT, bool CHECKS>
int launch_surface_grouped(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                           int32_t *flags, hipStream_t s)
{
    // a wave per surface needs surfaces enough to fill the chip and room for two workgroups per CU
    if (N >= 2048 && surface_lds_bytes(g, AFIV_IV_THREADS / 64) <= 76 * 1024)
        return launch_surface_kernel<VARIANT, CHECKS, 64, AFIV_IV_THREADS>(g, ratio, N, spot, iv_out, flags, s);
    return launch_surface_kernel<VARIANT, CHECKS, 256, 256>(g, ratio, N, spot, iv_out, flags, s);
}
}  // namespace

// implied vols and (flags != nullptr) the checks in one pass; bs form of the checks = variant
int afiv_launch_surface_iv(const GridView &g, const double *ratio, int64_t N, double spot, int variant,
                           double *iv_out, int32_t *flags, hipStream_t s)
{
    if (N <= 0) return 0;
    if (variant == 0)
        return flags ? launch_surface_grouped<0, true>(g, ratio, N, spot, iv_out, flags, s)
                     : launch_surface_grouped<0, false>(g, ratio, N, spot, iv_out, flags, s);
    return flags ? launch_surface_grouped<1, true>(g, ratio, N, spot, iv_out, flags, s)
                 : launch_surface_grouped<1, false>(g, ratio, N, spot, iv_out, flags, s);
}

int afiv_launch_implied_vol(const GridView &g, const double *ratio, int64_t N, double spot, int variant,
                            double *iv_out, hipStream_t s)
{
    return afiv_launch_surface_iv(g, ratio, N, spot, variant, iv_out, nullptr, s);
}

int afiv_launch_arb_check(const GridView &g, const double *iv, const double *ratio, int64_t N, double spot,
                          int bs_form, int32_t *flags_out, uint8_t *mask_out, hipStream_t s)
{
    if (N <= 0) return 0;
    int64_t blocks = N < 256 * 32 ? N : 256 * 32;
    if (bs_form == 0)
        hipLaunchKernelGGL(arb_kernel<0>, dim3((unsigned)blocks), dim3(ARB_THREADS), 0, s, g, iv, ratio, N, spot, flags_out, mask_out);
    else
        hipLaunchKernelGGL(arb_kernel<1>, dim3((unsigned)blocks), dim3(ARB_THREADS), 0, s, g, iv, ratio, N, spot, flags_out, mask_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int afiv_launch_penalty(const GridView &g, const double *ratio, int64_t B, double *acc_out, hipStream_t s)
{
    hipLaunchKernelGGL(penalty_kernel, dim3(1), dim3(256), 0, s, g, ratio, B, acc_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int afiv_launch_stage1(const GridView &g, const double *model_ratio, const double *params, int64_t C,
                       const double *market_ratio, int64_t D, double drift_weight, double *loss_out,
                       hipStream_t s)
{
    if (C <= 0 || D <= 0) return 0;
    int64_t blocks = C < 256 * 16 ? C : 256 * 16;
    hipLaunchKernelGGL(stage1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, model_ratio, params, C,
                       market_ratio, D, drift_weight, loss_out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
