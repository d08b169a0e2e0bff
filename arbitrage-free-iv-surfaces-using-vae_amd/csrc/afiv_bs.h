// Black-Scholes building blocks shared by the implied-vol and check kernels (device code only).
//
// Every translation unit that includes this file is compiled with -ffp-contract=off: the kernels
// reproduce NumPy/SciPy evaluation order so that guards, bracket signs and violation counts agree
// with the Python reference bit for bit on identical inputs (mul and add are never fused unless
// written as an fma here).
#pragma once
#include "afiv_internal.h"
#include "afiv_math.h"
#include "erfcx_table.inc"

namespace afivbs {

// Normal CDF through one scaled complementary error function: N(a) = erfc(-a/sqrt 2)/2 with
// erfc(u) = exp(-u^2) erfcx(u), erfcx from a 64-segment degree-8 table in y = 1/(1+u) (see
// tools/gen_erfcx_table.py; 1.5 ulp).  One code path for every lane -- the C library's erf/erfc
// pick one of five ranges per argument, which a wave executes back to back when its lanes
// disagree.  The table lives in LDS (ERFCX_LDS doubles, filled by load_erfcx_table).
inline __constant__ double c_erfcx[AFIV_ERFCX_NSEG * AFIV_ERFCX_NCOEF] = {AFIV_ERFCX_TABLE};
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
// `aux`, when given, receives what a derivative-based root finder needs from the same evaluation:
// d1, d2 and exp(-d1^2/2)/2 (the vega is a per-point constant times the latter).
struct BsAux { double d1, d2, e; };

template <int VARIANT>
__device__ __forceinline__ double bs_objective(const BsPoint &c, double iv, const double *__restrict__ tab,
                                               const afivm::ExpCoef &ec, BsAux *aux = nullptr)
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
    if (aux) { aux->d1 = d1; aux->d2 = d2; aux->e = e; }
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

__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

constexpr double BRENT_XA = 0.001, BRENT_XB = 3.0, BRENT_XTOL = 1e-6, BRENT_RTOL = 4 * 2.220446049250313e-16;
constexpr int BRENT_MAXITER = 100;

// j / n for 0 <= j < 4096, 1 <= n <= 64 with inv = 1.0f / n (exact: (j + 1/2)/n is at least
// 1/(2n) away from an integer, far more than the float rounding)
__device__ __forceinline__ int small_div(int j, float inv) { return (int)(((float)j + 0.5f) * inv); }

__device__ __forceinline__ bool opposite_sign(double a, double b)
{
    return ((__double2hiint(a) ^ __double2hiint(b)) < 0);
}

}  // namespace afivbs
