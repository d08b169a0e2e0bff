// FP64 elementary functions written for these kernels (gfx950, wave64).
//
// Why not the device math library: its double-precision exp / log / sincos / atan2 cost 40 / 95 /
// 155 / 100 vector instructions each and carry a Payne-Hanek reduction and special-value handling
// that these kernels never need but whose registers are allocated all the same.  The functions
// below have ONE code path (no data-dependent branches, so a wave never executes two variants back
// to back), assume finite arguments of moderate size, and cost 20-30 instructions: sincos, log1p
// and atan2 reduce their argument against a small table that the kernel keeps in LDS (1, 2 and
// 4 KB; load_*_table), which leaves a polynomial of degree 3-7 on an interval of 2^-8..2^-5
// instead of degree 11-23 on the classical one.
//
// Accuracy (tests/test_gpu_math.py checks each against 50-digit references):
//   exp_f64         relative 2 ulp for x in [-745, 709]; 0 below, +inf above
//   sincos_tab_f64  ABSOLUTE error <= 3e-16 for |x| < 1e9 (two-term Cody-Waite; sin and cos only
//                   ever multiply an amplitude here, so absolute error is what matters)
//   log1p_tab_f64   relative 2 ulp for f > -1 (relative accuracy near f = 0 is the point: the
//                   characteristic function divides this logarithm by sigma_v^2)
//   atan2_tab_f64   relative 2 ulp, all quadrants, finite arguments not both zero
//   rcp_f64         relative 1 ulp for normal x (v_rcp_f64 + two Newton steps)
#pragma once
#include <hip/hip_runtime.h>

#include "afiv_math_coeffs.inc"

namespace afivm {

// Polynomial coefficients live in constant memory and are fetched where they are used, with scalar
// loads, so that the FMAs read them as scalar operands.  f64 instructions take no 64-bit literal:
// written as literals each coefficient would be copied into a vector register pair, and hoisted
// out of the kernels' loops those pairs (or, hoisted as scalars, twice as many scalar registers) end up
// spilled.  coeffs() hands out the table's address through an empty asm statement that names a
// value the caller has just computed: the compiler cannot prove the address loop-invariant, nor
// move the loads above that value, so each polynomial's coefficients are live only around it.
enum { OFF_EXP = 0, OFF_SINT = 16, OFF_COST = 19, OFF_LOGT = 22, N_COEF = 32 };
inline __constant__ __attribute__((aligned(64))) double c_tab[N_COEF] = {
    AFIV_EXP_C0, AFIV_EXP_C1, AFIV_EXP_C2, AFIV_EXP_C3, AFIV_EXP_C4, AFIV_EXP_C5, AFIV_EXP_C6, AFIV_EXP_C7,
    AFIV_EXP_C8, AFIV_EXP_C9, AFIV_EXP_C10, AFIV_EXP_C11, AFIV_EXP_C12, AFIV_EXP_C13, 0.0, 0.0,
    AFIV_SINT_C0, AFIV_SINT_C1, AFIV_SINT_C2, AFIV_COST_C0, AFIV_COST_C1, AFIV_COST_C2,
    AFIV_LOGT_C0, AFIV_LOGT_C1, AFIV_LOGT_C2, AFIV_LOGT_C3, AFIV_LOGT_C4, AFIV_LOGT_C5, 0.0, 0.0, 0.0, 0.0};

typedef const double __attribute__((address_space(4))) *coef_ptr;

__device__ __forceinline__ coef_ptr coeffs(double after)
{
    unsigned long long a = (unsigned long long)(const void *)c_tab;
    asm volatile("" : "+s"(a) : "v"(after));
    return (coef_ptr)a;
}

// The same table without the pinning: for kernels with scalar registers to spare, where the
// compiler may load the coefficients once, outside the loops.
__device__ __forceinline__ coef_ptr coeffs_hoistable()
{
    return (coef_ptr)(unsigned long long)(const void *)c_tab;
}


// p z + c with the coefficient read straight from its scalar register pair.  (Left to itself the
// compiler copies a coefficient that arrives as part of a wide scalar load into a vector register
// pair first: two extra instructions per Horner step.)
__device__ __forceinline__ double fma_sc(double p, double z, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(z), "s"(c));
    return d;
}

__device__ __forceinline__ double rcp_f64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}

// exp(x) without clamping or special-value selects: exact limits for every finite x (the
// float-to-int conversion saturates and v_ldexp_f64 then under/overflows to 0 / +inf), NaN for a
// NaN argument (it propagates through the reduction), but NaN -- not 0 or +inf -- for x = -+inf.
template <bool PINNED = true>
__device__ __forceinline__ double exp_core_f64(double x)
{
    const double k = __builtin_rint(x * AFIV_LOG2E);
    double r = __builtin_fma(k, -AFIV_LN2_HI, x);
    r = __builtin_fma(k, -AFIV_LN2_LO, r);
    const coef_ptr c_exp = (PINNED ? coeffs(r) : coeffs_hoistable()) + OFF_EXP;
    double p = c_exp[13];
    p = fma_sc(p, r, c_exp[12]);
    p = fma_sc(p, r, c_exp[11]);
    p = fma_sc(p, r, c_exp[10]);
    p = fma_sc(p, r, c_exp[9]);
    p = fma_sc(p, r, c_exp[8]);
    p = fma_sc(p, r, c_exp[7]);
    p = fma_sc(p, r, c_exp[6]);
    p = fma_sc(p, r, c_exp[5]);
    p = fma_sc(p, r, c_exp[4]);
    p = fma_sc(p, r, c_exp[3]);
    p = fma_sc(p, r, c_exp[2]);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);                          // v_ldexp_f64: exact scaling, graceful underflow
}

// exp(x) and exp(x) - 1 together: for |x| < log(2)/2 the reduction leaves k = 0 and the polynomial
// itself is 1 + r q, so r q is exp(x) - 1 without the cancellation of subtracting 1 afterwards
// (relative 2 ulp down to x -> 0); elsewhere the subtraction loses nothing.
template <bool PINNED = true>
__device__ __forceinline__ double exp_expm1_f64(double x, double *em1)
{
    const double k = __builtin_rint(x * AFIV_LOG2E);
    double r = __builtin_fma(k, -AFIV_LN2_HI, x);
    r = __builtin_fma(k, -AFIV_LN2_LO, r);
    const coef_ptr c_exp = (PINNED ? coeffs(r) : coeffs_hoistable()) + OFF_EXP;
    double p = c_exp[13];
    p = fma_sc(p, r, c_exp[12]);
    p = fma_sc(p, r, c_exp[11]);
    p = fma_sc(p, r, c_exp[10]);
    p = fma_sc(p, r, c_exp[9]);
    p = fma_sc(p, r, c_exp[8]);
    p = fma_sc(p, r, c_exp[7]);
    p = fma_sc(p, r, c_exp[6]);
    p = fma_sc(p, r, c_exp[5]);
    p = fma_sc(p, r, c_exp[4]);
    p = fma_sc(p, r, c_exp[3]);
    p = fma_sc(p, r, c_exp[2]);
    const double q = __builtin_fma(p, r, 1.0);
    const double e = __builtin_ldexp(__builtin_fma(q, r, 1.0), (int)k);
    *em1 = k == 0.0 ? q * r : e - 1.0;
    return e;
}

// The same with the coefficients in caller-held registers: a kernel with scalar registers to
// spare loads them once (load_exp_coef) instead of fetching them at every call.
struct ExpCoef { double c[12]; };   // x^2 ... x^13

__device__ __forceinline__ ExpCoef load_exp_coef()
{
    const coef_ptr c_exp = coeffs_hoistable() + OFF_EXP;
    ExpCoef k;
#pragma unroll
    for (int i = 0; i < 12; ++i) k.c[i] = c_exp[i + 2];
    return k;
}

__device__ __forceinline__ double exp_core_f64(double x, const ExpCoef &k)
{
    const double n = __builtin_rint(x * AFIV_LOG2E);
    double r = __builtin_fma(n, -AFIV_LN2_HI, x);
    r = __builtin_fma(n, -AFIV_LN2_LO, r);
    double p = k.c[11];
#pragma unroll
    for (int i = 10; i >= 0; --i) p = fma_sc(p, r, k.c[i]);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// exp(x) through a 128-entry table in LDS: x = (128 m + j) ln(2)/128 + r with |r| <= ln(2)/256,
// exp x = 2^m T_j (1 + w), w = exp r - 1 by the Taylor polynomial to r^5 -- 16 vector instructions
// and one LDS read against 22 for the degree-13 form, and a dependent chain a third as long.  The
// table holds 2^(j/128) as (hi, lo); the plain exponential uses hi alone (relative error 1 ulp).
// Same limits as exp_core_f64: graceful underflow to 0, +inf above the range, NaN for a NaN.
// `tab` is EXP_TAB_DOUBLES doubles filled by load_exp_table.
constexpr int EXP_TAB_DOUBLES = 256;
inline __constant__ __attribute__((aligned(16))) double c_exp_tab[EXP_TAB_DOUBLES] = {AFIV_EXP_TAB_LIST};

__device__ __forceinline__ void load_exp_table(double *lds_tab)
{
    for (int i = threadIdx.x; i < EXP_TAB_DOUBLES; i += blockDim.x) lds_tab[i] = c_exp_tab[i];
}

// w = exp(r) - 1 for the reduced argument, and the table index / binary exponent of x
__device__ __forceinline__ double exp_tab_reduce(double x, int *j, int *m)
{
    // k = rint(128 x / ln 2) by the shift trick: the sum's low mantissa bits hold k in two's complement
    const double shifted = __builtin_fma(x, AFIV_128_OVER_LN2, 0x1.8p52);
    const double k = shifted - 0x1.8p52;
    double r = __builtin_fma(k, -AFIV_LN2O128_HI, x);
    r = __builtin_fma(k, -AFIV_LN2O128_LO, r);
    const int ki = __double2loint(shifted);
    *j = ki & 127;
    *m = ki >> 7;
    const coef_ptr c_exp = coeffs(r) + OFF_EXP;                 // 1/2, 1/6, 1/24, 1/120: the Taylor coefficients
    double p = c_exp[5];
    p = fma_sc(p, r, c_exp[4]);
    p = fma_sc(p, r, c_exp[3]);
    p = fma_sc(p, r, c_exp[2]);
    return __builtin_fma(p, r, 1.0) * r;
}

__device__ __forceinline__ double exp_tab_f64(double x, const double *__restrict__ tab)
{
    int j, m;
    const double w = exp_tab_reduce(x, &j, &m);
    const double th = tab[2 * j];
    return __builtin_ldexp(__builtin_fma(th, w, th), m);
}

// exp(x) and exp(x) - 1 together, the latter without cancellation: with s = 2^m,
// exp x - 1 = (s T_hi - 1) + s (T_hi w + T_lo), where s T_hi - 1 is exact whenever the two are
// within a factor 2 (and elsewhere nothing cancels), and for m = j = 0 it is w itself.
// Relative error 2 ulp for every x, x -> 0 included.
__device__ __forceinline__ double exp_expm1_tab_f64(double x, const double *__restrict__ tab, double *em1)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    int j, m;
    const double w = exp_tab_reduce(x, &j, &m);
    const d2 t = reinterpret_cast<const d2 *>(tab)[j];
    const double v = __builtin_fma(t.x, w, t.y);
    const double s = __builtin_ldexp(1.0, m);
    const double ts = t.x * s, vs = v * s;
    *em1 = (ts - 1.0) + vs;
    return ts + vs;
}

template <bool PINNED = true>
__device__ __forceinline__ double exp_f64(double x)
{
    // the clamp keeps the exponent inside int range; NaN in, NaN out
    const double y = exp_core_f64<PINNED>(fmin(fmax(x, -1000.0), 1000.0));
    return x != x ? x : y;
}

// sin and cos of the same argument through a 64-entry table in LDS: x = j pi/32 + r with
// |r| <= pi/64, (S, C) = (sin, cos)(j pi/32) from the table, and
//   sin x = S + (S (cos r - 1) + C sin r),   cos x = C + (C (cos r - 1) - S sin r)
// with two short Taylor polynomials: no quadrant logic and half the polynomial work of the
// classical pi/2 reduction (21 vector instructions and one 16-byte LDS read instead of 34).  Same absolute accuracy:
// the table entries are correctly rounded and the corrections are below pi/64 in size.
// `tab` is SINCOS_TAB_DOUBLES doubles filled by load_sincos_table.
constexpr int SINCOS_TAB_DOUBLES = 128;
inline __constant__ __attribute__((aligned(16))) double c_sincos_tab[SINCOS_TAB_DOUBLES] = {AFIV_SINCOS_TAB_LIST};

__device__ __forceinline__ void load_sincos_table(double *lds_tab)
{
    for (int i = threadIdx.x; i < SINCOS_TAB_DOUBLES; i += blockDim.x) lds_tab[i] = c_sincos_tab[i];
}

__device__ __forceinline__ void sincos_tab_f64(double x, const double *__restrict__ tab, double *s_out, double *c_out)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    // k = rint(32 x / pi) by the shift trick: the sum's low mantissa bits hold k as an integer
    // (|k| < 2^51, where a float-to-int conversion would saturate at 2^31)
    const double shifted = __builtin_fma(x, AFIV_32_OVER_PI, 0x1.8p52);
    const double k = shifted - 0x1.8p52;
    double r = __builtin_fma(k, -AFIV_PIO32_1, x);
    r = __builtin_fma(k, -AFIV_PIO32_2, r);
    const int j = __double2loint(shifted) & 63;
    const d2 t = reinterpret_cast<const d2 *>(tab)[j];
    const double z = r * r;
    const coef_ptr cs = coeffs(z) + OFF_SINT;
    double ps = cs[2];
    ps = fma_sc(ps, z, cs[1]);
    ps = fma_sc(ps, z, cs[0]);
    const double sr = __builtin_fma(r * z, ps, r);              // sin r
    double pc = cs[5];
    pc = fma_sc(pc, z, cs[4]);
    pc = fma_sc(pc, z, cs[3]);
    const double dc = z * __builtin_fma(pc, z, -0.5);           // cos r - 1
    *s_out = __builtin_fma(t.x, dc, __builtin_fma(t.y, sr, t.x));
    *c_out = __builtin_fma(t.y, dc, __builtin_fma(-t.x, sr, t.y));
}

// sin, cos and cos - 1 of the same argument: the last one without the cancellation of subtracting
// 1 from a cosine near 1 (the table entry is 1 there and the polynomial gives cos r - 1 directly).
__device__ __forceinline__ void sincos_cm1_tab_f64(double x, const double *__restrict__ tab, double *s_out,
                                                   double *c_out, double *cm1_out)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double shifted = __builtin_fma(x, AFIV_32_OVER_PI, 0x1.8p52);
    const double k = shifted - 0x1.8p52;
    double r = __builtin_fma(k, -AFIV_PIO32_1, x);
    r = __builtin_fma(k, -AFIV_PIO32_2, r);
    const int j = __double2loint(shifted) & 63;
    const d2 t = reinterpret_cast<const d2 *>(tab)[j];
    const double z = r * r;
    const coef_ptr cs = coeffs(z) + OFF_SINT;
    double ps = cs[2];
    ps = fma_sc(ps, z, cs[1]);
    ps = fma_sc(ps, z, cs[0]);
    const double sr = __builtin_fma(r * z, ps, r);              // sin r
    double pc = cs[5];
    pc = fma_sc(pc, z, cs[4]);
    pc = fma_sc(pc, z, cs[3]);
    const double dc = z * __builtin_fma(pc, z, -0.5);           // cos r - 1
    *s_out = __builtin_fma(t.x, dc, __builtin_fma(t.y, sr, t.x));
    const double u = __builtin_fma(t.y, dc, -(t.x * sr));       // cos x - C
    *c_out = t.y + u;
    *cm1_out = (t.y - 1.0) + u;                                 // C - 1 is exact
}

// log(1 + f) through a 128-entry table in LDS.  u = fl(1 + f) = 2^k z with the high word of z in
// [AFIV_LOGT_OFF_HI, AFIV_LOGT_OFF_HI + 2^20), i.e. z in [0.6855, 1.371); bin i = bits 13..19 of
// that offset; the table holds (invc_i, logc_i = -log invc_i) with invc_i next to the reciprocal
// of the bin's midpoint, and
//   log(1 + f) = k log 2 + logc_i + log1p(r),   r = z invc_i - 1 + c 2^-k invc_i,  |r| <= 2^-8,
// where c = f - (u - 1) restores what the rounding of 1 + f lost.  The bin around 1.0 holds (1, 0)
// exactly, so there r = f and the result is as accurate, relatively, as the seven-term series:
// the characteristic function divides this logarithm by sigma_v^2, so relative accuracy near
// f = 0 is the point.  27 vector instructions and one 16-byte LDS read, no reciprocal, against 44
// for the atanh-series form on [sqrt(1/2), sqrt(2)); relative error <= 1 ulp (tools/gen_math_coeffs.py measures a binary64 model).
// f > -1, finite.  `tab` is LOG_TAB_DOUBLES doubles filled by load_log_table.
constexpr int LOG_TAB_DOUBLES = 256;
inline __constant__ __attribute__((aligned(16))) double c_log_tab[LOG_TAB_DOUBLES] = {AFIV_LOG_TAB_LIST};

__device__ __forceinline__ void load_log_table(double *lds_tab)
{
    for (int i = threadIdx.x; i < LOG_TAB_DOUBLES; i += blockDim.x) lds_tab[i] = c_log_tab[i];
}

__device__ __forceinline__ double log1p_tab_f64(double f, const double *__restrict__ tab)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double u = 1.0 + f;
    const double c = f - (u - 1.0);
    const int hi = __double2hiint(u);
    const int tmp = hi - AFIV_LOGT_OFF_HI;
    const int k = tmp >> 20;                                    // arithmetic shift: floor
    const int i = (tmp >> 13) & 127;
    const double z = __hiloint2double(hi - (tmp & (int)0xfff00000), __double2loint(u));
    const d2 t = reinterpret_cast<const d2 *>(tab)[i];          // (invc, logc)
    double r = __builtin_fma(z, t.x, -1.0);
    r = __builtin_fma(__builtin_ldexp(c, -k), t.x, r);
    const coef_ptr cl = coeffs(r) + OFF_LOGT;
    double p = cl[5];
    p = fma_sc(p, r, cl[4]);
    p = fma_sc(p, r, cl[3]);
    p = fma_sc(p, r, cl[2]);
    p = fma_sc(p, r, cl[1]);
    p = fma_sc(p, r, cl[0]);
    const double lg = __builtin_fma(r * r, p, r);
    const double kd = (double)k;
    return __builtin_fma(kd, AFIV_LN2_HI, t.y) + __builtin_fma(kd, AFIV_LN2_LO, lg);
}

// atan2(y, x) through a 65-row table in LDS: t = min(|x|,|y|)/max(|x|,|y|) in [0, 1], j = rint(64 t),
// h = t - j/64 (exact), atan t = sum_{k<8} A[j][k] h^k; row 0 is the odd Taylor series with exact
// 0 and 1, so small angles keep their relative accuracy.  No second reduction (the classical
// (t-1)/(t+1) step above tan(pi/8)) and a degree-7 instead of a degree-23 polynomial: 29 vector instructions and
// four 16-byte LDS reads against 47.  Relative error <= 2 ulp (the quotient's ulp included).
// Finite arguments, not both zero.  `tab` is ATAN_TAB_DOUBLES doubles filled by load_atan_table.
constexpr int ATAN_TAB_DOUBLES = 65 * 8;
inline __constant__ __attribute__((aligned(16))) double c_atan_tab[ATAN_TAB_DOUBLES] = {AFIV_ATAN_TAB_LIST};

__device__ __forceinline__ void load_atan_table(double *lds_tab)
{
    for (int i = threadIdx.x; i < ATAN_TAB_DOUBLES; i += blockDim.x) lds_tab[i] = c_atan_tab[i];
}

// atan2(y, x) given t = min(|x|,|y|) / max(|x|,|y|): the caller may have that quotient from a
// reciprocal it needs anyway.
__device__ __forceinline__ double atan2_tab_quot_f64(double y, double x, double t, const double *__restrict__ tab);

__device__ __forceinline__ double atan2_tab_f64(double y, double x, const double *__restrict__ tab)
{
    const double ax = fabs(x), ay = fabs(y);
    return atan2_tab_quot_f64(y, x, fmin(ax, ay) * rcp_f64(fmax(ax, ay)), tab);
}

__device__ __forceinline__ double atan2_tab_quot_f64(double y, double x, double t, const double *__restrict__ tab)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const double ax = fabs(x), ay = fabs(y);
    const double shifted = __builtin_fma(t, 64.0, 0x1.8p52);   // low word = rint(64 t)
    const double kd = shifted - 0x1.8p52;
    const double h = __builtin_fma(kd, -0x1p-6, t);
    const unsigned j = min((unsigned)__double2loint(shifted), 64u);   // (a NaN quotient must not index past the table)
    const d2 *row = reinterpret_cast<const d2 *>(tab) + 4 * j;
    const d2 c01 = row[0], c23 = row[1], c45 = row[2], c67 = row[3];
    double a = c67.y;
    a = __builtin_fma(a, h, c67.x);
    a = __builtin_fma(a, h, c45.y);
    a = __builtin_fma(a, h, c45.x);
    a = __builtin_fma(a, h, c23.y);
    a = __builtin_fma(a, h, c23.x);
    a = __builtin_fma(a, h, c01.y);
    a = __builtin_fma(a, h, c01.x);                             // angle of (mx, mn) in [0, pi/4]
    a = ay > ax ? 2.0 * AFIV_PIO4 - a : a;                      // first octant pair
    a = x < 0.0 ? AFIV_PI - a : a;
    return __builtin_copysign(a, y);
}

}  // namespace afivm
