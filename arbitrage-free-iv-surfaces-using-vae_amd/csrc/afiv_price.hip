// Heston price ratios over a (surface x maturity x strike) lattice on gfx950.
//
// Replaces one QuantLib AnalyticHestonEngine evaluation per option (heston_model_ql.py:87,
// :109-179) by a factored evaluation of the same quadrature:
//
//   C(K,T) = S Dq (P1 + 1/2) - K Dr (P2 + 1/2),   P_j = (1/pi) sum_n W_n Im exp(A_j(x_n) + i x_n b) / x_n
//
// with b = dd_T - log K.  A_j depends on (parameters, T, node) only, so for every (surface,
// maturity) the 288 values E = W_n exp(Re A)/(pi x_n) and alpha = Im A + x_n dd_T are computed once
// (phase 1, FP64 vector work: complex sqrt/exp/log) and all strikes follow from
//
//   sum_n E sin(alpha - x_n log K) = sum_n [E sin alpha] cos(x_n log K) - [E cos alpha] sin(x_n log K),
//
// a [rows x 576] x [576 x strikes] FP64 contraction against strike tables that depend only on the
// grid (phase 2, v_mfma_f64_16x16x4_f64).
//
// Work decomposition: one workgroup = 7 surfaces x 144 Laguerre nodes = 1008 of 1024 threads =
// 16 waves, 4 per SIMD; a thread owns one node of one surface and evaluates both integrals
// j = 1, 2.  Maturity-independent quantities (d, t1-d, p, 1/(1-p)) stay in registers.  The
// workgroup advances in steps of 2 maturities: 7 surfaces x 2 maturities fill 14 of the 16 rows
// of one MFMA row block.  Phase 1 of step G writes its 16 x 576 operand block into one half of a
// double-buffered LDS area while phase 2 of step G-1 reads the other half: each wave owns one
// (column tile, K range) of the contraction and issues its MFMA chain after the step's first or
// second characteristic-function evaluation (waves w, w+4, w+8, w+12 share a SIMD: two chains at
// a time), so the chain's table loads and LDS reads are covered by the other waves' vector work.  B
// fragments stream from the L2-resident table through raw buffer loads (one offset register
// instead of a 64-bit address pair per fragment), two batches in flight; partial tiles are
// reduced through LDS in a fixed order (bitwise reproducible results).
// The elementary functions of phase 1 (afiv_math.h) reduce their arguments against three small
// tables that live in LDS beside the operand buffers; the per-strike and per-maturity constants
// of the epilogue are staged there too, because the epilogue runs between two workgroup barriers
// where no other wave's work covers a global load.
#include <algorithm>

#include "afiv_internal.h"
#include "afiv_math.h"
#include "laguerre144.inc"

namespace {

constexpr int NQ = AFIV_NQ_NODES;          // 144
constexpr int KDIM = AFIV_KDIM;            // 576
// Workgroup shape.  The kernel needs <= 128 VGPRs for 16 waves per CU; it uses about 115 because
// polynomial coefficients are scalar operands fetched next to their use (afiv_math.h) and table
// addresses are buffer offsets.  Measured on MI355X (20 x 50 lattice, 131 072 surfaces):
// 3 surfaces x 512 threads / 5 maturities per step 15.5 ms, 5 x 768 / 3 12.6 ms, 7 x 1024 / 2 11.5 ms.
#ifndef AFIV_SPB
#define AFIV_SPB 7
#define AFIV_THREADS 1024
#define AFIV_MT 2
#endif
constexpr int SPB = AFIV_SPB;              // surfaces per workgroup
constexpr int THREADS = AFIV_THREADS;
constexpr int WAVES = THREADS / 64;
constexpr int MT = AFIV_MT;                // maturities per step
constexpr int NEVAL = 2 * MT;              // characteristic-function evaluations per thread and step
constexpr int ROWS = 16;                   // MFMA M; SPB*MT rows carry data
constexpr int XSTRIDE = KDIM + 2;          // doubles; 1156 dwords = 4 mod 64 -> conflict-free A reads
constexpr int XBUF = ROWS * XSTRIDE;       // doubles per operand buffer
static_assert(SPB * MT <= ROWS && SPB * NQ <= THREADS, "row / lane layout");

typedef double v4d __attribute__((ext_vector_type(4)));
// function tables in LDS: sincos | log | atan | exp
constexpr int EXP_TAB_AT = afivm::SINCOS_TAB_DOUBLES + afivm::LOG_TAB_DOUBLES + afivm::ATAN_TAB_DOUBLES;
constexpr int FTAB_DOUBLES = EXP_TAB_AT + afivm::EXP_TAB_DOUBLES;
#ifndef AFIV_EXP_TABLE
#define AFIV_EXP_TABLE 1
#endif

__constant__ double c_lag_x[NQ] = {AFIV_LAG_X_LIST};
__constant__ double c_lag_wp[NQ] = {AFIV_LAG_WP_LIST};   // W_n / (pi x_n)

struct Cx { double r, i; };
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ Cx crcp(Cx b)
{
    const double inv = afivm::rcp_f64(b.r * b.r + b.i * b.i);
    return {b.r * inv, -b.i * inv};
}
// sqrt(x) and 1/sqrt(x) for a positive normal x well inside the exponent range (sums of squares of
// moderate numbers here): v_rsq_f64 and two Newton steps on the pair -- a dozen instructions where
// the library's square root (scaling for subnormals, special values) and an IEEE division take
// forty.  Both results to 1 ulp.
__device__ __forceinline__ double sqrt_rsqrt(double x, double *rs)
{
    double y = __builtin_amdgcn_rsq(x);                   // ~ 2^-26 relative... refined below
    double g = x * y;                                     // ~ sqrt(x)
    double h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double e = __builtin_fma(-g, g, x);             // residual of the root
    g = __builtin_fma(e, h, g);
    *rs = 2.0 * h;
    return g;
}

__device__ __forceinline__ Cx csqrt_principal(Cx z)
{
    double unused, rs;
    const double m = sqrt_rsqrt(__builtin_fma(z.r, z.r, z.i * z.i), &unused);
    // the larger of the two parts is sqrt((m + |Re z|)/2), never a difference of close numbers; the
    // other is Im z / (2 x that): one reciprocal square root serves both
    const double big = sqrt_rsqrt(0.5 * (m + fabs(z.r)), &rs);
    const double small = 0.5 * fabs(z.i) * rs;
    Cx d;
    if (z.r >= 0.0) {
        d.r = big;
        d.i = copysign(small, z.i);
    } else {
        d.r = small;
        d.i = copysign(big, z.i);
    }
    return d;
}

// Largest batch of table loads (<= 12) that cuts n k-steps into one batch or an even number of them.
constexpr int pick_batch(int n)
{
    for (int nb = 12; nb >= 1; --nb)
        if (n % nb == 0 && (n / nb == 1 || (n / nb) % 2 == 0)) return nb;
    return 1;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ldtab(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

// Maturity-independent part of A_j(phi) for one integral: d, a = t1 - d, sum = t1 + d and
// pq = p / (1 - p) with p = a / sum (the series branch for tiny vol-of-vol keeps p itself in pq).
struct JState { Cx d, a, pq, sum; };

struct Model { double sigma2, c1, c2; bool main_branch; };

// QuantLib's Fj_Helper forms d = sqrt(t1^2 - sigma^2 phi u), u = (-phi, +-1), and then the
// differences t1 - d and t1 + d.  One of the two cancels (d -> +-t1 as sigma -> 0) and the formula
// divides by sigma^2 afterwards, which is where a double-precision evaluation picks up an
// absolute noise of eps/sigma^2.  Since (t1 - d)(t1 + d) = sigma^2 phi u exactly, the cancelling
// one is obtained here from the other by division -- same value, no cancellation.
template <int J>   // 0 -> P1, 1 -> P2
__device__ __forceinline__ JState cf_setup(double kappa, double sigma, double rho, const Model &m, double phi)
{
    const double sj = J == 0 ? 1.0 : -1.0;
    const double rsig = rho * sigma;
    const Cx t1 = {kappa - (J == 0 ? rsig : 0.0), -(rsig * phi)};
    const Cx u = {-phi, sj};
    const Cx zz = {t1.r * t1.r - t1.i * t1.i + m.sigma2 * phi * phi,
                   2.0 * t1.r * t1.i - m.sigma2 * phi * sj};
    JState s;
    s.d = csqrt_principal(zz);
    if (m.main_branch) {
        const Cx su = {m.sigma2 * phi * u.r, m.sigma2 * phi * u.i};       // sigma^2 phi u
        if (t1.r >= 0.0) {
            s.sum = {t1.r + s.d.r, t1.i + s.d.i};
            s.a = cmul(su, crcp(s.sum));
        } else {
            s.a = {t1.r - s.d.r, t1.i - s.d.i};
            s.sum = cmul(su, crcp(s.a));
        }
        // p / (1 - p) with p = a / sum: 1 - p = (sum - a) / sum = 2 d / sum, so the quotient is a / (2 d)
        s.pq = cmul(s.a, crcp(Cx{2.0 * s.d.r, 2.0 * s.d.i}));
    } else {
        // vol-of-vol <= 1e-5 series, as the reference engine switches: td = phi/(2 t1) u ;
        // p = td sigma^2/(t1+d)
        const Cx h = crcp(Cx{2.0 * t1.r, 2.0 * t1.i});
        s.a = cmul(Cx{phi * h.r, phi * h.i}, u);
        s.sum = {t1.r + s.d.r, t1.i + s.d.i};
        s.pq = cmul(Cx{s.a.r * m.sigma2, s.a.i * m.sigma2}, crcp(s.sum));
    }
    return s;
}

// 1 - exp(-d T), the quantity the characteristic function uses, without forming the difference:
// with exp(-d T) = e (cos b - i sin b), 1 - exp(-d T) = -((e - 1) + e (cos b - 1)) + i e sin b, and
// both e - 1 and cos b - 1 come out of their polynomials directly.  Subtracting the exponential
// from 1 instead costs eps / |d T| in relative accuracy: 1e-12 for a mean reversion of 0.01 and a
// maturity of three days.
__device__ __forceinline__ Cx comex_neg(Cx d, double T, const double *__restrict__ sct)
{
    double em1, sn, cs, cm1;
#if AFIV_EXP_TABLE
    const double er = afivm::exp_expm1_tab_f64(-d.r * T, sct + EXP_TAB_AT, &em1);
#else
    const double er = afivm::exp_expm1_f64(-d.r * T, &em1);
#endif
    afivm::sincos_cm1_tab_f64(d.i * T, sct, &sn, &cs, &cm1);
    return {-__builtin_fma(er, cm1, em1), er * sn};
}

// 1 - exp(-d (T + dT)) from o = 1 - exp(-d T) and f = 1 - exp(-d dT):  o + f - o f
__device__ __forceinline__ Cx comex_step(Cx o, Cx f)
{
    return {__builtin_fma(-o.r, f.r, __builtin_fma(o.i, f.i, o.r + f.r)),
            __builtin_fma(-o.r, f.i, __builtin_fma(-o.i, f.r, o.i + f.i))};
}

// NFAC = 3: one step of the lattice's program (GridView::prog) for one integral.  p = the previous
// maturity's 1 - exp(-d T), a / b = the two saved ones; op is wave-uniform.
__device__ __forceinline__ void prog_step(unsigned op, Cx &p, Cx &a, Cx &b, Cx d, double T,
                                          const double *__restrict__ sct)
{
    Cx r;
    switch (op & 15u) {
        case 1: r = comex_step(p, p); break;
        case 2: r = comex_step(p, a); break;
        case 3: r = comex_step(p, b); break;
        case 4: r = comex_step(a, a); break;
        case 5: r = comex_step(a, b); break;
        case 6: r = comex_step(b, b); break;
        default: r = comex_neg(d, T, sct); break;
    }
    p = r;
    // Wave-uniform branches around the copies.  (The empty asm statements keep them branches: as
    // plain conditional assignments through the references the compiler selects the ADDRESS and
    // moves a and b to scratch memory; as selects on the values they cost eight instructions at
    // every maturity.)
    if ((op >> 4) == 1u) {
        a = r;
        asm volatile("" : "+v"(a.r), "+v"(a.i));
    }
    if ((op >> 4) == 2u) {
        b = r;
        asm volatile("" : "+v"(b.r), "+v"(b.i));
    }
}

// One characteristic-function evaluation given omex = 1 - exp(-d T): returns (E sin alpha,
// E cos alpha) with the discount factor of its integral folded in.
__device__ __forceinline__ double2 cf_eval(const JState &s, const Model &m, double phi, double wp,
                                           double T, double ddt, double disc, Cx omex,
                                           const double *__restrict__ sct)
{
    Cx A;
    if (m.main_branch) {
        // w1 = p (1 - ex)/(1 - p), so that (1 - p ex)/(1 - p) = 1 + w1: the logarithm g = log(1 + w1)
        // keeps its relative accuracy when w1 ~ sigma^2, and the first term of A,
        // a (1 - ex)/(1 - p ex), equals (t1 + d) w1/(1 + w1) -- one complex reciprocal serves both
        const Cx w1 = cmul(s.pq, omex);
        const double q = __builtin_fma(w1.r, w1.r, __builtin_fma(w1.i, w1.i, 2.0 * w1.r));   // |1 + w1|^2 - 1
        const double zr = 1.0 + w1.r;
        // 1/|1 + w1|^2 and the arctangent's quotient min/max of (|zr|, |w1.i|) from ONE reciprocal
        // (a quarter-rate instruction and two refinement steps): R = 1/(P mx), 1/P = R mx, 1/mx = R P
        const double P = 1.0 + q;
        const double ax = fabs(zr), ay = fabs(w1.i);
        const double mx = fmax(ax, ay), mn = fmin(ax, ay);
        const double R = afivm::rcp_f64(P * mx);
        const double inv = R * mx;
        // w1 / (1 + w1) = w1 conj(1 + w1) / |1 + w1|^2 = (w1.r + |w1|^2, w1.i) / P, and |w1|^2 = q - 2 w1.r
        const Cx wz = {(q - w1.r) * inv, w1.i * inv};
        const Cx f1 = cmul(s.sum, wz);                     // s.sum carries the factor v0 / sigma^2 (see the kernel)
        const double gl2 = afivm::log1p_tab_f64(q, sct + afivm::SINCOS_TAB_DOUBLES);          // 2 Re g = log |1 + w1|^2
        const double gli = afivm::atan2_tab_quot_f64(w1.i, zr, mn * (R * P),
                                                    sct + afivm::SINCOS_TAB_DOUBLES + afivm::LOG_TAB_DOUBLES);
        A.r = f1.r + m.c2 * (s.a.r * T - gl2);
        A.i = f1.i + m.c2 * (s.a.i * T - 2.0 * gli);
    } else {
        const Cx pe = cmul(s.pq, Cx{1.0 - omex.r, -omex.i});
        const Cx f1 = cmul(cmul(s.a, omex), crcp(Cx{1.0 - pe.r, -pe.i}));
        const Cx gl = cmul(s.pq, omex);              // g = p (1 - ex)
        A.r = m.c1 * f1.r + m.c2 * (s.a.r * T - 2.0 * gl.r / m.sigma2);
        A.i = m.c1 * f1.i + m.c2 * (s.a.i * T - 2.0 * gl.i / m.sigma2);
    }
    const double alpha = A.i + phi * ddt;
#if AFIV_EXP_TABLE
    const double E = wp * afivm::exp_tab_f64(A.r, sct + EXP_TAB_AT) * disc;
#else
    const double E = wp * afivm::exp_core_f64(A.r) * disc;
#endif
    double sa, ca;
    afivm::sincos_tab_f64(alpha, sct, &sa, &ca);
    return make_double2(E * sa, E * ca);
}

// NCTP = column tiles rounded up to a power of two (1, 2, 4); the 16 waves split into NCTP column
// tiles x (16/NCTP) ranges of the 576-long contraction.
//
// NFAC = 0: exp(-d T) is evaluated for every maturity (one exp and one sincos).  NFAC = 1, 2: the
// year fractions are whole days / 365 and successive maturities are base + k days apart with
// k = 0 (NFAC = 1) or k <= 3 (NFAC = 2) -- a regular term structure such as
// round(linspace(30, 730, 20)) days, truncated as the reference truncates, steps by 36..38 days
// -- so exp(-d T_t) = exp(-d T_{t-1}) exp(-d base/365) exp(-d/365)^k costs 1 + k complex
// multiplications per maturity; the thread keeps the two factors.  What is carried is
// o = 1 - exp(-d T) (comex_neg, comex_step: o' = o + f - o f with f = 1 - exp(-d step)), so that
// the small quantity the characteristic function needs is never a difference of two numbers near
// 1.  Some 40 steps add rounding of a few eps to a value that is evaluated to 2 ulp otherwise.
//
// NFAC = 3: whole-day maturities that are sums of earlier ones (the reference's own term structure):
// 1 - exp(-d T_t) follows from two earlier values by the same product form, as the grid's program
// says; the thread keeps the previous maturity's value and two saved ones in the registers that hold
// the step factors of NFAC = 2.  A workgroup that starts in the middle of the term structure (small
// batches split their maturities over several workgroups) first replays those of the program's
// earlier steps that its own values descend from (GridView::need), exponentials only, so its
// results are those of an unsplit walk bit for bit.
//
// SYM: the strikes are symmetric about the forward-free money line (log K_k = -log K_{nK-1-k}, as
// in linspace(-a, a, n)), so cos(x log K) is shared and sin(x log K) only changes sign between a
// strike and its mirror image: with C_j = sum_n [E sin alpha] cos(x_n log K), S_j = sum_n
// [E cos alpha] sin(x_n log K), the integral is C_j - S_j at a strike and C_j + S_j at its mirror.
// The contraction then runs over half the strikes: four [16 x 144] x [144 x nH] products (C_1, C_2,
// S_1, S_2; NCTP column tiles each) instead of one [16 x 576] x [576 x nK] -- half the matrix work.
// The operand block is stored as four planes (E sin alpha | E cos alpha) x (j = 1 | 2) of 144
// columns; the strike multiplier K_k of the second integral is applied when the tiles are combined.
template <int NCTP, int NFAC, bool SYM>
__global__ __launch_bounds__(THREADS) void price_kernel(GridView g, const double *__restrict__ params,
                                                        int64_t N, double *__restrict__ ratio_out)
{
    // general: WAVES = NCTP column tiles x KS ranges of the 576-long contraction
    // SYM:     WAVES = 4 planes x NCTP column tiles x KS ranges of a plane's 144 columns
    constexpr int KS = SYM ? WAVES / (4 * NCTP) : WAVES / NCTP;
    constexpr int KLEN = (SYM ? NQ : KDIM) / KS;   // contraction indices per wave
    constexpr int NKP = 16 * NCTP;            // padded (half-)strike count used for the partial tiles
    static_assert(KS >= 1 && KS * (SYM ? 4 * NCTP : NCTP) == WAVES, "waves split evenly");
    static_assert(KLEN % 4 == 0, "K range must be a multiple of the MFMA depth");
    static_assert((SYM ? 4 : 1) * KS * ROWS * NKP <= XBUF, "partial tiles must fit in one operand buffer");

    __shared__ double lds[2 * XBUF];
    // function tables: (sin, cos)(j pi/32), the logarithm's (invc, logc), the arctangent's rows
    __shared__ __attribute__((aligned(16))) double sct[FTAB_DOUBLES];
    __shared__ double ecst[4][AFIV_MAX_STRIKES];     // K_k | Dq_t | Dr_t | 1/exp(-r tau_t) (0: expired)
    static_assert(AFIV_MAX_STRIKES == AFIV_MAX_MATURITIES, "one row length for both");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keep it scalar
    const int sl = tid / NQ;                  // < SPB: real surfaces; == SPB: spare lanes of the last wave
    const int n = tid - sl * NQ;
    const double phi = c_lag_x[n];
    const double wp = c_lag_wp[n];
    // general: (column tile, K range).  SYM: (plane = wave & 3, column tile, K range of the plane)
    const int plane = wave & 3;
    const int ct = SYM ? (wave >> 2) % NCTP : wave % NCTP;
    const int kq = SYM ? (wave >> 2) / NCTP : wave / NCTP;
    // waves w, w+4, w+8, w+12 share a SIMD: two of them issue their MFMA chains after the step's
    // first characteristic-function evaluation, the other two after the second.  A chain is a run
    // of dependent accumulations; two of them at a time fill each other's issue gaps, and none is
    // left for the end of the step, where no vector work would cover it (chains spread over all
    // four evaluations: +1 %).
    const int my_slot = (wave >> 2) & 1;
    static_assert(NEVAL >= 2, "two MFMA slots");
    const int nT = g.nT, nK = g.nK, nKpad = SYM ? g.nHpad : g.nKpad;   // nKpad: table row length
    const bool ct_live = 16 * ct < nKpad;
    // Workgroup b takes the chunks b, b + gridDim.x, ... and walks each through its maturity tiles --
    // except in batches of fewer than eight chunks per workgroup whose maturities are evaluated
    // independently (NFAC == 0) or can be re-derived (NFAC == 3 replays the steps of the program it
    // skipped).  There the batch's steps, numbered chunk by chunk and tile by tile, are dealt out as
    // contiguous ranges that differ by one step at most (the longer ones to the workgroups that start
    // first) and may begin and end inside a chunk: that evens out batches of a few chunks per CU (4096
    // surfaces of 8 x 21, two or three chunks each: -9 %) and spreads the smallest ones over more
    // workgroups than they have chunks.  With many chunks per workgroup the extra set-up at the two
    // ragged ends costs more than the partial chunk it evens out.  No bit of the result depends on
    // how the steps are dealt out.
    // (chunk numbers and step counts fit 32 bits: the launcher checks; scalar registers are scarce here)
    const int ntiles_all = (nT + MT - 1) / MT;
    const int nchunks = (int)((N + SPB - 1) / SPB);
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const bool by_step = (NFAC == 0 || NFAC == 3) && (unsigned)nchunks < 8u * nb;
    int chunk0, tile0, total, chunk_stride;
    if (by_step) {
        const unsigned units = (unsigned)nchunks * (unsigned)ntiles_all;
        const unsigned share = units / nb, extra = units - share * nb;    // the first `extra` ranges are one longer
        const unsigned u0 = b * share + (b < extra ? b : extra);
        chunk0 = (int)(u0 / (unsigned)ntiles_all);
        tile0 = (int)u0 - chunk0 * ntiles_all;
        total = (int)(share + (b < extra ? 1u : 0u));
        chunk_stride = 1;
    } else {
        chunk0 = (int)b;
        tile0 = 0;
        total = b < (unsigned)nchunks ? (int)(((unsigned)nchunks - b + nb - 1) / nb) * ntiles_all : 0;
        chunk_stride = (int)nb;
    }

    afivm::load_sincos_table(sct);
    afivm::load_log_table(sct + afivm::SINCOS_TAB_DOUBLES);
    afivm::load_atan_table(sct + afivm::SINCOS_TAB_DOUBLES + afivm::LOG_TAB_DOUBLES);
    afivm::load_exp_table(sct + EXP_TAB_AT);
    if (tid < nK) ecst[0][tid] = g.Kmult[tid];
    if (tid < nT) {
        ecst[1][tid] = g.Dq[tid];
        ecst[2][tid] = g.Dr[tid];
        // price -> ratio: divide by exp(-r tau); an expired option has NPV 0 whatever its sums
        ecst[3][tid] = g.term[tid] > 0.0 ? 1.0 / g.disc_tau[tid] : 0.0;
    }
    __syncthreads();

    Model m = {};
    JState st0 = {}, st1 = {};
    Cx ex0 = {}, ex1 = {}, fa0 = {}, fa1 = {}, fb0 = {}, fb1 = {};   // NFAC > 0: running 1 - exp(-d T), and 1 - exp(-d step) of the steps
    int prev_chunk = 0, chunk = chunk0;              // advanced with the tile counter: no division per step
    int prev_tile = tile0, tile = tile0;
    int tile_first = tile0;                          // where the walk enters the current chunk: tile0, then 0

    for (int G = 0; G <= total; ++G) {
        const bool have_cur = G < total, have_prev = G > 0;
        double *xw = lds + (G & 1) * XBUF;                  // written by this step's phase 1
        double *xr = lds + ((G + 1) & 1) * XBUF;            // operand block of the previous step

        if (have_cur && tile == tile_first) {
            int64_t s = (int64_t)chunk * SPB + sl;
            if (s >= N) s = N - 1;                           // spare lanes / ragged tail: never stored
            const double kappa = params[5 * s + 0], theta = params[5 * s + 1], sigma = params[5 * s + 2],
                         rho = params[5 * s + 3], v0 = params[5 * s + 4];
            m.sigma2 = sigma * sigma;
            m.main_branch = sigma > 1e-5;
            // (one reciprocal for both quotients: an IEEE division is twenty instructions)
            const double inv_s2 = m.main_branch ? afivm::rcp_f64(m.sigma2) : 1.0;
            m.c1 = v0 * inv_s2;
            m.c2 = (kappa * theta) * inv_s2;
            st0 = cf_setup<0>(kappa, sigma, rho, m, phi);
            st1 = cf_setup<1>(kappa, sigma, rho, m, phi);
            if (m.main_branch) {
                // t1 + d only ever multiplies w1 / (1 + w1) and v0 / sigma^2: fold the factor in once
                st0.sum = {st0.sum.r * m.c1, st0.sum.i * m.c1};
                st1.sum = {st1.sum.r * m.c1, st1.sum.i * m.c1};
            }
            if (NFAC == 3) {
                typedef const unsigned __attribute__((address_space(4))) *prog_ptr;
                const prog_ptr pw = (prog_ptr)(unsigned long long)(const void *)g.prog;
                const afivm::coef_ptr tm = (afivm::coef_ptr)(unsigned long long)(const void *)g.term;
                typedef const unsigned long long __attribute__((address_space(4))) *need_ptr;
                // only the maturities this part's values descend from
                const unsigned long long need = ((need_ptr)(unsigned long long)(const void *)g.need)[tile_first * MT];
                for (int t = 0; t < tile_first * MT; ++t) {
                    if (!((need >> t) & 1ull)) continue;
                    const unsigned op = (pw[t >> 2] >> (8 * (t & 3))) & 0xffu;
                    const double T = tm[t];
                    prog_step(op, ex0, fa0, fb0, st0.d, T, sct);
                    prog_step(op, ex1, fa1, fb1, st1.d, T, sct);
                }
            }
            if (NFAC == 1 || NFAC == 2) {
                const double T0 = g.term[0];
                ex0 = comex_neg(st0.d, T0, sct);
                ex1 = comex_neg(st1.d, T0, sct);
                fa0 = comex_neg(st0.d, g.step_dt[0], sct);
                fa1 = comex_neg(st1.d, g.step_dt[0], sct);
                if (NFAC > 1) {
                    fb0 = comex_neg(st0.d, g.step_dt[1], sct);
                    fb1 = comex_neg(st1.d, g.step_dt[1], sct);
                }
            }
        }

        v4d acc = {0.0, 0.0, 0.0, 0.0};
        const bool do_mfma = have_prev && ct_live;
        auto mfma_chain = [&]() {
            // [16 x KLEN] x [KLEN x 16] on the matrix pipe.  B fragments come from the
            // L2-resident table: they are fetched a batch ahead so that the MFMA chain
            // never waits on a load it has just issued.
            constexpr int NSTEP = KLEN / 4;
            constexpr int NB = pick_batch(NSTEP);                  // table loads in flight per batch
            constexpr int NPAIR = NSTEP / (2 * NB);                // 0: a single batch
            static_assert(NB * (NPAIR ? 2 * NPAIR : 1) == NSTEP, "one batch or pairs of batches");
            const int kfirst = (SYM ? plane * NQ : 0) + kq * KLEN;          // first operand column of this wave
            const double *xa = xr + (lane & 15) * XSTRIDE + kfirst + (lane >> 4);
            // Table fragments through a buffer resource: one 32-bit per-lane offset plus a
            // scalar offset per load (plain pointers cost a 64-bit address register pair per
            // load here).  Reads past the table return 0.
            // SYM: table rows are [cos | sin][144]; planes 0, 1 (E sin alpha) take cos, planes 2, 3 sin
            const int tfirst = SYM ? (plane >> 1) * NQ + kq * KLEN : kfirst;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double *>(SYM ? g.tab_sym : g.tab), 0, (SYM ? 2 * NQ : KDIM) * nKpad * (int)sizeof(double),
                0x00020000);
            const unsigned lo = (unsigned)(((tfirst + (lane >> 4)) * nKpad + 16 * ct + (lane & 15)) * 8);
            const unsigned kstride = (unsigned)(4 * nKpad * 8);     // bytes per MFMA k-step
            double b0[NB], b1[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) b0[i] = ldtab(rs, lo, kstride * i);
            if (NPAIR == 0) {
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * i], b0[i], acc, 0, 0, 0);
            }
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const int base = 2 * pr * NB;
#pragma unroll
                for (int i = 0; i < NB; ++i) b1[i] = ldtab(rs, lo, kstride * (base + NB + i));
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * (base + i)], b0[i], acc, 0, 0, 0);
                if (pr + 1 < NPAIR) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) b0[i] = ldtab(rs, lo, kstride * (base + 2 * NB + i));
                }
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[4 * (base + NB + i)], b1[i], acc, 0, 0, 0);
            }
        };
        // per-maturity constants (wave-uniform -> scalar registers), fetched one maturity ahead
        // (term, dd, Dq, Dr are consecutive [nT] arrays of the grid block: one base pointer)
        // read through the constant address space: the grid block is immutable, and wave-uniform
        // loads from there are scalar loads into scalar registers (as plain global pointers they
        // are vector loads of 64 identical values, waited for at the top of every maturity)
        const afivm::coef_ptr mc = (afivm::coef_ptr)(unsigned long long)(const void *)g.term;
        typedef const unsigned __attribute__((address_space(4))) *prog_ptr;
        const prog_ptr progw = (prog_ptr)(unsigned long long)(const void *)g.prog;
#pragma unroll 1
        for (int tl = 0; tl < MT; ++tl) {
            const int t = tile * MT + tl;
            const int tc = min(t, nT - 1);
            const double T = mc[tc], ddt = mc[nT + tc], Dq = mc[2 * nT + tc], Dr = mc[3 * nT + tc];
            // extra days on top of the base step that lead to maturity t (wave-uniform)
            const int extra = NFAC == 2 ? (int)((g.step_sel >> (2 * t)) & 3ull) : 0;
            const unsigned op = NFAC == 3 ? (progw[tc >> 2] >> (8 * (tc & 3))) & 0xffu : 0u;
            if (have_cur) {
                double2 x0 = make_double2(0.0, 0.0);
                if (t < nT) {
                    if (NFAC == 0) ex0 = comex_neg(st0.d, T, sct);
                    else if (NFAC == 3) prog_step(op, ex0, fa0, fb0, st0.d, T, sct);
                    else if (t > 0) {
                        ex0 = comex_step(ex0, fa0);
                        for (int e = 0; e < extra; ++e) ex0 = comex_step(ex0, fb0);
                    }
                    x0 = cf_eval(st0, m, phi, wp, T, ddt, Dq, ex0, sct);
                }
                // The previous step's epilogue read its partial tiles from this buffer: every wave
                // must be past it before the first operand of this step is stored.  Placed here, one
                // evaluation into the step, the barrier finds the waves already through.
                if (tl == 0) __syncthreads();
                if (sl < SPB) {   // spare lanes of the last wave own no row
                    if (SYM) {
                        xw[(sl * MT + tl) * XSTRIDE + 0 * NQ + n] = x0.x;
                        xw[(sl * MT + tl) * XSTRIDE + 2 * NQ + n] = x0.y;
                    } else {
                        *reinterpret_cast<double2 *>(&xw[(sl * MT + tl) * XSTRIDE + 2 * n]) = x0;
                    }
                }
            }
            if (do_mfma && 2 * tl == my_slot) mfma_chain();
            if (have_cur) {
                double2 x1 = make_double2(0.0, 0.0);
                if (t < nT) {
                    if (NFAC == 0) ex1 = comex_neg(st1.d, T, sct);
                    else if (NFAC == 3) prog_step(op, ex1, fa1, fb1, st1.d, T, sct);
                    else if (t > 0) {
                        ex1 = comex_step(ex1, fa1);
                        for (int e = 0; e < extra; ++e) ex1 = comex_step(ex1, fb1);
                    }
                    x1 = cf_eval(st1, m, phi, wp, T, ddt, Dr, ex1, sct);
                }
                if (sl < SPB) {
                    if (SYM) {
                        xw[(sl * MT + tl) * XSTRIDE + 1 * NQ + n] = x1.x;
                        xw[(sl * MT + tl) * XSTRIDE + 3 * NQ + n] = x1.y;
                    } else {
                        *reinterpret_cast<double2 *>(&xw[(sl * MT + tl) * XSTRIDE + 2 * (NQ + n)]) = x1;
                    }
                }
            }
            if (do_mfma && 2 * tl + 1 == my_slot) mfma_chain();
        }
        __syncthreads();          // operand block G complete; every MFMA chain of step G-1 done
        if (have_prev && ct_live) {
            // D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
            double *pw = xr + (((SYM ? plane * KS : 0) + kq) * ROWS) * NKP + 16 * ct + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) pw[((lane >> 4) + 4 * i) * NKP] = acc[i];
        }
        __syncthreads();
        // One output item per thread.  Everything an item needs besides the partial tiles comes
        // from the constants staged in LDS at kernel start and from 32-bit index arithmetic on
        // top of one wave-uniform 64-bit base: between the two barriers around this block every
        // wave waits, so a chain of table loads from global memory here is exposed latency.
        static_assert(ROWS * NKP <= THREADS, "one output item per thread");
        if (have_prev) {
            const int per = nT * nK;
            const int64_t left = N - (int64_t)prev_chunk * SPB;
            const int nvalid = left < SPB ? (int)left : SPB;              // surfaces of the previous chunk
            double *__restrict__ out = ratio_out + (int64_t)prev_chunk * (int64_t)(SPB * per);
            // The item's indices are re-derived from the thread index at every step: hoisted out
            // of the loop they would be spilled and come back through scratch loads, here, where
            // nothing covers their latency.
            int o = tid;
            asm volatile("" : "+v"(o));
            if (SYM) {
              if (64 * wave < SPB * MT * g.nH) {
                // item = (row, half-strike k), numbered densely over the rows and half-strikes that
                // exist, so that whole waves past the last item skip the block (the float quotient is
                // exact: o + 1/2 is at least 1/(2 nH) away from a multiple of nH).  A thread sums the
                // item's partial tiles once and writes both the strike and its mirror image: half the
                // LDS reads and index arithmetic of one thread per output, and half the waves.
                const int nH = g.nH;
                const int row = (int)(((float)o + 0.5f) * (1.0f / (float)nH));
                const int k = o - __mul24(row, nH);
                const int km = nK - 1 - k;                         // the mirror image's strike index
                const int rs = row / MT;
                const int t = prev_tile * MT + (row - rs * MT);
                if (row < SPB * MT && rs < nvalid && t < nT) {
                    double v[4];
#pragma unroll
                    for (int pl = 0; pl < 4; ++pl) {
                        double sum = 0.0;
#pragma unroll
                        for (int q = 0; q < KS; ++q) sum += xr[((pl * KS + q) * ROWS + row) * NKP + k];
                        v[pl] = sum;
                    }
                    // planes: 0 = C_1, 1 = C_2, 2 = S_1, 3 = S_2
                    const double Dq = ecst[1][t], Dr = ecst[2][t], inv = ecst[3][t];
                    double *__restrict__ dst = out + (__mul24(rs, per) + __mul24(t, nK));
                    {
                        const double K = ecst[0][k];
                        const double call = (0.5 * Dq - 0.5 * (K * Dr)) + ((v[0] - v[2]) - K * (v[1] - v[3]));
                        dst[k] = inv != 0.0 ? call * inv : 0.0;
                    }
                    if (km != k) {
                        const double K = ecst[0][km];
                        const double call = (0.5 * Dq - 0.5 * (K * Dr)) + ((v[0] + v[2]) - K * (v[1] + v[3]));
                        dst[km] = inv != 0.0 ? call * inv : 0.0;
                    }
                }
              }
            } else {
                const int row = o / NKP;
                const int col = o - row * NKP;
                const int rs = row / MT;
                const int t = prev_tile * MT + (row - rs * MT);
                if (row < ROWS && rs < nvalid && t < nT && col < nK) {
                    double sum = 0.0;
#pragma unroll
                    for (int q = 0; q < KS; ++q) sum += xr[(q * ROWS + row) * NKP + col];
                    const double Dq = ecst[1][t], Dr = ecst[2][t], K = ecst[0][col], inv = ecst[3][t];
                    const double call = (0.5 * Dq - 0.5 * (K * Dr)) + sum;
                    out[__mul24(rs, per) + __mul24(t, nK) + col] = inv != 0.0 ? call * inv : 0.0;
                }
            }
        }
        prev_chunk = chunk;
        prev_tile = tile;
        if (++tile == ntiles_all) {
            tile = 0;
            tile_first = 0;
            chunk += chunk_stride;
        }
    }
}

// Unit-test hook: the elementary functions above, element-wise.
__global__ void debug_math_kernel(int op, const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                                  double *__restrict__ out0, double *__restrict__ out1)
{
    __shared__ __attribute__((aligned(16))) double sct[FTAB_DOUBLES];
    afivm::load_sincos_table(sct);
    afivm::load_log_table(sct + afivm::SINCOS_TAB_DOUBLES);
    afivm::load_atan_table(sct + afivm::SINCOS_TAB_DOUBLES + afivm::LOG_TAB_DOUBLES);
    afivm::load_exp_table(sct + EXP_TAB_AT);
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0, b = 0.0;
        switch (op) {
            case 0: a = afivm::exp_f64(x[i]); break;
            case 1: afivm::sincos_tab_f64(x[i], sct, &a, &b); break;
            case 2: a = afivm::log1p_tab_f64(x[i], sct + afivm::SINCOS_TAB_DOUBLES); break;
            case 3: a = afivm::atan2_tab_f64(x[i], y[i], sct + afivm::SINCOS_TAB_DOUBLES + afivm::LOG_TAB_DOUBLES); break;
            case 4: a = afivm::rcp_f64(x[i]); break;
            case 6: b = afivm::exp_expm1_f64(x[i], &a); break;                    // out0 = exp(x) - 1, out1 = exp(x)
            case 8: a = afivm::exp_tab_f64(x[i], sct + EXP_TAB_AT); break;
            case 9: b = afivm::exp_expm1_tab_f64(x[i], sct + EXP_TAB_AT, &a); break;   // out0 = exp(x) - 1, out1 = exp(x)
            case 7: { double sn, cs; afivm::sincos_cm1_tab_f64(x[i], sct, &sn, &cs, &a); b = cs; break; }   // cos(x) - 1, cos(x)
            default: break;
        }
        out0[i] = a;
        if (out1) out1[i] = b;
    }
}

}  // namespace

int afiv_launch_debug_math(int op, const double *x, const double *y, int64_t n, double *out0, double *out1,
                           hipStream_t s)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(debug_math_kernel, dim3((unsigned)blocks), dim3(256), 0, s, op, x, y, n, out0, out1);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

int afiv_launch_price(const GridView &g, const double *params, int64_t N, double *ratio_out, hipStream_t s)
{
    if (N <= 0) return 0;
    const int nct = g.nKpad / 16;
    if (nct < 1 || nct > 4) return AFIV_E_SHAPE;
    // The kernel counts chunks of seven surfaces and its own steps in 32 bits: batches beyond 2^24
    // chunks (117 million surfaces) go as several launches.
    constexpr int64_t MAX_LAUNCH = (int64_t)SPB << 24;
    if (N > MAX_LAUNCH) {
        const int64_t per = (int64_t)g.nT * g.nK;
        for (int64_t off = 0; off < N; off += MAX_LAUNCH) {
            const int64_t n = N - off < MAX_LAUNCH ? N - off : MAX_LAUNCH;
            if (const int rc = afiv_launch_price(g, params + 5 * off, n, ratio_out + off * per, s)) return rc;
        }
        return AFIV_OK;
    }
    const int64_t chunks = (N + SPB - 1) / SPB;
    // One workgroup per CU is resident (LDS).  A workgroup's walk ends with one step that only
    // drains the pipeline (the last tile's contraction and epilogue), and a batch ends when the
    // last workgroup does: with fewer than 64 chunks per CU both are worth more than the dynamic
    // balancing of many short workgroups, so each CU gets ONE workgroup and an equal share of the
    // steps (50 000 surfaces on 8 x 21: -3 %); large batches keep eight rounds of workgroups.
    // The share is counted in steps (a chunk's maturity tiles) where the kernel can enter a chunk at
    // any maturity -- no stepped exponential -- and in chunks otherwise: a batch of fewer chunks than
    // CUs is then spread over up to one workgroup per step.
    const int ntiles = (g.nT + MT - 1) / MT;
    const int64_t units = (g.step_count == 0 || g.step_count == 3) ? chunks * ntiles : chunks;
    const int64_t blocks = chunks < 256 * 64 ? std::min<int64_t>(units, 256) : 256 * 8;
    dim3 grid((unsigned)blocks), block(THREADS);
#define AFIV_LAUNCH_PRICE(NCT, NF, SY) \
    hipLaunchKernelGGL((price_kernel<NCT, NF, SY>), grid, block, 0, s, g, params, N, ratio_out)
#define AFIV_LAUNCH_PRICE_NF(NCT, SY)                        \
    do {                                                     \
        if (g.step_count == 1) AFIV_LAUNCH_PRICE(NCT, 1, SY); \
        else if (g.step_count == 2) AFIV_LAUNCH_PRICE(NCT, 2, SY); \
        else if (g.step_count == 3) AFIV_LAUNCH_PRICE(NCT, 3, SY); \
        else AFIV_LAUNCH_PRICE(NCT, 0, SY);                  \
    } while (0)
    if (g.sym) {
        if (g.nHpad == 16) AFIV_LAUNCH_PRICE_NF(1, true);
        else AFIV_LAUNCH_PRICE_NF(2, true);
    } else if (nct == 1) AFIV_LAUNCH_PRICE_NF(1, false);
    else if (nct == 2) AFIV_LAUNCH_PRICE_NF(2, false);
    else AFIV_LAUNCH_PRICE_NF(4, false);
#undef AFIV_LAUNCH_PRICE_NF
#undef AFIV_LAUNCH_PRICE
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}
