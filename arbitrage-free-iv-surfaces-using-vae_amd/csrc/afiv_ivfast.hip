// Implied volatility for the forward Black-Scholes form (variant A) and the arbitrage checks on
// the re-priced vols, one wave per surface, lanes marching through the lattice in natural order.
//
// Compiled with -ffp-contract=off (see afiv_bs.h).
//
// What is kept from the reference (generate_iv_surface_by_date.py:296-318): the guard, the bracket
// [0.001, 3.0] and scipy.optimize.brentq's rules for it -- f(0.001) == 0 returns 0.001, equal signs
// at the two ends give NaN -- evaluated with the very objective the Brent kernel (afiv_ivarb.hip)
// uses, so the NaN mask is that kernel's bit for bit.  What differs is how the root inside a valid
// bracket is found.  Variant A's objective is increasing in the vol, so any method that converges
// lands within the reference's xtol = 1e-6 of brentq's answer; this kernel takes
//   1. a closed-form first guess: the NORMAL-model (Bachelier) implied vol of the option's
//      out-of-the-money value -- a one-variable inverse, two small float32 tables
//      (tools/gen_bachelier_table.py) -- converted to a lognormal vol by the first term of the
//      normal/lognormal expansion, s_LN = s_N (theta/m) / (1 - s_N^2/24): relative error 1e-6 at the
//      median and below 1e-4 for total vols under 1;
//   2. third-order Householder steps on the objective itself (its derivatives are closed forms of
//      d1, d2 and the exponential the evaluation has computed anyway), each candidate kept inside
//      the bracket (bisection otherwise), until the Newton step is below 1e-8, the objective is
//      zero to the rounding of its own terms, the bracket is narrower than brentq's xtol -- or the
//      step itself is so small that the vol it lands on is within 1e-9 of the root, which is then
//      returned without a further evaluation.
// One evaluation per point on the benchmark lattice (two where the guess is more than 1e-3 off)
// instead of Brent's ten, the same for every lane: a wave solves 64 consecutive lattice points in
// lockstep, reads and writes them coalesced, and compares neighbours (strike k-1, maturity t-1)
// through cross-lane moves instead of staging the surface in LDS.
//
// Where the objective is flat to rounding (deep in the money, a time value below one ulp of the
// price) every vol of a whole interval is a root in double precision; brentq's pick depends on its
// path, this kernel returns the one its guess lands on.  Both re-price to the target to within the
// rounding of the objective.
#include <algorithm>
#include "afiv_bs.h"
#include "bachelier_table.inc"

namespace {
using namespace afivbs;

constexpr int FAST_THREADS = 512;
constexpr int FAST_WAVES = FAST_THREADS / 64;
// A vol that a small final step lands on is returned unevaluated when its re-priced value is within
// LAND_PRICE_ERR of the target price (which then stands in for it in the checks); a check whose two
// values are closer than CLOSE_CALL to flipping is repeated with exactly re-priced values.
constexpr double LAND_PRICE_ERR = 1e-8, CLOSE_CALL = 4e-8;
constexpr int BACH_ROW = 8;                       // 5 coefficients padded to 32-byte rows
constexpr int BACH_ROWS = AFIV_BACH_NSEG_A + AFIV_BACH_NSEG_B;
static_assert(AFIV_BACH_NCOEF == 5, "bachelier table degree");
__constant__ float c_bach_a[AFIV_BACH_NSEG_A * AFIV_BACH_NCOEF] = {AFIV_BACH_TABLE_A};
__constant__ float c_bach_b[AFIV_BACH_NSEG_B * AFIV_BACH_NCOEF] = {AFIV_BACH_TABLE_B};

typedef float f4v __attribute__((ext_vector_type(4)));

struct FastMat {
    double tau, sqrt_tau, disc, F;
    double kv;          // vega = kv * exp(-d1^2/2)/2:  disc F sqrt(tau) 2/sqrt(2 pi)
    double ffloor;      // rounding level of the objective's terms: 8 eps disc F
    float rsqrt_tau, inv_disc;
};

__device__ __forceinline__ float bach_poly(const float *__restrict__ rows, float t, int nseg)
{
    // fmaxf / fminf return the other operand for a NaN: the index is always inside the table
    const float tc = fminf(fmaxf(t, 0.0f), (float)nseg - 0.001f);
    const int seg = (int)tc;
    const float f = tc - (float)seg - 0.5f;
    const float *r = rows + seg * BACH_ROW;
    const f4v c = *reinterpret_cast<const f4v *>(r);
    float p = r[4];
    p = fmaf(p, f, c.w);
    p = fmaf(p, f, c.z);
    p = fmaf(p, f, c.y);
    return fmaf(p, f, c.x);
}

// First guess for the root of BS(sigma) = target: see the file comment.  otm = the target's
// out-of-the-money (time) value, discounted.  Everything in float: a guess needs 1e-5.
__device__ __forceinline__ double first_guess(double otm, double F, double K, double lnfk, const FastMat &m,
                                              const float *__restrict__ bach)
{
    const float th = fabsf((float)lnfk);
    const float th2 = th * th;
    const float ser = fmaf(th2, fmaf(th2, 1.0f / 1920.0f, 1.0f / 24.0f), 1.0f);   // sinh(th/2)/(th/2)
    const float mny = th * ser;                                                  // |F - K| / sqrt(F K)
    const float inv_norm = __builtin_amdgcn_rsqf((float)F * (float)K) * m.inv_disc;
    const double beta_d = otm * (double)inv_norm;            // option value in units of sqrt(F K)
    const float beta = (float)beta_d;
    const float w = fmaf(0.5f, mny, beta);
    const float u = mny * __builtin_amdgcn_rcpf(w);
    const float s_a = w * bach_poly(bach, u * ((float)AFIV_BACH_NSEG_A / AFIV_BACH_UA), AFIV_BACH_NSEG_A);
    // log(beta / mny) from the double's exponent and mantissa: beta may be far below float's range
    const double rho = beta_d * (double)__builtin_amdgcn_rcpf(mny);
    const float lg2 = (float)__builtin_amdgcn_frexp_exp(rho) + __builtin_amdgcn_logf((float)__builtin_amdgcn_frexp_mant(rho));
    const float zeta = __builtin_amdgcn_sqrtf(fmaxf(-2.0f * 0.693147181f * lg2, 0.0f));
    const float s_b = mny * bach_poly(bach + AFIV_BACH_NSEG_A * BACH_ROW,
                                      (zeta - AFIV_BACH_ZL) * ((float)AFIV_BACH_NSEG_B / (AFIV_BACH_ZH - AFIV_BACH_ZL)),
                                      AFIV_BACH_NSEG_B);
    const float s_n = u <= 1.7f ? s_a : s_b;                 // normal total vol, units of sqrt(F K)
    const float s_ln = s_n * __builtin_amdgcn_rcpf(ser * fmaf(s_n * s_n, -1.0f / 24.0f, 1.0f));
    return (double)(s_ln * m.rsqrt_tau);
}

// v_rcp_f64 and one Newton step: 1e-10 relative, ample for a root finder's step
__device__ __forceinline__ double rcp_step(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
}

// Four waves per SIMD (128 VGPRs; the compiler's own choice is 130 and three waves), two workgroups
// per CU: the lattice's tables are shared by eight surfaces in flight.
template <bool CHECKS>
__global__ __launch_bounds__(FAST_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void iv_fast_kernel(GridView g, const double *__restrict__ ratio, int64_t N, double spot,
                    double *__restrict__ iv_out, int32_t *__restrict__ flags)
{
    extern __shared__ __align__(16) double dyn[];
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    double *tab = dyn;                                                         // ERFCX_LDS
    FastMat *mat = reinterpret_cast<FastMat *>(tab + ERFCX_LDS);               // [64]
    double *strike = reinterpret_cast<double *>(mat + AFIV_MAX_MATURITIES);    // [64]
    d2v *ends = reinterpret_cast<d2v *>(strike + AFIV_MAX_STRIKES);            // [per] objective at 0.001 and 3.0
    d2v *fk = ends + per;                                                      // [per] log(F/K), F/K
    float *bach = reinterpret_cast<float *>(fk + per);                         // [BACH_ROWS][BACH_ROW]
    // per wave and maturity: negative-butterfly count, and whether the row holds a NaN / negative ratio
    __shared__ int fly_all[FAST_WAVES][AFIV_MAX_MATURITIES];
    __shared__ int skip_all[FAST_WAVES][AFIV_MAX_MATURITIES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: the surface index and its pointers stay scalar
    int *skip_t = skip_all[wave];
    int *fly_t = fly_all[wave];

    load_erfcx_table(tab);
    for (int i = tid; i < BACH_ROWS * BACH_ROW; i += FAST_THREADS) {
        const int row = i / BACH_ROW, j = i - row * BACH_ROW;
        float v = 0.0f;
        if (j < AFIV_BACH_NCOEF)
            v = row < AFIV_BACH_NSEG_A ? c_bach_a[row * AFIV_BACH_NCOEF + j]
                                       : c_bach_b[(row - AFIV_BACH_NSEG_A) * AFIV_BACH_NCOEF + j];
        bach[i] = v;
    }
    for (int t = tid; t < nT; t += FAST_THREADS) {
        FastMat m;
        m.tau = g.tau[t];
        m.sqrt_tau = sqrt(m.tau);
        m.disc = g.disc_tau[t];
        m.F = spot * g.fwd_mult[t];
        m.kv = m.disc * m.F * m.sqrt_tau * 0.7978845608028654;
        m.ffloor = 8.0 * 2.220446049250313e-16 * m.disc * m.F;
        m.rsqrt_tau = (float)(1.0 / m.sqrt_tau);
        m.inv_disc = (float)(1.0 / m.disc);
        mat[t] = m;
    }
    for (int k = tid; k < nK; k += FAST_THREADS) strike[k] = spot * g.Kmult[k];
    __syncthreads();

    const float inv_nK = 1.0f / (float)nK;
    const double nan = __builtin_nan("");
    const afivm::ExpCoef ec = afivm::load_exp_coef();
    const double xa = BRENT_XA, xb = BRENT_XB;
    // the objective at the two bracket ends, and the per-point constants, for every lattice point
    for (int i = tid; i < per; i += FAST_THREADS) {
        const int t = small_div(i, inv_nK), k = i - __mul24(t, nK);
        const BsPoint c = make_point(g, t, k, spot, AFIV_VARIANT_A);
        d2v e, f;
        e.x = bs_objective<AFIV_VARIANT_A>(c, xa, tab, ec);
        e.y = bs_objective<AFIV_VARIANT_A>(c, xb, tab, ec);
        f.x = c.lnFK;
        f.y = c.ratio;
        ends[i] = e;
        fk[i] = f;
    }
    __syncthreads();

    const int nbatch = (per + 63) >> 6;
    for (int64_t s = (int64_t)blockIdx.x * FAST_WAVES + wave; s < N; s += (int64_t)gridDim.x * FAST_WAVES) {
        const double *src = ratio + s * per;
        double *dst = iv_out + s * per;
        if (CHECKS) {
            // Maturities whose ratios hold a NaN or a negative value take no part in the butterfly
            // rule, which is only known once the whole row has been seen: negative butterflies are
            // counted per maturity (both events are rare, so the LDS traffic is too) and the rows
            // are sorted out after the surface's last point.  Each ratio is read from memory once.
            __builtin_amdgcn_wave_barrier();
            skip_t[lane] = 0;
            fly_t[lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        int n_nan = 0, n_bad = 0, n_stat = 0, n_cal = 0;
        double p_prev = nan;                                 // re-priced values of the previous 64 points
        unsigned long long landed_prev = 0;                  // ... and which of them are target prices of landed vols
        for (int b = 0; b < nbatch; ++b) {
            const int i = (b << 6) + lane;
            const bool live = i < per;
            const int ic = live ? i : per - 1;
            const int t = small_div(ic, inv_nK), k = ic - __mul24(t, nK);
            const FastMat m = mat[t];
            const d2v pe = ends[ic], pf = fk[ic];
            BsPoint c;
            c.tau = m.tau;
            c.sqrt_tau = m.sqrt_tau;
            c.disc = m.disc;
            c.F = m.F;
            c.K = strike[k];
            c.lnFK = pf.x;
            c.ratio = pf.y;
            const double y = src[ic];
            const double hp = y * spot * c.disc;
            c.target = hp;
            const double intrinsic = fmax(c.F - c.K, 0.0) * c.disc;
            const bool ok = !(hp < intrinsic * 0.99);
            // brentq up to its loop
            const double fa = pe.x - hp, fb = pe.y - hp;
            double res = nan, pr = hp;
            bool go = false, landed = false;
            if (ok && !isnan(fa) && !isnan(fb)) {
                if (fa == 0) res = xa;
                else if (fb == 0) res = xb;
                else if (opposite_sign(fa, fb)) go = live;
            }
            double sig = first_guess(hp - intrinsic, c.F, c.K, c.lnFK, m, bach);
            if (!(sig > xa && sig < xb)) sig = 0.5 * (xa + xb);
            double lo = xa, hi = xb;
            const bool lo_neg = fa < 0;                       // sign of the objective at the low end
            int it = 0;
            while (__ballot(go)) {
                BsAux ax;
                const double bsv = bs_objective<AFIV_VARIANT_A>(c, sig, tab, ec, &ax);
                const double f = bsv - hp;
                if (go) {
                    ++it;
                    const double v = m.kv * ax.e;
                    const double af = fabs(f);
                    // on the low side of the root?  (f == 0 is accepted below)
                    const bool low = (f < 0) == lo_neg;
                    lo = low ? sig : lo;
                    hi = low ? hi : sig;
                    const bool fail = isnan(f) || it >= BRENT_MAXITER;
                    if (fail || af <= 1e-8 * v || af <= m.ffloor || hi - lo < 1e-6) {
                        res = fail ? nan : sig;
                        pr = fail ? hp : bsv;
                        go = false;
                    } else {
                        // Householder's third-order step: nu (1 + g nu / 2) / (1 + g nu + d nu^2 / 6) with
                        // nu = -f/f', g = f''/f' = d1 d2 / sigma, d = f'''/f' = g^2 - (d1^2 + d1 d2 + d2^2)/sigma^2
                        // -- multiplied through by (sigma f')^2, so that ONE reciprocal (a quarter-rate
                        // instruction and its refinement) serves where nu, 1/sigma and the quotient took three:
                        //   n sigma (a + dd n / 2) / (a^2 + dd n a + q n^2 / 6),   n = -f, a = sigma f', dd = d1 d2,
                        //   q = dd^2 - d1^2 - dd - d2^2.  A vanishing vega gives 0/0: the bracket test below
                        // sends the NaN to bisection.
                        const double n = -f;
                        const double a = sig * v;
                        const double dd = ax.d1 * ax.d2;
                        const double q = dd * dd - (ax.d1 * ax.d1 + dd + ax.d2 * ax.d2);
                        const double tn = dd * n;
                        const double num = (n * sig) * (a + 0.5 * tn);
                        const double den = a * (a + tn) + q * ((n * n) * (1.0 / 6.0));
                        const double step = num * rcp_step(den);
                        const double cand = sig + step;
                        const bool inside = cand > lo && cand < hi;
                        // A third-order step from an evaluated point leaves an error of the order
                        // step (g step)^3, g ~ max(d1^2, d2^2) / sigma the scale of the objective's
                        // derivative ratios: a step below 1e-3 with g step below 1e-2 lands within 1e-9 of
                        // the root, and the stepped-to vol is returned without a second evaluation.  Its
                        // Black-Scholes value for the checks is then the target price itself: they
                        // differ by vega x 1e-9 at most (in practice by rounding: the typical step is
                        // 1e-6 of the vol), where the reference's own re-priced value is vega x 2e-7
                        // from the target -- brentq stops that far from the root.
                        const double dmax = fmax(ax.d1 * ax.d1, ax.d2 * ax.d2);
                        const double as = fabs(step), gs = dmax * as;
                        // ... and in PRICE units, so that the rule does not loosen as the spot grows (the
                        // checks compare re-priced values against an absolute 1e-6): vega x that error,
                        // v as (gs / sigma)^3, stays below LAND_PRICE_ERR
                        const double s3 = (sig * sig) * sig, g3 = (gs * gs) * gs;
                        if (inside && as <= 1e-3 && gs <= 1e-2 * sig && (v * as) * g3 <= LAND_PRICE_ERR * s3) {
                            res = cand;
                            pr = hp;
                            landed = true;
                            go = false;
                        } else {
                            sig = inside ? cand : 0.5 * (lo + hi);
                        }
                    }
                }
            }
            if (live) dst[i] = res;
            if (CHECKS) {
                if (live) {
                    if (isnan(res)) { n_nan++; n_bad++; }
                    else if (res <= 0) n_bad++;
                }
                // static: strikes k-1, k of one maturity; calendar: maturities t-1, t of one strike.
                // The neighbours sit 1 and nK lattice points back: in this batch or the previous one.
                double p_cur = live ? pr : nan;
                const int back = (lane - nK) & 63;
                bool v_stat, v_cal, close;
                auto compare = [&]() {
                    const double up1 = __shfl(p_cur, (lane + 63) & 63, 64);
                    const double last = __shfl(p_prev, 63, 64);   // (cross-lane reads stay outside the selects:
                    const double left = lane == 0 ? last : up1;   //  a read under a lane mask returns 0 for the others)
                    const double c_cur = __shfl(p_cur, back, 64), c_prev = __shfl(p_prev, back, 64);
                    const double below = lane >= nK ? c_cur : c_prev;
                    const bool hs = live && k >= 1, hc = live && t >= 1;
                    // left < p_cur - 1e-6 and below > p_cur + 1e-6 as the signs of two differences (a
                    // difference of doubles has the sign of the comparison), whose sizes say how far
                    // each comparison is from flipping
                    const double ds = (p_cur - 1e-6) - left, dc = below - (p_cur + 1e-6);
                    v_stat = hs && ds > 0;
                    v_cal = hc && dc > 0;
                    close = (hs && fabs(ds) <= CLOSE_CALL) || (hc && fabs(dc) <= CLOSE_CALL);
                };
                compare();
                const unsigned long long landed_now = __ballot(landed);
                if (__ballot(close) && (landed_now | landed_prev)) {
                    // A comparison sits within CLOSE_CALL of its threshold and one of the two values
                    // may be a target price standing in for the Black-Scholes value of a landed vol
                    // (they differ by LAND_PRICE_ERR at most): re-price the landed vols of this batch
                    // and the previous one exactly and compare again.  Rare: the whole wave takes it.
                    auto reprice = [&](int idx, double vol) {
                        const int tt = small_div(idx, inv_nK), kk = idx - __mul24(tt, nK);
                        const FastMat mm = mat[tt];
                        const d2v ff = fk[idx];
                        BsPoint cc;
                        cc.tau = mm.tau;
                        cc.sqrt_tau = mm.sqrt_tau;
                        cc.disc = mm.disc;
                        cc.F = mm.F;
                        cc.K = strike[kk];
                        cc.lnFK = ff.x;
                        cc.ratio = ff.y;
                        return bs_objective<AFIV_VARIANT_A>(cc, vol, tab, ec);
                    };
                    const double e_cur = reprice(ic, landed ? res : 0.5);
                    if (landed) p_cur = e_cur;
                    const bool was = b > 0 && ((landed_prev >> lane) & 1ull);
                    const int ip = was ? i - 64 : 0;
                    const double e_prev = reprice(ip, was ? dst[ip] : 0.5);     // this lane's own store of the last batch
                    if (was) p_prev = e_prev;
                    compare();
                }
                landed_prev = landed_now;
                if (v_stat) n_stat++;
                if (v_cal) n_cal++;
                if (live && (isnan(y) || y < 0)) skip_t[t] = 1;         // benign race: all writers store 1
                if (live && k + 2 < nK) {
                    const double bf = y - 2 * src[i + 1] + src[i + 2];  // the same cache lines as y
                    if (bf < 0) atomicAdd(&fly_t[t], 1);
                }
                p_prev = p_cur;
            }
        }
        if (CHECKS) {
            n_nan = wave_sum(n_nan);
            n_bad = wave_sum(n_bad);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int n_fly = wave_sum(lane < nT && !skip_t[lane] ? fly_t[lane] : 0);
            n_stat = wave_sum(n_stat);
            n_cal = wave_sum(n_cal);
            if (lane == 0) {
                const bool valid = n_bad == 0;
                const int st = valid ? n_stat : -1;
                const int ca = valid ? n_cal : -1;
                int status = 0;
                if (valid && st == 0 && ca == 0) status |= 1;
                if (n_fly == 0) status |= 2;
                if (valid) status |= 4;
                if ((double)n_nan / (double)per < 0.1) status |= 8;
                if (n_nan == 0) status |= 16;
                flags[4 * s + 0] = st;
                flags[4 * s + 1] = ca;
                flags[4 * s + 2] = n_fly;
                flags[4 * s + 3] = status;
            }
        }
    }
}

constexpr size_t FAST_STATIC_LDS = 2 * sizeof(int) * FAST_WAVES * AFIV_MAX_MATURITIES;   // fly_all, skip_all
static_assert((sizeof(FastMat) * AFIV_MAX_MATURITIES) % 16 == 0 && sizeof(FastMat) % 8 == 0,
              "the arrays behind mat[] in dynamic LDS stay 16-byte aligned");
static_assert((ERFCX_LDS * sizeof(double)) % 16 == 0 && (AFIV_MAX_STRIKES * sizeof(double)) % 16 == 0, "LDS sub-array offsets");

size_t fast_lds_bytes(const GridView &g)
{
    const size_t per = (size_t)g.nT * g.nK;
    return sizeof(double) * ((size_t)ERFCX_LDS + AFIV_MAX_STRIKES + 4 * per) + sizeof(FastMat) * AFIV_MAX_MATURITIES +
           sizeof(float) * BACH_ROWS * BACH_ROW;
}

template <bool CHECKS>
int launch_fast(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out, int32_t *flags,
                hipStream_t s)
{
    const size_t lds = fast_lds_bytes(g);
    if (lds + FAST_STATIC_LDS > AFIV_LDS_PER_CU) return AFIV_E_SHAPE;
    static DynLdsLimit raised;
    if (const int rc = afiv_ensure_dynamic_lds(reinterpret_cast<const void *>(&iv_fast_kernel<CHECKS>), lds, raised))
        return rc;
    // A wave per surface, workgroups of eight.  Each workgroup tabulates the bracket ends of the whole
    // lattice first -- two objective evaluations per lattice point, a quarter of the work of its first
    // eight surfaces on 8 x 21 -- so a workgroup should live for four rounds of surfaces or more; on
    // the other hand surfaces differ in their share of NaN points and second evaluations, and many
    // short workgroups even that out.  Up to 512 workgroups (two per CU, all resident) take everything
    // in one go; beyond, a quarter as many workgroups as rounds, 8192 at most (sixteen rounds per
    // workgroup at a million surfaces; 2048 there: +3.6 %, and one round each on 50 000 surfaces of
    // 8 x 21: +13 %).
    int64_t blocks = (N + FAST_WAVES - 1) / FAST_WAVES;
    if (blocks > 512) blocks = std::min<int64_t>(std::max<int64_t>(blocks / 4, 512), 256 * 32);
    hipLaunchKernelGGL((iv_fast_kernel<CHECKS>), dim3((unsigned)blocks), dim3(FAST_THREADS), lds, s, g, ratio, N, spot,
                       iv_out, flags);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

}  // namespace

// variant A: implied vols and (flags != nullptr) the checks in one pass
int afiv_launch_fast_iv(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                        int32_t *flags, hipStream_t s)
{
    if (N <= 0) return 0;
    return flags ? launch_fast<true>(g, ratio, N, spot, iv_out, flags, s)
                 : launch_fast<false>(g, ratio, N, spot, iv_out, flags, s);
}
