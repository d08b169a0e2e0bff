// Implied-volatility inversion and arbitrage checks on gfx950.
//
// Compiled with -ffp-contract=off: these kernels reproduce NumPy/SciPy evaluation order so that
// guards, bracket signs and violation counts agree with the Python reference bit for bit on
// identical inputs (mul and add are never fused here).
//
//   implied vol : generate_iv_surface_by_date.py:296-318 (variant A) and
//                 generate_regime_surfaces.py:138-177 (variant B); the root finder makes
//                 scipy.optimize.brentq's decisions one for one (one grid point per lane at a time;
//                 a step's arithmetic differs from brentq.c's in the last bits), because variant
//                 B's objective is not monotone and only the same iteration path lands on the
//                 same root.
//   checks      : generate_regime_surfaces.py:188-283 and cvae_model.py:272-285.
#include "afiv_bs.h"

namespace {
using namespace afivbs;

// One workgroup per surface.
constexpr int ARB_THREADS = 256;
#ifndef AFIV_IV_REFILL_LANES
#define AFIV_IV_REFILL_LANES 8            // idle lanes of a wave that trigger a refill
#endif
constexpr int MAX_PTS = AFIV_MAX_STRIKES * AFIV_MAX_MATURITIES;

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
// every workgroup tabulates them in LDS before its first surface (two evaluations per lattice point,
// against ~10 per point and surface in the search), which saves two of the ~12.7 evaluations per
// point and keeps the call free of any state outside its arguments.
// A wave therefore spends sum(iterations)/64 trips on a surface plus a short tail, 1.7 x fewer
// evaluations than lanes bound to fixed points on the benchmark lattice.
//
// The surface's price ratios are staged in LDS (coalesced read); the butterfly check runs on them
// at once, and each entry is then overwritten by the Black-Scholes price of the solved vol,
// taken from Brent's last objective value (objective + target, exact at a root), so the static
// and calendar checks cost no further normal CDFs.  Vols go straight to global memory.
struct MatRow { double tau, sqrt_tau, disc, F, sdq, fwd; };
static_assert(sizeof(MatRow) % 16 == 0 && (ERFCX_LDS * sizeof(double)) % 16 == 0,
              "the arrays behind the maturity rows in dynamic LDS are read as 16-byte pairs");

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

enum { PH_IDLE = 0, PH_ITER = 1, PH_DONE = 2 };

// FKTAB: the forward form's log(F/K) and F/K per lattice point are tabulated in LDS too (false: the
// largest lattices, where that table does not fit, compute them as points are handed out).
template <int VARIANT, bool CHECKS, int GT, int THREADS, bool FKTAB = true>
__global__ __launch_bounds__(THREADS) void surface_kernel(GridView g, const double *__restrict__ ratio,
                                                          int64_t N, double spot, double *__restrict__ iv_out,
                                                          int32_t *__restrict__ flags)
{
    constexpr int NG = THREADS / GT;
    extern __shared__ __align__(16) double dyn[];
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    double *tab = dyn;                                   // ERFCX_LDS
    MatRow *mat = reinterpret_cast<MatRow *>(tab + ERFCX_LDS);                // [64]
    double *strike = reinterpret_cast<double *>(mat + AFIV_MAX_MATURITIES);   // [64]
    double *logm = strike + AFIV_MAX_STRIKES;            // [64]
    double *kmult = logm + AFIV_MAX_STRIKES;             // [64]
    d2v *ends = reinterpret_cast<d2v *>(kmult + AFIV_MAX_STRIKES);            // [per] objective at the bracket ends
    d2v *fk = ends + per;                                // [per] log(F/K), F/K (forward form only)
    double *price_all = reinterpret_cast<double *>(fk + (VARIANT == AFIV_VARIANT_A && FKTAB ? per : 0));   // [NG][per]
    unsigned short *order = reinterpret_cast<unsigned short *>(price_all + (size_t)NG * per);   // [per] queue position -> lattice point
    __shared__ int skip_all[NG][AFIV_MAX_MATURITIES];
    __shared__ int ctr_all[NG][8];                       // queue head, nan, bad, static, calendar, butterfly
    const int tid = threadIdx.x, lane = tid & 63, gid = tid / GT, lt = tid % GT;
    double *price = price_all + (size_t)gid * per;
    int *skip_t = skip_all[gid];
    int *ctr = ctr_all[gid];
    load_erfcx_table(tab);
    for (int t = tid; t < nT; t += THREADS) {
        MatRow m;
        m.tau = g.tau[t];
        m.sqrt_tau = sqrt(m.tau);
        m.disc = g.disc_tau[t];
        m.F = spot * g.fwd_mult[t];
        m.sdq = spot * g.dq_tau[t];
        m.fwd = g.fwd_mult[t];
        mat[t] = m;
    }
    for (int k = tid; k < nK; k += THREADS) {
        strike[k] = spot * g.Kmult[k];
        logm[k] = g.logm[k];
        kmult[k] = g.Kmult[k];
    }
    for (int i = tid; i < per; i += THREADS) order[i] = g.order[i];
    __syncthreads();

    const float inv_nK = 1.0f / (float)nK;
    const double nan = __builtin_nan("");
    const double rq = g.r - g.q;
    const afivm::ExpCoef ec = afivm::load_exp_coef();
    // Black-Scholes values at the two bracket ends for every lattice point
    for (int i = tid; i < per; i += THREADS) {
        const int t = small_div(i, inv_nK), k = i - __mul24(t, nK);
        const BsPoint c = make_point(g, t, k, spot, VARIANT);
        d2v e;
        e.x = bs_objective<VARIANT>(c, BRENT_XA, tab, ec);
        e.y = bs_objective<VARIANT>(c, BRENT_XB, tab, ec);
        ends[i] = e;
        if (VARIANT == AFIV_VARIANT_A && FKTAB) {
            d2v f;
            f.x = c.lnFK;
            f.y = c.ratio;
            fk[i] = f;
        }
    }
    __syncthreads();
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
                        const d2v pt = ends[i_cur];
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
                        // the per-point constants of make_point, from the staged rows
                        if (VARIANT == AFIV_VARIANT_A && FKTAB) {
                            const d2v pf = fk[i_cur];
                            c.lnFK = pf.x;
                            c.ratio = pf.y;
                        } else if (VARIANT == AFIV_VARIANT_A) {
                            c.ratio = c.F / c.K;
                            c.lnFK = log(c.ratio);
                        } else {
                            c.lnFK = 0.0;
                            c.ratio = kmult[k] * m.fwd;
                        }
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
                        const double fa = pt.x - hp, fb = pt.y - hp;
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
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

namespace {
size_t surface_lds_bytes(const GridView &g, int groups, int variant, bool fktab = true)
{
    // erfcx table, maturity rows, three strike arrays, the per-point tables (bracket ends; the forward
    // form also log(F/K) and F/K), the staged surfaces, the service order; the kernel's static arrays
    // (skip flags and counters per group) come on top
    const size_t per = (size_t)g.nT * g.nK;
    const size_t tables = variant == AFIV_VARIANT_A && fktab ? 4 * per : 2 * per;
    return sizeof(double) * ((size_t)ERFCX_LDS + 6 * AFIV_MAX_MATURITIES + 3 * AFIV_MAX_STRIKES + tables +
                             (size_t)groups * per) +
           (per * sizeof(unsigned short) + 7) / 8 * 8;
}

size_t surface_static_lds(int groups) { return (size_t)groups * (AFIV_MAX_MATURITIES + 8) * sizeof(int); }

constexpr size_t LDS_PER_CU = AFIV_LDS_PER_CU;

template <int VARIANT, bool CHECKS, int GT, int THREADS, bool FKTAB = true>
int launch_surface_kernel(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                          int32_t *flags, hipStream_t s)
{
    constexpr int NG = THREADS / GT;
    const size_t lds = surface_lds_bytes(g, NG, VARIANT, FKTAB);
    if (lds + surface_static_lds(NG) > LDS_PER_CU) return AFIV_E_SHAPE;
    static DynLdsLimit raised;
    if (const int rc = afiv_ensure_dynamic_lds(
            reinterpret_cast<const void *>(&surface_kernel<VARIANT, CHECKS, GT, THREADS, FKTAB>), lds, raised))
        return rc;
    // few, long-lived workgroups: each one tabulates the bracket ends of the whole lattice first
    int64_t blocks = (N + NG - 1) / NG;
    const int64_t cap = GT == 64 ? 256 * 4 : 256 * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((surface_kernel<VARIANT, CHECKS, GT, THREADS, FKTAB>), dim3((unsigned)blocks), dim3(THREADS), lds,
                       s, g, ratio, N, spot, iv_out, flags);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

template <int VARIANT, bool CHECKS>
int launch_surface_grouped(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                           int32_t *flags, hipStream_t s)
{
    // A wave per surface needs surfaces enough to fill the chip.  Sixteen waves per CU where the
    // lattice's LDS footprint allows it: two workgroups of eight surfaces, else one of sixteen, else
    // two of four; a workgroup per surface otherwise.
    if (N >= 2048) {
        if (2 * (surface_lds_bytes(g, 8, VARIANT) + surface_static_lds(8)) <= LDS_PER_CU)
            return launch_surface_kernel<VARIANT, CHECKS, 64, 512>(g, ratio, N, spot, iv_out, flags, s);
        if (surface_lds_bytes(g, 16, VARIANT) + surface_static_lds(16) <= LDS_PER_CU)
            return launch_surface_kernel<VARIANT, CHECKS, 64, 1024>(g, ratio, N, spot, iv_out, flags, s);
        if (2 * (surface_lds_bytes(g, 4, VARIANT) + surface_static_lds(4)) <= LDS_PER_CU)
            return launch_surface_kernel<VARIANT, CHECKS, 64, 256>(g, ratio, N, spot, iv_out, flags, s);
    }
    if (VARIANT == AFIV_VARIANT_A && surface_lds_bytes(g, 1, VARIANT) + surface_static_lds(1) > LDS_PER_CU)
        return launch_surface_kernel<VARIANT, CHECKS, 256, 256, false>(g, ratio, N, spot, iv_out, flags, s);
    return launch_surface_kernel<VARIANT, CHECKS, 256, 256>(g, ratio, N, spot, iv_out, flags, s);
}
}  // namespace

// implied vols and (flags != nullptr) the checks in one pass; bs form of the checks = variant
int afiv_launch_surface_iv(const GridView &g, const double *ratio, int64_t N, double spot, int variant,
                           double *iv_out, int32_t *flags, hipStream_t s)
{
    if (N <= 0) return 0;
    if (variant == AFIV_VARIANT_A) return afiv_launch_fast_iv(g, ratio, N, spot, iv_out, flags, s);
    if (variant == AFIV_VARIANT_A_BRENT)
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
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

int afiv_launch_penalty(const GridView &g, const double *ratio, int64_t B, double *acc_out, hipStream_t s)
{
    hipLaunchKernelGGL(penalty_kernel, dim3(1), dim3(256), 0, s, g, ratio, B, acc_out);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}
