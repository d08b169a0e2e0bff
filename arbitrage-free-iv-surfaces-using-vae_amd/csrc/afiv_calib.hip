// Calibration Stage-1 loss matrix on gfx950 (run_single_heston_calibration.py:252-287).
//
// loss[c, d] = sqrt(mean_p (model[c, p] - market[d, p])^2) + w |mean_atm model[c] - mean_atm market[d]|
//              + 5 max(0, sigma_c^2 - 2 kappa_c theta_c),   1e10 if a model value is NaN or negative,
// over the lattice points p the day's market surface holds (NaN = dropped, :185).  A "distance
// matrix" between model rows and market rows, evaluated term by term (expanding the squares into a
// matrix product would lose six digits of an RMSE that is a 1e-3 remainder of numbers of size 1).
//
// The work is 2 vector instructions per (candidate, day, point): e = m - y ; se = fma(e, e, se).
// What decides the kernel's speed is how the two operands reach the instruction:
//   * a lane owns S1_DPL days and keeps 16 consecutive lattice points of each market row in registers
//     (the tile is staged through LDS one pass ahead, so global reads are coalesced);
//   * the model value is the same for all 64 lanes: it is fetched by scalar loads (wave-uniform
//     addresses) and enters the instruction as a scalar operand -- no LDS read, no vector load;
//     every scalar fetched serves S1_DPL days per lane, which is what keeps the scalar cache's
//     miss traffic (each model value is read by one wave per day tile) under the vector work;
//   * a wave carries 8 candidates' sums of squares in registers across the passes, so every
//     market value read from LDS serves 8 candidates; a workgroup = 4 waves = 64 S1_DPL days x 32
//     candidates.
// Per-candidate quantities (any NaN / negative value, the at-the-money sum) are wave-uniform as
// long as every market point of the tile is present, and are then computed once per workgroup in
// a prologue, not per lane.  Tiles that do hold a NaN market point (noticed while staging) are
// done over with every test per lane and per point: exact, rare, slower.
// Compiled with -ffp-contract=off; the one fused operation is written out.
#include "afiv_internal.h"

namespace {

#ifndef AFIV_S1_CB
#define AFIV_S1_CB 8
#endif
#ifndef AFIV_S1_DPL
#define AFIV_S1_DPL 2
#endif
constexpr int S1_DPL = AFIV_S1_DPL;          // days per lane
constexpr int S1_DAYS = 64 * S1_DPL;         // days per workgroup: day = d0 + lane + 64 i
constexpr int S1_PTS = 16;                   // lattice points per pass (held in registers per lane and day)
constexpr int S1_CB = AFIV_S1_CB;            // candidates per wave
constexpr int S1_WAVES = 4;
constexpr int S1_THREADS = 64 * S1_WAVES;
constexpr int S1_CANDS = S1_CB * S1_WAVES;   // candidates per workgroup
constexpr int S1_MAXPTS = AFIV_MAX_STRIKES * AFIV_MAX_MATURITIES;
constexpr int S1_STAGE = S1_DAYS * S1_PTS / S1_THREADS;   // tile elements staged per thread and pass
constexpr int S1_TPC = S1_THREADS / S1_CANDS;  // prologue threads per candidate
static_assert(S1_CANDS <= 32, "one bit per candidate in a 32-bit word");
static_assert(S1_THREADS % S1_PTS == 0, "a thread stages one column of the tile");
static_assert(32 % S1_PTS == 0, "a pass's at-the-money bits sit in one word");

struct S1Shared {
    double mk[2][S1_DAYS][S1_PTS + 1];       // market tile, double-buffered; odd row stride: lanes = rows
    double atm_sum[S1_CANDS][S1_TPC];        // at-the-money sum of each candidate, interleaved partials
    unsigned atm_bits[S1_MAXPTS / 32 + 2];   // bit p: lattice point p is at the money (|log-moneyness| < 0.05)
    unsigned bad;                            // bit c: candidate c holds a NaN or negative value
    unsigned market_nan;                     // a NaN market value was staged
};

// One tile element per thread and k: column = tid mod S1_PTS, rows srow + k * (threads / S1_PTS).
__device__ __forceinline__ void s1_fetch(double (&staged)[S1_STAGE], const double *__restrict__ market_ratio,
                                         int per, int pass, int64_t d0, int64_t D)
{
    const int tid = threadIdx.x;
    const int scol = tid & (S1_PTS - 1), srow = tid / S1_PTS;
    const int p = pass * S1_PTS + scol;
#pragma unroll
    for (int k = 0; k < S1_STAGE; ++k) {
        const int64_t d = d0 + srow + k * (S1_THREADS / S1_PTS);
        // days past the end and points past the row are zeros: present, harmless, never stored
        staged[k] = (d < D && p < per) ? market_ratio[d * per + p] : 0.0;
    }
}

__device__ __forceinline__ bool s1_put(S1Shared &sh, const double (&staged)[S1_STAGE], int buf)
{
    const int tid = threadIdx.x;
    const int scol = tid & (S1_PTS - 1), srow = tid / S1_PTS;
    bool any_nan = false;
#pragma unroll
    for (int k = 0; k < S1_STAGE; ++k) {
        sh.mk[buf][srow + k * (S1_THREADS / S1_PTS)][scol] = staged[k];
        any_nan |= isnan(staged[k]);
    }
    return any_nan;
}

// A wave-uniform pointer to memory no one writes while the kernel runs (the model rows): held in
// scalar registers whatever unit computed it, and typed so that loads through it are scalar loads.
typedef const double __attribute__((address_space(4))) *s1_cptr;
__device__ __forceinline__ s1_cptr s1_uniform(const double *p)
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (s1_cptr)(((unsigned long long)hi << 32) | lo);
}

// A candidate's S1_PTS model values of one pass: wave-uniform address -> scalar loads.
struct S1Row { double v[S1_PTS]; };
__device__ __forceinline__ S1Row s1_row(s1_cptr p)
{
    S1Row r;
#pragma unroll
    for (int j = 0; j < S1_PTS; ++j) r.v[j] = p[j];
    return r;
}

// What a lane accumulates for one of its days.
struct S1Day {
    double se[S1_CB];        // sums of squared differences, one per candidate of the wave
    double sm_mkt;           // at-the-money sum of the market row
    int n, n_atm;            // points present, at-the-money points present
};

// Every market point present: sums of squares only; the per-candidate tests and the at-the-money
// sums were done in the prologue.
__device__ __forceinline__ void s1_walk_plain(S1Shared &sh, const double *__restrict__ model_ratio,
                                              const double *__restrict__ market_ratio, int per, int npass,
                                              int64_t d0, int64_t D, int64_t crow0, int64_t C,
                                              S1Day (&day)[S1_DPL])
{
    const int lane = threadIdx.x & 63;
    s1_cptr rows[S1_CB];                     // this wave's model rows (rows past the last candidate: the last)
#pragma unroll
    for (int cb = 0; cb < S1_CB; ++cb) rows[cb] = s1_uniform(model_ratio + min(crow0 + cb, C - 1) * per);
    double staged[S1_STAGE];
    s1_fetch(staged, market_ratio, per, 0, d0, D);
    bool any_nan = s1_put(sh, staged, 0);
    for (int pass = 0; pass < npass; ++pass) {
        const int p0 = pass * S1_PTS;
        const int len = min(S1_PTS, per - p0);
        __syncthreads();                     // tile `pass` is in place; the other buffer is free
        // the first candidate's values are asked for together with the tile reads
        S1Row cur = {};
        if (len == S1_PTS) cur = s1_row(rows[0] + p0);
        if (pass + 1 < npass) s1_fetch(staged, market_ratio, per, pass + 1, d0, D);
        const unsigned am = (__builtin_amdgcn_readfirstlane(sh.atm_bits[p0 >> 5]) >> (p0 & 31)) & ((1u << S1_PTS) - 1u);
        double y[S1_DPL][S1_PTS];
#pragma unroll
        for (int i = 0; i < S1_DPL; ++i) {
#pragma unroll
            for (int j = 0; j < S1_PTS; ++j) y[i][j] = sh.mk[pass & 1][lane + 64 * i][j];
            day[i].n += len;
            day[i].n_atm += __builtin_popcount(am);
        }
        if (am) {
            // y * 1 + s and y * 0 + s are exact: no select, no branch per point
#pragma unroll
            for (int j = 0; j < S1_PTS; ++j) {
                const double w = (double)((am >> j) & 1u);
#pragma unroll
                for (int i = 0; i < S1_DPL; ++i) day[i].sm_mkt = __builtin_fma(y[i][j], w, day[i].sm_mkt);
            }
        }
        if (len == S1_PTS) {
            // Scalar loads return out of order, so waiting for one waits for all that are in
            // flight: a candidate's values are waited for BEFORE the next candidate's loads are
            // issued, and those then have the 2 x S1_PTS x S1_DPL vector instructions below (times
            // the waves that share the SIMD) to arrive.
#pragma unroll
            for (int cb = 0; cb < S1_CB; ++cb) {
                asm volatile("" ::"s"(cur.v[0]));
                __builtin_amdgcn_sched_barrier(0);
                S1Row nxt = cur;
                if (cb + 1 < S1_CB) nxt = s1_row(rows[cb + 1] + p0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < S1_DPL; ++i) {
                    double acc = day[i].se[cb];
#pragma unroll
                    for (int j = 0; j < S1_PTS; ++j) {
                        const double e = cur.v[j] - y[i][j];
                        acc = __builtin_fma(e, e, acc);
                    }
                    day[i].se[cb] = acc;
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        } else {
#pragma unroll
            for (int cb = 0; cb < S1_CB; ++cb) {
                s1_cptr mrow = rows[cb] + p0;
#pragma unroll
                for (int j = 0; j < S1_PTS; ++j)
                    if (j < len) {
                        const double m = mrow[j];
#pragma unroll
                        for (int i = 0; i < S1_DPL; ++i) {
                            const double e = m - y[i][j];
                            day[i].se[cb] = __builtin_fma(e, e, day[i].se[cb]);
                        }
                    }
            }
        }
        if (pass + 1 < npass) any_nan |= s1_put(sh, staged, (pass + 1) & 1);
    }
    if (any_nan) sh.market_nan = 1;          // benign race: every writer stores 1
}

// Tiles holding a NaN market point: every test per lane and per point.
__device__ __forceinline__ void s1_walk_masked(S1Shared &sh, const double *__restrict__ model_ratio,
                                               const double *__restrict__ market_ratio, int per, int npass,
                                               int64_t d0, int64_t D, int64_t crow0, int64_t C,
                                               S1Day (&day)[S1_DPL], double (&smm)[S1_DPL][S1_CB],
                                               unsigned (&lbad)[S1_DPL])
{
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass < npass; ++pass) {
        double staged[S1_STAGE];
        s1_fetch(staged, market_ratio, per, pass, d0, D);
        __syncthreads();                     // the tile of the previous pass has been read
        s1_put(sh, staged, 0);
        __syncthreads();
        const int p0 = pass * S1_PTS;
        const int len = min(S1_PTS, per - p0);
        const unsigned am = (__builtin_amdgcn_readfirstlane(sh.atm_bits[p0 >> 5]) >> (p0 & 31)) & ((1u << S1_PTS) - 1u);
#pragma unroll 1
        for (int j = 0; j < len; ++j) {
            const bool a = (am >> j) & 1u;
            double m[S1_CB];
#pragma unroll
            for (int cb = 0; cb < S1_CB; ++cb) m[cb] = model_ratio[min(crow0 + cb, C - 1) * per + p0 + j];
#pragma unroll
            for (int i = 0; i < S1_DPL; ++i) {
                const double y = sh.mk[0][lane + 64 * i][j];
                if (isnan(y)) continue;                           // a point the reference drops (:185)
                day[i].n++;
                if (a) { day[i].sm_mkt += y; day[i].n_atm++; }
#pragma unroll
                for (int cb = 0; cb < S1_CB; ++cb) {
                    if (isnan(m[cb]) || m[cb] < 0) lbad[i] |= 1u << cb;
                    const double e = m[cb] - y;
                    day[i].se[cb] = __builtin_fma(e, e, day[i].se[cb]);
                    if (a) smm[i][cb] += m[cb];
                }
            }
        }
    }
}

__device__ __forceinline__ void s1_clear(S1Day (&day)[S1_DPL])
{
#pragma unroll
    for (int i = 0; i < S1_DPL; ++i) {
#pragma unroll
        for (int q = 0; q < S1_CB; ++q) day[i].se[q] = 0.0;
        day[i].sm_mkt = 0.0;
        day[i].n = 0;
        day[i].n_atm = 0;
    }
}

__global__ __launch_bounds__(S1_THREADS) void stage1_kernel(GridView g, const double *__restrict__ model_ratio,
                                                            const double *__restrict__ params, int64_t C,
                                                            const double *__restrict__ market_ratio, int64_t D,
                                                            double drift_weight, double *__restrict__ loss_out)
{
    __shared__ S1Shared sh;
    const int nK = g.nK, per = g.nT * nK;
    const int npass = (per + S1_PTS - 1) / S1_PTS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t d0 = (int64_t)blockIdx.x * S1_DAYS, c0 = (int64_t)blockIdx.y * S1_CANDS;
    const int64_t crow0 = c0 + wave * S1_CB;                      // this wave's first candidate
    const double nan = __builtin_nan("");

    if (tid == 0) { sh.bad = 0; sh.market_nan = 0; }
    for (int p = tid; p < (per + 63) / 64 * 64; p += S1_THREADS) {
        const bool a = p < per && fabs(g.logm[p % nK]) < 0.05;
        const unsigned long long b = __ballot(a);
        if (lane == 0) {
            sh.atm_bits[(p >> 5) + 0] = (unsigned)b;
            sh.atm_bits[(p >> 5) + 1] = (unsigned)(b >> 32);
        }
    }
    __syncthreads();
    {
        // per candidate: any NaN / negative value, and the at-the-money sum (S1_TPC threads per
        // candidate, each over every S1_TPC-th point; partials combined in a fixed order below)
        const int cc = tid / S1_TPC, sub = tid % S1_TPC;
        double part = 0.0;
        bool bad = false;
        if (c0 + cc < C) {
            const double *__restrict__ row = model_ratio + (c0 + cc) * per;
#pragma unroll 8
            for (int p = sub; p < per; p += S1_TPC) {
                const double m = row[p];
                if (isnan(m) || m < 0) bad = true;
                if ((sh.atm_bits[p >> 5] >> (p & 31)) & 1u) part += m;
            }
        }
        sh.atm_sum[cc][sub] = part;
        if (bad) atomicOr(&sh.bad, 1u << cc);
    }
    // (the first barrier inside s1_walk_plain orders the prologue's LDS writes before their readers)

    S1Day day[S1_DPL];
    double smm[S1_DPL][S1_CB];
    unsigned lbad[S1_DPL];
#pragma unroll
    for (int i = 0; i < S1_DPL; ++i) {
        lbad[i] = 0;
#pragma unroll
        for (int q = 0; q < S1_CB; ++q) smm[i][q] = 0.0;
    }
    s1_clear(day);
    s1_walk_plain(sh, model_ratio, market_ratio, per, npass, d0, D, crow0, C, day);
    __syncthreads();                                              // every wave is through with the tiles
    if (sh.market_nan != 0) {                                     // uniform over the workgroup
        s1_clear(day);
        s1_walk_masked(sh, model_ratio, market_ratio, per, npass, d0, D, crow0, C, day, smm, lbad);
    }
    // the prologue's per-candidate quantities: what a day with every point present takes on either
    // path, so that its losses do not depend on what else its tile holds
    const unsigned ubad = sh.bad >> (wave * S1_CB);
    double usmm[S1_CB];
#pragma unroll
    for (int q = 0; q < S1_CB; ++q) {
        const double *ps = sh.atm_sum[wave * S1_CB + q];
        double t = ps[0];
#pragma unroll
        for (int i = 1; i < S1_TPC; ++i) t += ps[i];
        usmm[q] = t;
    }
#pragma unroll
    for (int i = 0; i < S1_DPL; ++i) {
        const int64_t d = d0 + lane + 64 * i;
        if (d >= D) continue;
        const bool full = day[i].n == per;
        const int n = day[i].n, n_atm = day[i].n_atm;
#pragma unroll
        for (int q = 0; q < S1_CB; ++q) {
            const int64_t c = crow0 + q;
            if (c >= C) continue;
            const double kappa = params[5 * c + 0], theta = params[5 * c + 1], sigma = params[5 * c + 2];
            const double feller = 5.0 * fmax(0.0, sigma * sigma - 2 * kappa * theta);
            const double sm_model = full ? usmm[q] : smm[i][q];
            const bool bad = ((full ? ubad : lbad[i]) >> q) & 1u;
            double loss;
            if (n < 10) loss = nan;                               // the reference skips such a day (:202-203)
            else if (bad) loss = 1e10;
            else {
                const double rmse = sqrt(day[i].se[q] / n);
                const double drift = n_atm > 0 ? fabs(sm_model / n_atm - day[i].sm_mkt / n_atm) : 0.0;
                loss = rmse + drift_weight * drift + feller;
            }
            loss_out[c * D + d] = loss;
        }
    }
}

}  // namespace

int afiv_launch_stage1(const GridView &g, const double *model_ratio, const double *params, int64_t C,
                       const double *market_ratio, int64_t D, double drift_weight, double *loss_out,
                       hipStream_t s)
{
    if (C <= 0 || D <= 0) return 0;
    const int64_t by = (C + S1_CANDS - 1) / S1_CANDS, bx = (D + S1_DAYS - 1) / S1_DAYS;
    if (by > 65535 || bx > 0x7fffffff) return AFIV_E_SHAPE;
    hipLaunchKernelGGL(stage1_kernel, dim3((unsigned)bx, (unsigned)by), dim3(S1_THREADS), 0, s, g, model_ratio,
                       params, C, market_ratio, D, drift_weight, loss_out);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}
