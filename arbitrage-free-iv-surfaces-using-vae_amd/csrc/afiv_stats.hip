// Ensemble statistics over the sample axis of a stack of surfaces on gfx950:
// generate_iv_surface_by_date.py:474-478, i.e. np.nanmean / np.nanmedian /
// np.nanpercentile(5) / np.nanpercentile(95) along axis 0.
//
// One workgroup per grid point.  The point's column (stride nT*nK doubles) is gathered into LDS in
// sample order with NaNs and unselected surfaces dropped; the mean is the sequential sum in that
// order divided by the count, which is how NumPy reduces over a leading axis (it adds whole rows
// one after another), so the mean is bit-identical to np.nanmean.  The column is then sorted in
// LDS (bitonic network, padded with +inf) and the order statistics are interpolated exactly as
// NumPy's "linear" method does, including its virtual-index arithmetic and the two-sided lerp.
// HBM-bound by construction (each 8-byte element sits in its own 64-byte line for the gather), but
// tiny: N x nT x nK x 8 bytes read once.
#include "afiv_internal.h"

namespace {

constexpr int STAT_THREADS = 256;
constexpr int MAX_POINTS = AFIV_MAX_STRIKES * AFIV_MAX_MATURITIES;

// numpy/lib/_function_base_impl.py, method "linear": virtual index (n - 1) q, then _get_indexes,
// _get_gamma and _lerp for one quantile q of m sorted values v.
__device__ double numpy_linear_quantile(const double *v, int m, double q)
{
    const double virt = (double)(m - 1) * q;
    double prev = floor(virt);
    int ip, in;
    if (virt >= (double)(m - 1)) { ip = in = m - 1; prev = -1.0; }
    else if (virt < 0.0) { ip = in = 0; prev = 0.0; }
    else { ip = (int)prev; in = ip + 1; }
    const double gamma = virt - prev;
    const double a = v[ip], b = v[in];
    const double diff = b - a;
    double r = a + diff * gamma;
    if (gamma >= 0.5) r = b - diff * (1.0 - gamma);
    return r;
}

// `segments` consecutive groups of N surfaces each get their own statistics (out [segments][4][per]):
// one launch for, say, the 100 samples of each of 500 dates.
__global__ __launch_bounds__(STAT_THREADS) void ensemble_kernel(const double *__restrict__ iv_all,
                                                                const uint8_t *__restrict__ select_all, int64_t N,
                                                                int per, int padded, int64_t segments,
                                                                double *__restrict__ out_all)
{
    extern __shared__ double vals[];              // [padded]
    __shared__ int wave_count[STAT_THREADS / 64];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double inf = __builtin_inf();
    for (int64_t item = blockIdx.x; item < segments * per; item += gridDim.x) {
        const int64_t seg = item / per;
        const int p = (int)(item - seg * per);
        const double *__restrict__ iv = iv_all + seg * N * per;
        const uint8_t *__restrict__ select = select_all ? select_all + seg * N : nullptr;
        double *__restrict__ out = out_all + seg * 4 * per;
        __syncthreads();
        if (tid == 0) total = 0;
        __syncthreads();
        // stable compaction, 256 samples per pass
        for (int64_t base = 0; base < N; base += STAT_THREADS) {
            const int64_t i = base + tid;
            double x = 0.0;
            bool keep = false;
            if (i < N && (!select || select[i])) {
                x = iv[i * per + p];
                keep = !(x != x);
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) wave_count[wave] = __popcll(m);
            __syncthreads();
            int off = total;
            for (int w = 0; w < wave; ++w) off += wave_count[w];
            off += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (keep) vals[off] = x;
            __syncthreads();
            if (tid == 0) {
                int t = total;
                for (int w = 0; w < STAT_THREADS / 64; ++w) t += wave_count[w];
                total = t;
            }
            __syncthreads();
        }
        const int m = total;
        if (tid == 0) {
            double sum = 0.0;
            for (int i = 0; i < m; ++i) sum += vals[i];
            out[0 * per + p] = m > 0 ? sum / (double)m : __builtin_nan("");
        }
        __syncthreads();
        for (int i = m + tid; i < padded; i += STAT_THREADS) vals[i] = inf;
        __syncthreads();
        for (int k = 2; k <= padded; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < padded; i += STAT_THREADS) {
                    const int l = i ^ j;
                    if (l > i) {
                        const double a = vals[i], b = vals[l];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { vals[i] = b; vals[l] = a; }
                    }
                }
                __syncthreads();
            }
        }
        if (tid == 0) {
            double med, p5, p95;
            if (m == 0) {
                med = p5 = p95 = __builtin_nan("");
            } else {
                // np.median: the middle value, or the mean of the two middle values
                med = (m & 1) ? vals[m / 2] : (vals[m / 2 - 1] + vals[m / 2]) / 2.0;
                p5 = numpy_linear_quantile(vals, m, 5.0 / 100.0);
                p95 = numpy_linear_quantile(vals, m, 95.0 / 100.0);
            }
            out[1 * per + p] = med;
            out[2 * per + p] = p5;
            out[3 * per + p] = p95;
        }
    }
}

// ---- groups of at most 128 surfaces (the reference's 100 samples per date) -------------------------
// One WAVE per (group, grid point): a workgroup of 256 threads spends most of such an item in
// barriers -- a compaction pass, the one-thread sum and 28 sorting stages of 64 exchanges each.
// A wave holds the column in two registers per lane, compacts it into its own 1 KB of LDS in sample
// order (ballot / prefix count), adds it up in that order on one lane, sorts the 128 padded slots with
// the same bitonic network -- one exchange per lane and stage, wave-level ordering instead of
// barriers -- and interpolates as above.  Same arithmetic, same results as ensemble_kernel bit for bit.
constexpr int SMALL_N = 128;
constexpr int SMALL_WAVES = STAT_THREADS / 64;

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS operations of one wave are executed in order; this only pins the compiler's order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(STAT_THREADS) void ensemble_small_kernel(const double *__restrict__ iv_all,
                                                                      const uint8_t *__restrict__ select_all, int N,
                                                                      int per, int64_t segments,
                                                                      double *__restrict__ out_all)
{
    __shared__ double cols[SMALL_WAVES][SMALL_N];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *vals = cols[wave];
    const double inf = __builtin_inf();
    const int64_t items = segments * per;
    for (int64_t item = (int64_t)blockIdx.x * SMALL_WAVES + wave; item < items; item += (int64_t)gridDim.x * SMALL_WAVES) {
        const int64_t seg = item / per;
        const int p = (int)(item - seg * per);
        const double *__restrict__ iv = iv_all + seg * N * per;
        const uint8_t *__restrict__ select = select_all ? select_all + seg * N : nullptr;
        double *__restrict__ out = out_all + seg * 4 * per;
        // rows lane and lane + 64
        const int i0 = lane, i1 = lane + 64;
        double x0 = 0.0, x1 = 0.0;
        bool k0 = false, k1 = false;
        if (i0 < N && (!select || select[i0])) {
            x0 = iv[(int64_t)i0 * per + p];
            k0 = x0 == x0;
        }
        if (i1 < N && (!select || select[i1])) {
            x1 = iv[(int64_t)i1 * per + p];
            k1 = x1 == x1;
        }
        const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        const int n0 = __popcll(m0), m = n0 + __popcll(m1);
        const int o0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
        const int o1 = n0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
        wave_lds_sync();                       // the previous item's readers are done with the column
        if (k0) vals[o0] = x0;
        if (k1) vals[o1] = x1;
        if (lane >= m) vals[lane] = inf;       // padding of the sort: slots m .. 127
        if (lane + 64 >= m) vals[lane + 64] = inf;
        wave_lds_sync();
        // the sum in sample order (one chain, its operands read eight ahead; the padding is +inf and
        // stays out through the count)
        double sum = 0.0;
        for (int r0 = 0; r0 < m; r0 += 8) {
            double x[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = vals[(r0 + r) & (SMALL_N - 1)];
#pragma unroll
            for (int r = 0; r < 8; ++r) sum = r0 + r < m ? sum + x[r] : sum;
        }
        // bitonic network over 128 slots: exchange (i, i ^ j) with i the lower index, one per lane
        for (int k = 2; k <= SMALL_N; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int i = ((lane & ~(j - 1)) << 1) | (lane & (j - 1));      // lane-th index with bit j clear
                const int l = i | j;
                const double a = vals[i], b = vals[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) { vals[i] = b; vals[l] = a; }      // (the stage's pairs are disjoint)
                wave_lds_sync();
            }
        }
        if (lane == 0) {
            double mean, med, p5, p95;
            if (m == 0) {
                mean = med = p5 = p95 = __builtin_nan("");
            } else {
                mean = sum / (double)m;
                med = (m & 1) ? vals[m / 2] : (vals[m / 2 - 1] + vals[m / 2]) / 2.0;
                p5 = numpy_linear_quantile(vals, m, 5.0 / 100.0);
                p95 = numpy_linear_quantile(vals, m, 95.0 / 100.0);
            }
            out[0 * per + p] = mean;
            out[1 * per + p] = med;
            out[2 * per + p] = p5;
            out[3 * per + p] = p95;
        }
    }
}

// ---- ensembles larger than one LDS-resident column ---------------------------------------------
// The same four statistics, bit for bit, for any number of surfaces, without scratch memory.  One
// workgroup owns TILE_P adjacent grid points (half a 64-byte line of every surface row; the workgroup
// next door reads the other half at about the same time, out of L2) and streams
// the stack several times:
//   sweep 1   count of usable values, their sum in sample order (the order NumPy adds the rows of a
//             leading axis in, so the mean is np.nanmean's), and the smallest / largest value;
//   sweep 2.. order statistics by narrowing: the median and the two percentiles need six ranks of the
//             sorted column.  Values map to order-preserving 64-bit keys; every rank keeps a key
//             interval [lo, lo + 2^shift) known to contain it plus the number of values below lo,
//             starting from the common prefix of the smallest and largest key.  A sweep builds a
//             256-bin histogram of each distinct interval (LDS atomics) and every rank moves into
//             the bin that holds it -- eight key bits per sweep -- until its interval holds at most
//             SEL_CAP values (or a single key);
//   last      the values of the remaining intervals are collected in LDS and each rank is picked by
//             counting: value v is the k-th smallest iff #(values < v) <= k < #(values <= v).
// Selection is exact, so the interpolated statistics equal NumPy's whatever its partition does.
// Three sweeps for 50 000 surfaces, four for a million; each reads every value once, coalesced.
constexpr int SEL_THREADS = 256;
constexpr int SEL_BINS = 256;
constexpr int SEL_TARGETS = 6;
constexpr int SEL_CAP = 256;
constexpr int SEL_UNROLL = 16;                     // loads in flight per lane in the narrowing sweeps

__device__ __forceinline__ unsigned long long order_key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// The values of U consecutive surfaces i0 .. i0 + U - 1 at grid point p, NaN where the row is past the
// end or not selected.  A thread takes consecutive rows so that their selection bytes are ONE 8- or
// 16-byte load where alignment and range allow (a byte load per value doubled the vector-memory
// instructions of a sweep, which is what bounds it); values and bytes are loaded side by side, so
// unselected rows cost their bandwidth but no second round of latency.
template <int U>
__device__ __forceinline__ void masked_rows(const double *__restrict__ iv, const uint8_t *__restrict__ select,
                                            int64_t i0, int64_t N, int per, int p, bool have_p, double (&x)[U])
{
    static_assert(U == 8 || U == 16, "one 8- or 16-byte load of selection bytes");
    const double nan = __builtin_nan("");
    const int pc = have_p ? p : 0;
    unsigned sel[U / 4];
#pragma unroll
    for (int w = 0; w < U / 4; ++w) sel[w] = 0x01010101u;
    bool bytewise = false;
    if (select) {
        const uint8_t *sp = select + i0;
        if (i0 + U <= N && (reinterpret_cast<unsigned long long>(sp) & (U - 1)) == 0) {
            if (U == 8) {
                const uint2 v = *reinterpret_cast<const uint2 *>(sp);
                sel[0] = v.x; sel[1] = v.y;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(sp);
                sel[0] = v.x; sel[1] = v.y; sel[U / 4 - 2] = v.z; sel[U / 4 - 1] = v.w;
            }
        } else {
            bytewise = true;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + u;
        const bool in = have_p && i < N;
        const double v = iv[(in ? i : 0) * per + pc];
        bool keep = ((sel[u / 4] >> (8 * (u % 4))) & 0xffu) != 0;
        if (bytewise) keep = i < N && select[i] != 0;
        x[u] = in && keep ? v : nan;
    }
}

struct SelTarget {
    unsigned long long lo;    // interval [lo, lo + 2^shift); shift = 64: every key
    long long below;          // values with key < lo
    long long count;          // values inside the interval
    long long rank;           // 0-based rank in the sorted column
    int shift;
    int alias;                // first target of this point with the same interval (its histogram / list is used)
    int done;                 // 1: `value` is final
    double value;
};

// shift = -1 marks a slot that is switched off
__device__ __forceinline__ bool in_interval(unsigned long long key, unsigned long long lo, int shift)
{
    if (shift < 0) return false;
    return shift >= 64 ? true : ((key - lo) >> shift) == 0 && key >= lo;
}

// TILE_P = grid points per workgroup: 4 where that still makes 128 workgroups or more, else 2 or 1
// (the 168 points of the reference's lattice: 168 workgroups instead of 42; the eight workgroups
// that share a 64-byte line then each fetch it, out of the last-level cache for all but the first).
template <int TILE_P>
__global__ __launch_bounds__(SEL_THREADS) void ensemble_select_kernel(const double *__restrict__ iv,
                                                                      const uint8_t *__restrict__ select, int64_t N,
                                                                      int per, double *__restrict__ out)
{
    // histograms [TILE_P][SEL_TARGETS][SEL_BINS] u32 (24 KB), later candidate lists
    // [TILE_P][SEL_TARGETS][SEL_CAP] doubles (48 KB); the in-order sum's staging block shares it
    // (the staging block is 8 x 256 doubles whatever TILE_P is)
    constexpr int POOL_DOUBLES = TILE_P * SEL_TARGETS * SEL_CAP > 8 * SEL_THREADS ? TILE_P * SEL_TARGETS * SEL_CAP : 8 * SEL_THREADS;
    __shared__ __attribute__((aligned(16))) unsigned char pool[POOL_DOUBLES * sizeof(double)];
    __shared__ SelTarget tg[TILE_P][SEL_TARGETS];
    __shared__ unsigned long long kmin[TILE_P], kmax[TILE_P];
    __shared__ long long cnt[TILE_P];
    __shared__ double sum[TILE_P];
    __shared__ int list_n[TILE_P][SEL_TARGETS];
    __shared__ int again;
    constexpr int SEL_ROWS = SEL_THREADS / TILE_P;     // rows per sweep step
    constexpr int SEL_BLOCK = 8 * SEL_ROWS;            // rows staged per block of the in-order sum: eight per thread
    static_assert(SEL_BLOCK * TILE_P * sizeof(double) <= sizeof(pool), "the staging block fits the pool");
    unsigned *hist = reinterpret_cast<unsigned *>(pool);
    double *cand = reinterpret_cast<double *>(pool);
    double *stage = reinterpret_cast<double *>(pool);          // [SEL_BLOCK][TILE_P], NaN = skip
    const int tid = threadIdx.x, pt = tid & (TILE_P - 1), rl = tid / TILE_P;
    const double nan = __builtin_nan("");

    for (int p0 = blockIdx.x * TILE_P; p0 < per; p0 += gridDim.x * TILE_P) {
        const int p = p0 + pt;
        const bool have_p = p < per;
        __syncthreads();
        if (tid < TILE_P) { kmin[tid] = ~0ull; kmax[tid] = 0ull; cnt[tid] = 0; sum[tid] = 0.0; }
        __syncthreads();

        // ---- sweep 1: count, in-order sum, extremes
        unsigned long long mn = ~0ull, mx = 0ull;
        long long c = 0;
        for (int64_t base = 0; base < N; base += SEL_BLOCK) {
            double xb[8];
            masked_rows<8>(iv, select, base + rl * 8, N, per, p, have_p, xb);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double x = xb[u];
                stage[(rl * 8 + u) * TILE_P + pt] = x;
                if (x == x) {
                    const unsigned long long k = order_key(x);
                    mn = k < mn ? k : mn;
                    mx = k > mx ? k : mx;
                    ++c;
                }
            }
            __syncthreads();
            if (tid < TILE_P) {
                // the in-order sum: one dependent chain per point, its operands read eight ahead
                // (rows past the end were staged as NaN and drop out like every other NaN)
                double sacc = sum[tid];
                for (int r0 = 0; r0 < SEL_BLOCK; r0 += 8) {
                    double x[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] = stage[(r0 + r) * TILE_P + tid];
#pragma unroll
                    for (int r = 0; r < 8; ++r) sacc = x[r] == x[r] ? sacc + x[r] : sacc;
                }
                sum[tid] = sacc;
            }
            __syncthreads();
        }
        if (c) {
            atomicMin(&kmin[pt], mn);
            atomicMax(&kmax[pt], mx);
            atomicAdd(reinterpret_cast<unsigned long long *>(&cnt[pt]), (unsigned long long)c);
        }
        __syncthreads();

        // ---- ranks wanted: median (one or two middle values), 5th and 95th percentile neighbours
        if (tid < TILE_P) {
            const long long m = cnt[tid];
            long long ranks[SEL_TARGETS] = {0, 0, 0, 0, 0, 0};
            if (m > 0) {
                ranks[0] = (m & 1) ? m / 2 : m / 2 - 1;
                ranks[1] = m / 2;
                for (int q = 0; q < 2; ++q) {
                    // numpy "linear": virtual index (m - 1) q, neighbours floor and floor + 1 (clamped)
                    const double virt = (double)(m - 1) * (q == 0 ? 5.0 / 100.0 : 95.0 / 100.0);
                    long long ip, in;
                    if (virt >= (double)(m - 1)) ip = in = m - 1;
                    else if (virt < 0.0) ip = in = 0;
                    else { ip = (long long)floor(virt); in = ip + 1; }
                    ranks[2 + 2 * q] = ip;
                    ranks[3 + 2 * q] = in;
                }
            }
            const unsigned long long diff = kmin[tid] ^ kmax[tid];
            const int sh = m > 0 ? (diff ? 64 - __clzll((long long)diff) : 0) : 0;
            for (int j = 0; j < SEL_TARGETS; ++j) {
                SelTarget t;
                t.shift = sh;
                t.lo = sh >= 64 ? 0ull : (kmin[tid] >> sh) << sh;
                t.below = 0;
                t.count = m;
                t.rank = ranks[j];
                t.alias = 0;
                t.done = m == 0 || sh == 0;
                t.value = m == 0 ? nan : key_value(kmin[tid]);
                tg[tid][j] = t;
            }
        }
        __syncthreads();

        // ---- narrowing sweeps
        for (;;) {
            if (tid == 0) again = 0;
            for (int i = tid; i < TILE_P * SEL_TARGETS * SEL_BINS; i += SEL_THREADS) hist[i] = 0u;
            __syncthreads();
            if (tid < TILE_P * SEL_TARGETS) {
                const int q = tid / SEL_TARGETS, j = tid - q * SEL_TARGETS;
                if (!tg[q][j].done && tg[q][j].count > SEL_CAP) again = 1;
            }
            __syncthreads();
            if (!again) break;
            // this point's intervals still to be narrowed (aliases removed), in registers
            // (slot a = target a itself, switched off where it is done, small enough or an alias: static
            // indices keep the three arrays in registers)
            unsigned long long lo[SEL_TARGETS];
            int sh[SEL_TARGETS], slot[SEL_TARGETS];
            constexpr int nsel = SEL_TARGETS;
#pragma unroll
            for (int j = 0; j < SEL_TARGETS; ++j) {
                const SelTarget &t = tg[pt][j];
                const bool on = !t.done && t.count > SEL_CAP && t.alias == j;
                lo[j] = on ? t.lo : ~0ull;                 // an empty interval: lo = 2^64 - 1, shift = 0 holds one key
                sh[j] = on ? t.shift : -1;                 // ... and shift = -1 none at all (see in_interval)
                slot[j] = j;
            }
            // eight independent loads in flight per lane: a sweep is bound by memory latency otherwise
            for (int64_t base = 0; base < N; base += SEL_ROWS * SEL_UNROLL) {
                double xs[SEL_UNROLL];
                masked_rows<SEL_UNROLL>(iv, select, base + (int64_t)rl * SEL_UNROLL, N, per, p, have_p, xs);
#pragma unroll
                for (int u = 0; u < SEL_UNROLL; ++u) {
                    const double x = xs[u];
                    if (x == x) {
                        const unsigned long long k = order_key(x);
#pragma unroll
                        for (int a = 0; a < SEL_TARGETS; ++a)        // (fully unrolled: the arrays stay in registers)
                            if (a < nsel && in_interval(k, lo[a], sh[a])) {
                                const int down = sh[a] > 8 ? sh[a] - 8 : 0;
                                const unsigned bin = (unsigned)((k - lo[a]) >> down) & (SEL_BINS - 1);
                                atomicAdd(&hist[(pt * SEL_TARGETS + slot[a]) * SEL_BINS + bin], 1u);
                            }
                    }
                }
            }
            __syncthreads();
            if (tid < TILE_P * SEL_TARGETS) {
                const int q = tid / SEL_TARGETS, j = tid - q * SEL_TARGETS;
                SelTarget t = tg[q][j];
                if (!t.done && t.count > SEL_CAP) {
                    const unsigned *h = hist + (q * SEL_TARGETS + t.alias) * SEL_BINS;
                    const int down = t.shift > 8 ? t.shift - 8 : 0;
                    long long seen = t.below;
                    int b = 0;
                    for (; b < SEL_BINS - 1; ++b) {
                        if (seen + (long long)h[b] > t.rank) break;
                        seen += h[b];
                    }
                    t.lo += (unsigned long long)b << down;
                    t.below = seen;
                    t.count = h[b];
                    t.shift = down;
                    if (down == 0) { t.done = 1; t.value = key_value(t.lo); }     // a single key: that is the value
                    tg[q][j] = t;
                }
            }
            __syncthreads();
            if (tid < TILE_P) {      // intervals have moved: find the aliases again
                for (int j = 0; j < SEL_TARGETS; ++j) {
                    int a = j;
                    for (int i = 0; i < j; ++i)
                        if (tg[tid][i].lo == tg[tid][j].lo && tg[tid][i].shift == tg[tid][j].shift) { a = i; break; }
                    tg[tid][j].alias = a;
                }
            }
            __syncthreads();
        }

        // ---- last sweep: collect the values of the remaining intervals, pick the ranks by counting
        if (tid < TILE_P * SEL_TARGETS) list_n[tid / SEL_TARGETS][tid % SEL_TARGETS] = 0;
        __syncthreads();
        {
            unsigned long long lo[SEL_TARGETS];
            int sh[SEL_TARGETS], slot[SEL_TARGETS];
            constexpr int nsel = SEL_TARGETS;
            bool any = false;
#pragma unroll
            for (int j = 0; j < SEL_TARGETS; ++j) {
                const SelTarget &t = tg[pt][j];
                const bool on = !t.done && t.alias == j;
                lo[j] = on ? t.lo : ~0ull;
                sh[j] = on ? t.shift : -1;
                slot[j] = j;
                any |= on;
            }
            if (any)
                for (int64_t base = 0; base < N; base += SEL_ROWS * SEL_UNROLL) {
                    double xs[SEL_UNROLL];
                    masked_rows<SEL_UNROLL>(iv, select, base + (int64_t)rl * SEL_UNROLL, N, per, p, have_p, xs);
#pragma unroll
                    for (int u = 0; u < SEL_UNROLL; ++u) {
                        const double x = xs[u];
                        if (x == x) {
                            const unsigned long long k = order_key(x);
#pragma unroll
                            for (int a = 0; a < SEL_TARGETS; ++a)
                                if (a < nsel && in_interval(k, lo[a], sh[a])) {
                                    const int at = atomicAdd(&list_n[pt][slot[a]], 1);
                                    if (at < SEL_CAP) cand[(pt * SEL_TARGETS + slot[a]) * SEL_CAP + at] = x;
                                }
                        }
                    }
                }
        }
        __syncthreads();
        // one (point, rank) per group of 4 threads... simply: every thread takes candidates of one target in turn
        for (int item = tid; item < TILE_P * SEL_TARGETS * SEL_CAP; item += SEL_THREADS) {
            const int e = item % SEL_CAP, qj = item / SEL_CAP, q = qj / SEL_TARGETS, j = qj - q * SEL_TARGETS;
            const SelTarget &t = tg[q][j];
            if (t.done) continue;
            const int n = list_n[q][t.alias];
            if (e >= n) continue;
            const double *lst = cand + (q * SEL_TARGETS + t.alias) * SEL_CAP;
            const double v = lst[e];
            int less = 0, leq = 0;
            for (int i = 0; i < n; ++i) {
                const double w = lst[i];
                less += w < v;
                leq += w <= v;
            }
            const long long k = t.rank - t.below;
            if (less <= k && k < leq) tg[q][j].value = v;      // every writer stores the same value
        }
        __syncthreads();
        if (tid < TILE_P && p0 + tid < per) {
            const long long m = cnt[tid];
            const int o = p0 + tid;
            if (m == 0) {
                out[0 * per + o] = out[1 * per + o] = out[2 * per + o] = out[3 * per + o] = nan;
            } else {
                out[0 * per + o] = sum[tid] / (double)m;
                const double a = tg[tid][0].value, b = tg[tid][1].value;
                out[1 * per + o] = (m & 1) ? b : (a + b) / 2.0;
                for (int q = 0; q < 2; ++q) {
                    // numpy's _get_gamma and _lerp on the two neighbours
                    const double virt = (double)(m - 1) * (q == 0 ? 5.0 / 100.0 : 95.0 / 100.0);
                    double prev = floor(virt);
                    if (virt >= (double)(m - 1)) prev = -1.0;
                    else if (virt < 0.0) prev = 0.0;
                    const double gamma = virt - prev;
                    const double lo_v = tg[tid][2 + 2 * q].value, hi_v = tg[tid][3 + 2 * q].value;
                    const double diff = hi_v - lo_v;
                    double r = lo_v + diff * gamma;
                    if (gamma >= 0.5) r = hi_v - diff * (1.0 - gamma);
                    out[(2 + q) * per + o] = r;
                }
            }
        }
    }
}

// ---- density-Wasserstein term of the calibration's second stage -------------------------------
// run_single_heston_calibration.py:100-151 for every (candidate, day, maturity).  Both sides go
// through the same steps -- second differences of the price ratios on the fine strike grid,
// times exp(r tau), clipped at zero, normalised -- and scipy.stats.wasserstein_distance on a common
// support reduces to sum_i |U_i - V_i| (x_{i+1} - x_i) with U, V the normalised cumulative
// weights.  So each side is turned into its CDF once (density_cdf_kernel, one thread per
// (set, maturity)) and the distance matrix is an L1 distance between CDF rows.
//
// status per (set, maturity): 0 = usable, 1 = the reference returns inf (model ratio NaN or
// negative, :118-119, or a density that sums to < 1e-10, :139-140), 2 = maturity absent (NaN
// row: the caller marks maturities with fewer than 3 market points this way, :375-377).

// np.sum of n <= 128 doubles: NumPy's pairwise routine with 8 running sums.
__device__ double numpy_sum_small(const double *a, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

__global__ __launch_bounds__(64) void density_cdf_kernel(GridView g, const double *__restrict__ ratio, int64_t M,
                                                         int is_model, double *__restrict__ cdf,
                                                         double *__restrict__ status)
{
    const int nT = g.nT, nK = g.nK;
    const int64_t item = (int64_t)blockIdx.x * 64 + threadIdx.x;       // (set, maturity)
    if (item >= M * nT) return;
    const int t = (int)(item % nT);
    const double *pr = ratio + item * nK;
    double *out = cdf + item * nK;
    double dens[AFIV_MAX_STRIKES];
    bool any_nan = false, any_neg = false;
    for (int i = 0; i < nK; ++i) {
        const double v = pr[i];
        any_nan |= v != v;
        any_neg |= v < 0;
    }
    double st = 0.0;
    if (is_model ? (any_nan || any_neg) : false) st = 1.0;
    if (!is_model && any_nan) st = 2.0;
    if (st == 0.0) {
        const double dlm = g.logm[1] - g.logm[0];
        const double scale = exp(g.r * g.tau[t]);
        dens[0] = 0.0;
        dens[nK - 1] = 0.0;
        for (int i = 1; i + 1 < nK; ++i) {
            const double d2 = (pr[i + 1] - 2 * pr[i] + pr[i - 1]) / (dlm * dlm);
            dens[i] = fmax(scale * d2, 0.0);
        }
        const double total = numpy_sum_small(dens, nK);
        if (total < 1e-10) {
            st = 1.0;
        } else {
            for (int i = 0; i < nK; ++i) dens[i] = dens[i] / total;
            // cumulative weights, then divided by the last one as scipy's _cdf_distance does
            double run = 0.0;
            for (int i = 0; i < nK; ++i) { run += dens[i]; dens[i] = run; }
            for (int i = 0; i < nK; ++i) out[i] = dens[i] / run;
        }
    }
    status[item] = st;
}

__global__ __launch_bounds__(256) void wasserstein_kernel(GridView g, const double *__restrict__ cdf_model,
                                                          const double *__restrict__ st_model, int64_t C,
                                                          const double *__restrict__ cdf_market,
                                                          const double *__restrict__ st_market, int64_t D,
                                                          double *__restrict__ dist)
{
    __shared__ double model[MAX_POINTS];
    __shared__ double stm[AFIV_MAX_MATURITIES];
    __shared__ double dx[AFIV_MAX_STRIKES];
    const int nT = g.nT, nK = g.nK, per = nT * nK;
    for (int k = threadIdx.x; k + 1 < nK; k += 256) dx[k] = g.logm[k + 1] - g.logm[k];
    for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
        __syncthreads();
        for (int i = threadIdx.x; i < per; i += 256) model[i] = cdf_model[c * per + i];
        for (int t = threadIdx.x; t < nT; t += 256) stm[t] = st_model[c * nT + t];
        __syncthreads();
        for (int64_t item = threadIdx.x; item < D * nT; item += 256) {
            const int t = (int)(item % nT);
            const double sm = stm[t], sk = st_market[item];
            double w;
            if (sk == 2.0) w = __builtin_nan("");
            else if (sm != 0.0 || sk != 0.0) w = __builtin_inf();
            else {
                const double *u = cdf_market + item * nK;
                const double *v = model + t * nK;
                w = 0.0;
                for (int i = 0; i + 1 < nK; ++i) w += fabs(u[i] - v[i]) * dx[i];
            }
            dist[c * D * nT + item] = w;
        }
    }
}

}  // namespace

int afiv_launch_density_wasserstein(const GridView &g, const double *model_ratio, int64_t C,
                                    const double *market_ratio, int64_t D, double *scratch, double *dist,
                                    hipStream_t s)
{
    const int64_t per = (int64_t)g.nT * g.nK;
    double *cdf_model = scratch, *cdf_market = cdf_model + C * per;
    double *st_model = cdf_market + D * per, *st_market = st_model + C * g.nT;
    hipLaunchKernelGGL(density_cdf_kernel, dim3((unsigned)((C * g.nT + 63) / 64)), dim3(64), 0, s, g, model_ratio, C, 1,
                       cdf_model, st_model);
    hipLaunchKernelGGL(density_cdf_kernel, dim3((unsigned)((D * g.nT + 63) / 64)), dim3(64), 0, s, g, market_ratio, D, 0,
                       cdf_market, st_market);
    const int64_t blocks = C < 4096 ? C : 4096;
    hipLaunchKernelGGL(wasserstein_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, cdf_model, st_model, C, cdf_market,
                       st_market, D, dist);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}

int afiv_launch_ensemble(const double *iv, const uint8_t *select, int64_t N, int per, int64_t segments, double *out,
                         hipStream_t s)
{
    if (N > AFIV_ENSEMBLE_ONE_SWEEP) {
        // (1000 points: 250 workgroups of 4 points take 41 ms for a million surfaces, 500 of 2 points
        // 63 ms -- narrower tiles re-fetch more lines than their parallelism buys back; 168 points:
        // 42 workgroups of 4 take 2.0 ms for 50 000 surfaces, 168 of 1 point 1.3 ms)
        const int tp = per >= 4 * 128 ? 4 : per >= 2 * 128 ? 2 : 1;
        const int tiles = (per + tp - 1) / tp;
        const dim3 blocks((unsigned)(tiles < 65536 ? tiles : 65536));
        for (int64_t g = 0; g < segments; ++g) {
            const double *iv_g = iv + g * N * per;
            const uint8_t *sel_g = select ? select + g * N : nullptr;
            double *out_g = out + g * 4 * per;
            if (tp == 4) hipLaunchKernelGGL(ensemble_select_kernel<4>, blocks, dim3(SEL_THREADS), 0, s, iv_g, sel_g, N, per, out_g);
            else if (tp == 2) hipLaunchKernelGGL(ensemble_select_kernel<2>, blocks, dim3(SEL_THREADS), 0, s, iv_g, sel_g, N, per, out_g);
            else hipLaunchKernelGGL(ensemble_select_kernel<1>, blocks, dim3(SEL_THREADS), 0, s, iv_g, sel_g, N, per, out_g);
            if (hipGetLastError() != hipSuccess) return AFIV_E_HIP;
        }
        return AFIV_OK;
    }
    const int64_t items = segments * per;
    if (N <= SMALL_N) {
        const int64_t blocks = (items + SMALL_WAVES - 1) / SMALL_WAVES;
        hipLaunchKernelGGL(ensemble_small_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(STAT_THREADS), 0, s,
                           iv, select, (int)N, per, segments, out);
        return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
    }
    int padded = 2;
    while (padded < N) padded <<= 1;
    const size_t lds = (size_t)padded * sizeof(double);
    static DynLdsLimit raised;
    if (const int rc = afiv_ensure_dynamic_lds(reinterpret_cast<const void *>(&ensemble_kernel), lds, raised)) return rc;
    hipLaunchKernelGGL(ensemble_kernel, dim3((unsigned)(items < 65536 ? items : 65536)), dim3(STAT_THREADS), lds, s, iv, select, N,
                       per, padded, segments, out);
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}
