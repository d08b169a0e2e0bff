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

__global__ __launch_bounds__(STAT_THREADS) void ensemble_kernel(const double *__restrict__ iv,
                                                                const uint8_t *__restrict__ select, int64_t N,
                                                                int per, int padded, double *__restrict__ out)
{
    extern __shared__ double vals[];              // [padded]
    __shared__ int wave_count[STAT_THREADS / 64];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double inf = __builtin_inf();
    for (int p = blockIdx.x; p < per; p += gridDim.x) {
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
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int afiv_launch_ensemble(const double *iv, const uint8_t *select, int64_t N, int per, double *out,
                         hipStream_t s)
{
    int padded = 2;
    while (padded < N) padded <<= 1;
    const size_t lds = (size_t)padded * sizeof(double);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&ensemble_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -3;
    }
    hipLaunchKernelGGL(ensemble_kernel, dim3((unsigned)(per < 65536 ? per : 65536)), dim3(STAT_THREADS), lds, s, iv, select, N, per, padded,
                       out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
