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

}  // namespace

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
