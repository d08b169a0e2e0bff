// CVAE decoder (12 -> 32 -> 64 -> 128 -> 5, ReLU, eval-mode BatchNorm applied as one scale and
// shift per neuron) followed by de-normalisation and the exp / tanh transforms, in FP32 on gfx950.
//
// Replaces ConditionalVAE_SingleHeston.decode (cvae_model.py:145-160; stack :84-105) and
// generate_iv_surface_by_date.py:441-446.  The network is 11 k MACs per sample -- far too small
// for the matrix pipe -- so one workgroup keeps the 47 KB of weights in LDS and pushes
// tiles of 16 samples through the layers with lanes mapped to output neurons (weight reads are
// lane-contiguous, activation reads are LDS broadcasts).
//
// Compiled with -ffp-contract=off so that "x * std + mean" rounds as torch does (:441).
#include "afiv_internal.h"

namespace {

constexpr int W1 = 12 * 32, W2 = 32 * 64, W3 = 64 * 128, W4 = 128 * 5;
constexpr int WTOTAL = W1 + 3 * 32 + W2 + 3 * 64 + W3 + 3 * 128 + W4 + 5;

// out[s][o] = act(b[o] + sum_i in[s][i] * w[i*OUT + o]) for the tile; thread = (o, sample group)
// packed layout per hidden layer: w[IN*OUT], b[OUT], scale[OUT], shift[OUT]
template <int IN, int OUT, int SPT, bool HIDDEN, int THREADS, int TS>
__device__ __forceinline__ void layer(const float *__restrict__ w,
                                      const float *__restrict__ in /*[TS][IN]*/, float *__restrict__ out /*[TS][OUT]*/,
                                      int tid)
{
    constexpr int GROUPS = TS / SPT;
    static_assert(OUT * GROUPS <= THREADS, "one (neuron, sample group) per thread");
    if (tid < OUT * GROUPS) {
        const int o = tid % OUT;
        const int s0 = (tid / OUT) * SPT;
        float acc[SPT];
        const float *b = w + IN * OUT;
#pragma unroll
        for (int s = 0; s < SPT; ++s) acc[s] = b[o];
        for (int i = 0; i < IN; ++i) {
            const float wv = w[i * OUT + o];
#pragma unroll
            for (int s = 0; s < SPT; ++s) acc[s] = __builtin_fmaf(in[(s0 + s) * IN + i], wv, acc[s]);
        }
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            float y = acc[s];
            if (HIDDEN) y = fmaxf(y * b[OUT + o] + b[2 * OUT + o], 0.0f);   // BatchNorm(eval), ReLU
            out[(s0 + s) * OUT + o] = y;
        }
    }
}

// THREADS = 256 on tiles of 32 samples with 4 / 8 / 16 samples per thread in the three hidden layers:
// every weight read feeds up to sixteen multiply-adds, and two workgroups (79 KB of LDS each) give a
// CU eight waves -- the shape for batches that fill the chip (128 threads on 16 samples left one
// wave per SIMD).  THREADS = 512 on 16 samples with
// 1 / 2 / 4 samples per thread: four times the lanes on the same tile, for batches of fewer tiles
// than CUs, where a call's time is one tile's latency (a 100-sample decode -> surface call: 42 -> 35 us).  Same arithmetic
// per output either way (one accumulation chain per (sample, neuron), inputs in order).
template <int THREADS, int TS, int S1, int S2, int S3>
__global__ __launch_bounds__(THREADS) void decode_kernel(DecoderView d, const float *__restrict__ z,
                                                         const float *__restrict__ c, int64_t N, int64_t c_rows,
                                                         float *__restrict__ norm_out, float *__restrict__ p32,
                                                         double *__restrict__ p64)
{
    __shared__ float wts[WTOTAL];
    __shared__ float bufA[TS * 128];
    __shared__ float bufB[TS * 128];
    const int tid = threadIdx.x;
    const float *l1 = wts, *l2 = l1 + W1 + 3 * 32, *l3 = l2 + W2 + 3 * 64, *l4 = l3 + W3 + 3 * 128;
    for (int i = tid; i < WTOTAL; i += THREADS) wts[i] = d.packed[i];
    __syncthreads();

    const int64_t ntiles = (N + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * TS;
        // input rows [z | c]  (cvae_model.py:157)
        for (int i = tid; i < TS * 12; i += THREADS) {
            const int s = i / 12, f = i - s * 12;
            int64_t row = base + s;
            if (row >= N) row = N - 1;
            float v;
            if (f < 4) v = z[row * 4 + f];
            else v = c[(c_rows == 1 ? 0 : row) * 8 + (f - 4)];
            bufA[s * 12 + f] = v;
        }
        __syncthreads();
        layer<12, 32, S1, true, THREADS, TS>(l1, bufA, bufB, tid);
        __syncthreads();
        layer<32, 64, S2, true, THREADS, TS>(l2, bufB, bufA, tid);
        __syncthreads();
        layer<64, 128, S3, true, THREADS, TS>(l3, bufA, bufB, tid);
        __syncthreads();
        layer<128, 5, 1, false, THREADS, TS>(l4, bufB, bufA, tid);
        __syncthreads();
        static_assert(TS * 5 <= THREADS, "one output value per thread");
        if (tid < TS * 5) {
            const int s = tid / 5, f = tid - s * 5;
            const int64_t row = base + s;
            if (row < N) {
                const float nv = bufA[s * 5 + f];
                const float t = nv * d.pstd[f] + d.pmean[f];
                const float p = (f == 3) ? tanhf(t) : expf(t);
                if (norm_out) norm_out[row * 5 + f] = nv;
                if (p32) p32[row * 5 + f] = p;
                if (p64) p64[row * 5 + f] = (double)p;
            }
        }
        __syncthreads();
    }
}

}  // namespace

int afiv_launch_decode(const DecoderView &d, const float *z, const float *c, int64_t N, int64_t c_rows,
                       float *normalized_out, float *p32, double *p64, hipStream_t s)
{
    if (N <= 0) return 0;
    // fewer 16-sample tiles than CUs: four times the lanes per tile; else tiles of 32 samples on 256
    // threads, two workgroups (eight waves) per CU
    if ((N + 15) / 16 < 256) {
        const int64_t tiles = (N + 15) / 16;
        hipLaunchKernelGGL((decode_kernel<512, 16, 1, 2, 4>), dim3((unsigned)tiles), dim3(512), 0, s, d, z, c, N, c_rows,
                           normalized_out, p32, p64);
    } else {
        const int64_t tiles = (N + 31) / 32;
        const int64_t blocks = tiles < 256 * 2 ? tiles : 256 * 2;
        hipLaunchKernelGGL((decode_kernel<256, 32, 4, 8, 16>), dim3((unsigned)blocks), dim3(256), 0, s, d, z, c, N, c_rows,
                           normalized_out, p32, p64);
    }
    return hipGetLastError() == hipSuccess ? AFIV_OK : AFIV_E_HIP;
}
