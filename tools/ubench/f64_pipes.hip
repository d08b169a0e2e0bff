// Do v_mfma_f64_16x16x4_f64 and FP64 vector FMAs overlap on one SIMD of gfx950?
// mode 0: every wave runs a chain of FMAs; mode 1: every wave runs MFMAs; mode 2: even waves FMAs,
// odd waves MFMAs (4 waves per SIMD -> 2 + 2); mode 3: each wave interleaves both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int iters)
{
    const int wave = threadIdx.x >> 6;
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const bool do_fma = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1) == 0);
    const bool do_mfma = MODE == 1 || MODE == 3 || (MODE == 2 && (wave & 1) == 1);
    for (int i = 0; i < iters; ++i) {
        if (do_fma) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            }
        }
        if (do_mfma) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, m, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, m, acc1, 0, 0, 0);
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc0[1] + acc0[2] + acc0[3] +
                                           acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

template <int MODE>
float run(double *out, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out;
    if (hipMalloc(&out, 256 * 1024 * sizeof(double)) != hipSuccess) return 1;
    const int iters = 20000;
    // per wave and iteration: 32 FMAs (4 cycles each = 128) and/or 2 MFMAs (64 cycles each = 128)
    printf("fma only          %.3f ms\n", run<0>(out, iters));
    printf("mfma only         %.3f ms\n", run<1>(out, iters));
    printf("split waves       %.3f ms  (half the waves each: half the work of each kind)\n", run<2>(out, iters));
    printf("both in each wave %.3f ms\n", run<3>(out, iters));
    hipFree(out);
    return 0;
}
