// Issue cost of vector instructions on one gfx950 SIMD with 4 resident waves: cycles per wave
// instruction = time x clock / (instructions per wave x 4 waves); the clock is taken from a
// v_mfma_f64_16x16x4_f64 loop (64 cycles each).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int CH = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double m, double c)
{
    double a[CH];
    unsigned b[CH];
    for (int j = 0; j < CH; ++j) { a[j] = threadIdx.x * 1e-3 + j; b[j] = threadIdx.x + j; }
    v4d acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_fma(a[j], m, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] * m;
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] + c;
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = b[j] * 3u + 1u;       // v_mad_u32_u24 / v_mul_lo + add
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = (b[j] ^ (unsigned)i) + 7u;   // two 32-bit ALU ops
        } else if (MODE == 5) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], m, acc, 0, 0, 0);
        } else if (MODE == 6) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_amdgcn_rcp(a[j]);
        } else if (MODE == 7) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] > c ? a[j] : m;        // v_cmp + 2 v_cndmask
        }
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int j = 0; j < CH; ++j) s += a[j] + b[j];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}

template <int MODE>
float run(double *out, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    double *out;
    if (hipMalloc(&out, 256 * 1024 * sizeof(double)) != hipSuccess) return 1;
    const int iters = 20000;
    const float t_mfma = run<5>(out, iters);
    const double ghz = 2.0 * 64 * iters * 4 / (t_mfma * 1e-3) / 1e9;
    printf("mfma loop %.3f ms -> clock %.3f GHz if an MFMA takes 64 cycles\n", t_mfma, ghz);
    const char *name[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "u32 mul+add", "u32 xor+add", "", "v_rcp_f64", "f64 cmp+select"};
    const float t[] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), 0.f,
                       run<6>(out, iters), run<7>(out, iters)};
    for (int mode = 0; mode < 8; ++mode) {
        if (mode == 5) continue;
        printf("%-16s %.3f ms  = %.2f cycles per source-level operation per wave\n", name[mode], t[mode],
               t[mode] * 1e-3 * ghz * 1e9 / ((double)CH * iters * 4));
    }
    (void)hipFree(out);
    return 0;
}
