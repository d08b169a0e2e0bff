// Shader clock under load and issue cost per instruction, measured directly: s_memtime counts shader
// cycles, s_memrealtime a constant 100 MHz, so their ratio over a loop is the clock the chip holds
// while every SIMD runs that loop (4 waves each), and cycles / (4 waves x instructions) is the issue
// cost of one wave instruction.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int CH = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, unsigned long long *ticks, int iters, double m, double c)
{
    double a[CH];
    unsigned b[CH];
    for (int j = 0; j < CH; ++j) { a[j] = threadIdx.x * 1e-3 + j; b[j] = threadIdx.x + j; }
    v4d acc = {0, 0, 0, 0};
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
#pragma unroll 8
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_fma(a[j], m, c);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = b[j] * 3u + (unsigned)i;          // v_mad_u32_u24 or v_mul_lo + v_add
        } else if (MODE == 2) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], m, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], m, acc, 0, 0, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] * m;
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] + c;
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = (b[j] ^ (unsigned)i) + 7u;          // two 32-bit ALU ops
        } else if (MODE == 6) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_amdgcn_rcp(a[j]);
        } else if (MODE == 7) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = a[j] > c ? a[j] : m;               // v_cmp + 2 v_cndmask
        } else if (MODE == 8) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_rint(a[j] * m);           // v_mul + v_rndne_f64
        } else if (MODE == 9) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_amdgcn_fract(a[j] * m);   // v_mul + v_fract_f64
        } else if (MODE == 10) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_amdgcn_ldexp(a[j] * m, (int)b[j] & 1);   // v_mul + v_and + v_ldexp_f64
        } else if (MODE == 11) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = (double)(int)(a[j] * m);            // v_mul + v_cvt_i32_f64 + v_cvt_f64_i32
        } else if (MODE == 12) {
#pragma unroll
            for (int j = 0; j < CH; ++j) a[j] = __builtin_amdgcn_sqrt(a[j]);        // v_sqrt_f64 (no refinement)
        } else if (MODE == 13) {
#pragma unroll
            for (int j = 0; j < CH; ++j) b[j] = b[j] * (b[j] | 3u);                 // v_or + v_mul_lo_u32
        }
    }
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int j = 0; j < CH; ++j) s += a[j] + b[j];
    // every wave of the workgroup has finished its loop (and its results exist) before the end stamps
    __shared__ double sink[1024];
    sink[threadIdx.x] = s;
    __syncthreads();
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
void run(const char *name, double *out, unsigned long long *ticks, int iters, double insts_per_iter)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, ticks, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, ticks, iters, 1.0000001, 1e-9);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[512];
    (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < 256; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
    cyc /= 256; real /= 256;
    printf("%-12s %.3f ms by events; per workgroup %.0f shader cycles in %.0f ticks of 100 MHz -> %.3f GHz; "
           "%.2f cycles per wave instruction (4 waves per SIMD)\n", name, ms, cyc, real, cyc / (real * 10.0),
           cyc / (insts_per_iter * iters * 4));
}

int main()
{
    double *out; unsigned long long *ticks;
    if (hipMalloc(&out, 256 * 1024 * sizeof(double)) != hipSuccess) return 1;
    if (hipMalloc(&ticks, 512 * sizeof(unsigned long long)) != hipSuccess) return 1;
    const int iters = 20000;
    run<0>("(warm-up)", out, ticks, 5 * iters, CH);
    run<2>("mfma_f64", out, ticks, 2 * iters, 2);
    run<0>("v_fma_f64", out, ticks, 5 * iters, CH);
    run<3>("v_mul_f64", out, ticks, 5 * iters, CH);
    run<4>("v_add_f64", out, ticks, 5 * iters, CH);
    run<6>("v_rcp_f64", out, ticks, 2 * iters, CH);
    run<1>("u32 mul+add", out, ticks, 5 * iters, CH);
    run<5>("u32 xor, add", out, ticks, 5 * iters, 2 * CH);
    run<7>("f64 cmp+sel", out, ticks, 5 * iters, 3 * CH);
    run<8>("mul, rndne", out, ticks, 5 * iters, 2 * CH);
    run<9>("mul, fract", out, ticks, 5 * iters, 2 * CH);
    run<10>("mul,and,ldexp", out, ticks, 5 * iters, 3 * CH);
    run<11>("mul, 2 cvt", out, ticks, 5 * iters, 3 * CH);
    run<12>("v_sqrt_f64", out, ticks, 2 * iters, CH);
    run<13>("or, mul_lo_u32", out, ticks, 5 * iters, 2 * CH);
    (void)hipFree(out); (void)hipFree(ticks);
    return 0;
}
