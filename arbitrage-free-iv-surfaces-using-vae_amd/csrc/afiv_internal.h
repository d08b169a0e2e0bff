// Shared declarations between the translation units of libafiv.so (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "afiv.h"

#define AFIV_NQ_NODES 144                 // Gauss-Laguerre nodes (QuantLib's default order)
#define AFIV_KDIM (4 * AFIV_NQ_NODES)     // contraction length: 2 integrals x 144 nodes x (sin,cos)

// Device view of a grid: everything that depends only on (logm, tau, r, q).  All pointers are
// device memory owned by the afiv_grid.  Passed to kernels by value.
struct GridView {
    int nT, nK, nKpad;        // nKpad = nK rounded up to a multiple of 16 (MFMA column tiles)
    double r, q;
    const double *tau;        // [nT] maturities as given
    const double *term;       // [nT] year fraction actually priced (floor(365 tau)/365 or tau)
    const double *dd;         // [nT] -log(Dr/Dq) (sym: minus the strike grid's midpoint)
    const double *Dq;         // [nT] 1/exp(q*term)
    const double *Dr;         // [nT] 1/exp(r*term)
    const double *disc_tau;   // [nT] exp(-r*tau)
    const double *fwd_mult;   // [nT] exp((r-q)*tau)
    const double *dq_tau;     // [nT] exp(-q*tau)
    const double *logm;       // [nK]
    const double *Kmult;      // [nK] exp(logm)
    const double *tab;        // [AFIV_KDIM][nKpad] strike tables, see afiv_price.hip
    // Strike grids symmetric about their midpoint c (log K_k - c = -(log K_{nK-1-k} - c)): sym = 1,
    // nH = (nK + 1)/2 strikes carry the contraction, nHpad = nH rounded up to 16,
    // tab_sym = [2][144][nHpad]: cos, sin of x_n (log K_k - c).
    int sym, nH, nHpad;
    const double *tab_sym;
    // Regular term structures (afiv_price.hip, NFAC): step_count = 0 (none), 1 (successive year
    // fractions are step_dt[0] apart) or 2 (step_dt[0] plus k times step_dt[1] = plus or minus one
    // day, k <= 3 in bits 2t, 2t+1 of step_sel for the step from maturity t-1 to t).
    int step_count;
    double step_dt[2];
    unsigned long long step_sel;
    // Whole-day term structures whose maturities are sums of earlier ones (the reference's own
    // 30, 60, 91, 182, 273, 365, 547, 730 days: 60 = 30 + 30, 182 = 91 + 91, 273 = 182 + 91, ...):
    // step_count = 3 and prog[t] says how 1 - exp(-d T_t) is obtained (afiv_price.hip, NFAC = 3).
    // Low nibble: 0 = evaluated directly, else the product form of two earlier values, each the
    // previous maturity's (P) or one of two saved ones (A, B): 1 = P P, 2 = P A, 3 = P B, 4 = A A,
    // 5 = A B, 6 = B B.  High nibble: 1 / 2 = keep this maturity's value as A / B.
    const unsigned char *prog;     // [AFIV_MAX_MATURITIES] device bytes
    // need[t0] = the maturities below t0 whose values the steps from t0 on descend from (bit t): what a
    // workgroup that starts its walk at maturity t0 replays first
    const unsigned long long *need;   // [AFIV_MAX_MATURITIES] device words
    const unsigned short *order;   // [nT*nK] lattice points by decreasing |log(F/K)|/sqrt(tau): the order
                                   // in which the implied-vol kernel hands points to lanes (points far
                                   // from the money need the most root-finder iterations; served first,
                                   // they leave a short tail)
};

struct afiv_grid {
    GridView v;
    int device;
    void *block;              // one allocation holding every array above
    double *workspace;        // optional [workspace_surfaces][nT][nK]
    int64_t workspace_surfaces;
};

struct DecoderView {
    // One packed float array: for each hidden layer the weights stored input-major
    // (w[i*OUT + o]), the bias, and the eval-mode BatchNorm as scale = gamma/sqrt(var+eps),
    // shift = beta - mean*scale; then the output layer's weights and bias.
    const float *packed;
    const float *pmean, *pstd;
};

struct afiv_decoder {
    DecoderView v;
    int device;
    void *block;
};

// Raising a kernel's dynamic-LDS limit is a per-function, per-device attribute (not an allocation).
// It is set when a launch first needs more than the limit already raised for that kernel on the
// current device, not on every launch: `raised` is the kernel's own table, one entry per device.
#define AFIV_MAX_DEVICES 64
#define AFIV_LDS_PER_CU (160 * 1024)      // gfx950
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "these kernels are written for gfx950: 160 KB of LDS per CU, v_mfma_f64_16x16x4_f64, 64-wide waves"
#endif
struct DynLdsLimit { std::atomic<size_t> bytes[AFIV_MAX_DEVICES]; };
inline int afiv_ensure_dynamic_lds(const void *kernel, size_t bytes, DynLdsLimit &raised)
{
    if (bytes <= 48 * 1024) return AFIV_OK;               // within the default limit
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return AFIV_E_HIP;
    if (dev < 0 || dev >= AFIV_MAX_DEVICES) return AFIV_E_ARG;
    if (raised.bytes[dev].load(std::memory_order_acquire) >= bytes) return AFIV_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return AFIV_E_HIP;
    // racing first launches set the attribute twice; the larger size is the one that stays recorded
    size_t seen = raised.bytes[dev].load(std::memory_order_relaxed);
    while (seen < bytes && !raised.bytes[dev].compare_exchange_weak(seen, bytes, std::memory_order_release)) {}
    return AFIV_OK;
}

// Launchers implemented in the kernel translation units.
int afiv_launch_price(const GridView &g, const double *params, int64_t N, double *ratio_out,
                      hipStream_t s);
int afiv_launch_implied_vol(const GridView &g, const double *ratio, int64_t N, double spot,
                            int variant, double *iv_out, hipStream_t s);
int afiv_launch_arb_check(const GridView &g, const double *iv, const double *ratio, int64_t N,
                          double spot, int bs_form, int32_t *flags_out, uint8_t *mask_out, hipStream_t s);
int afiv_launch_decode(const DecoderView &d, const float *z, const float *c, int64_t N,
                       int64_t c_rows, float *normalized_out, float *p32, double *p64,
                       hipStream_t s);
int afiv_launch_penalty(const GridView &g, const double *ratio, int64_t B, double *acc_out,
                        hipStream_t s);
int afiv_launch_stage1(const GridView &g, const double *model_ratio, const double *params,
                       int64_t C, const double *market_ratio, int64_t D, double drift_weight,
                       double *loss_out, hipStream_t s);
int afiv_launch_debug_math(int op, const double *x, const double *y, int64_t n, double *out0, double *out1,
                           hipStream_t s);
int afiv_launch_debug_ndtr(const double *x, int64_t n, double *out, hipStream_t s);
int afiv_launch_surface_iv(const GridView &g, const double *ratio, int64_t N, double spot, int variant,
                           double *iv_out, int32_t *flags, hipStream_t s);
int afiv_launch_fast_iv(const GridView &g, const double *ratio, int64_t N, double spot, double *iv_out,
                        int32_t *flags, hipStream_t s);
int afiv_launch_ensemble(const double *iv, const uint8_t *select, int64_t N, int per, int64_t segments, double *out,
                         hipStream_t s);
int afiv_launch_density_wasserstein(const GridView &g, const double *model_ratio, int64_t C,
                                    const double *market_ratio, int64_t D, double *scratch, double *dist,
                                    hipStream_t s);
