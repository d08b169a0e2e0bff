// Heston price ratios over a (surface x maturity x strike) lattice on gfx950.
//
// Replaces one QuantLib AnalyticHestonEngine evaluation per option (heston_model_ql.py:87,
// :109-179) by a factored evaluation of the same quadrature:
//
//   C(K,T) = S Dq (P1 + 1/2) - K Dr (P2 + 1/2),   P_j = (1/pi) sum_n W_n Im exp(A_j(x_n) + i x_n b) / x_n
//
// with b = dd_T - log K.  A_j depends on (parameters, T, node) only, so for every (surface,
// maturity) the 288 values E = W_n exp(Re A)/(pi x_n) and alpha = Im A + x_n dd_T are computed once
// (phase 1, FP64 vector work: complex sqrt/exp/log) and all strikes follow from
//
//   sum_n E sin(alpha - x_n log K) = sum_n [E sin alpha] cos(x_n log K) - [E cos alpha] sin(x_n log K),
//
// a [rows x 576] x [576 x strikes] FP64 contraction against strike tables that depend only on the
// grid (phase 2, v_mfma_f64_16x16x4_f64).
//
// Work decomposition: one workgroup = 2 surfaces = 576 threads = 9 full wave64s; thread =
// (surface, j, node).  Maturity-independent quantities (d, p, 1/(1-p), ...) stay in registers
// while the thread walks the maturities in tiles of 8, so a tile fills the 16 rows of one MFMA
// row block (2 surfaces x 8 maturities).  Phase 1 writes the 16 x 576 operand block to LDS;
// phase 2 splits the 576-long contraction over the 9 waves (64 each), each wave issuing
// 16 x (nKpad/16) MFMAs with B fragments streamed from the L2-resident table; partial tiles are
// reduced through LDS in a fixed order (bitwise reproducible).
#include "afiv_internal.h"
#include "laguerre144.inc"

namespace {

constexpr int NQ = AFIV_NQ_NODES;          // 144
constexpr int KDIM = AFIV_KDIM;            // 576
constexpr int SURF_PER_BLOCK = 2;
constexpr int THREADS = SURF_PER_BLOCK * 2 * NQ;   // 576
constexpr int WAVES = THREADS / 64;        // 9
constexpr int MT = 8;                      // maturities per tile
constexpr int ROWS = SURF_PER_BLOCK * MT;  // 16 = MFMA M
constexpr int XSTRIDE = KDIM + 2;          // doubles; 1156 dwords = 4 mod 64 -> conflict-free A reads
constexpr int KSLICE = KDIM / WAVES;       // 64 contraction indices per wave
constexpr int AFIV_MAX_NKPAD = 64;
static_assert(KDIM % WAVES == 0 && KSLICE % 4 == 0, "split-K layout");

typedef double v4d __attribute__((ext_vector_type(4)));

__constant__ double c_lag_x[NQ] = {AFIV_LAG_X_LIST};
__constant__ double c_lag_wp[NQ] = {AFIV_LAG_WP_LIST};   // W_n / (pi x_n)

struct Cx { double r, i; };
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ Cx cdiv(Cx a, Cx b)
{
    const double inv = 1.0 / (b.r * b.r + b.i * b.i);
    return {(a.r * b.r + a.i * b.i) * inv, (a.i * b.r - a.r * b.i) * inv};
}
__device__ __forceinline__ Cx csqrt_principal(Cx z)
{
    const double m = sqrt(z.r * z.r + z.i * z.i);
    Cx d;
    if (z.r >= 0.0) {
        d.r = sqrt(0.5 * (m + z.r));
        d.i = (d.r > 0.0) ? 0.5 * z.i / d.r : 0.0;
    } else {
        const double t = sqrt(0.5 * (m - z.r));
        d.r = 0.5 * fabs(z.i) / t;
        d.i = copysign(t, z.i);
    }
    return d;
}

template <int NCT>
__global__ __launch_bounds__(THREADS) void price_kernel(GridView g, const double *__restrict__ params,
                                                        int64_t N, double *__restrict__ ratio_out)
{
    __shared__ double lds[ROWS * XSTRIDE];   // operand block, reused for the partial tiles
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int gsurf = tid / (2 * NQ);        // 0/1: which of the block's two surfaces
    const int rem = tid - gsurf * 2 * NQ;
    const int jj = rem / NQ;                 // 0 -> P1, 1 -> P2
    const int n = rem - jj * NQ;
    const double phi = c_lag_x[n];
    const double wp = c_lag_wp[n];
    const double sj = jj == 0 ? 1.0 : -1.0;
    const int ntiles = (g.nT + MT - 1) / MT;
    const int nKpad = g.nKpad;

    for (int64_t pair = blockIdx.x; pair * SURF_PER_BLOCK < N; pair += gridDim.x) {
        const int64_t s0 = pair * SURF_PER_BLOCK;
        int64_t s = s0 + gsurf;
        if (s >= N) s = N - 1;               // odd tail: duplicate work, never stored
        const double kappa = params[5 * s + 0], theta = params[5 * s + 1], sigma = params[5 * s + 2],
                     rho = params[5 * s + 3], v0 = params[5 * s + 4];

        // ---- maturity-independent part of A_j(phi) ------------------------------------------
        const double sigma2 = sigma * sigma;
        const double rsig = rho * sigma;
        const Cx t1 = {kappa - (jj == 0 ? rsig : 0.0), -(rsig * phi)};
        const Cx u = {-phi, sj};
        const Cx zz = {t1.r * t1.r - t1.i * t1.i + sigma2 * phi * phi,
                       2.0 * t1.r * t1.i - sigma2 * phi * sj};
        const Cx d = csqrt_principal(zz);
        const bool main_branch = sigma > 1e-5;
        const double kt = kappa * theta;
        Cx a, p, inv1mp;
        double c1, c2;
        if (main_branch) {
            a = {t1.r - d.r, t1.i - d.i};                       // t1 - d
            p = cdiv(a, Cx{t1.r + d.r, t1.i + d.i});
            inv1mp = cdiv(Cx{1.0, 0.0}, Cx{1.0 - p.r, -p.i});
            c1 = v0 / sigma2;
            c2 = kt / sigma2;
        } else {
            // small vol-of-vol series: td = phi/(2 t1) * u ; p = td sigma^2/(t1+d)
            const Cx h = cdiv(Cx{phi, 0.0}, Cx{2.0 * t1.r, 2.0 * t1.i});
            a = cmul(h, u);                                     // td
            const Cx q1 = cdiv(Cx{a.r * sigma2, a.i * sigma2}, Cx{t1.r + d.r, t1.i + d.i});
            p = q1;
            inv1mp = {0.0, 0.0};
            c1 = v0;
            c2 = kt;
        }

        for (int tile = 0; tile < ntiles; ++tile) {
            // ---- phase 1: operand rows for up to MT maturities ------------------------------
            for (int tl = 0; tl < MT; ++tl) {
                const int t = tile * MT + tl;
                double x0 = 0.0, x1 = 0.0;
                if (t < g.nT) {
                    const double T = g.term[t];
                    const double er = exp(-d.r * T);
                    double sn, cs;
                    sincos(d.i * T, &sn, &cs);
                    const Cx ex = {er * cs, -er * sn};           // exp(-d T)
                    const Cx pe = cmul(p, ex);
                    const Cx num = {1.0 - pe.r, -pe.i};          // 1 - p ex
                    const Cx omex = {1.0 - ex.r, -ex.i};         // 1 - ex
                    Cx A;
                    if (main_branch) {
                        const Cx w = cmul(num, inv1mp);          // (1 - p ex)/(1 - p)
                        const Cx gl = {0.5 * log(w.r * w.r + w.i * w.i), atan2(w.i, w.r)};
                        const Cx f1 = cdiv(cmul(a, omex), num);
                        A.r = c1 * f1.r + c2 * (a.r * T - 2.0 * gl.r);
                        A.i = c1 * f1.i + c2 * (a.i * T - 2.0 * gl.i);
                    } else {
                        const Cx gl = cmul(p, omex);             // g = p (1 - ex)
                        const Cx f1 = cdiv(cmul(a, omex), num);
                        A.r = c1 * f1.r + c2 * (a.r * T - 2.0 * gl.r / sigma2);
                        A.i = c1 * f1.i + c2 * (a.i * T - 2.0 * gl.i / sigma2);
                    }
                    const double alpha = A.i + phi * g.dd[t];
                    const double E = wp * exp(A.r) * (jj == 0 ? g.Dq[t] : g.Dr[t]);
                    double sa, ca;
                    sincos(alpha, &sa, &ca);
                    x0 = E * sa;
                    x1 = E * ca;
                }
                double2 *dst = reinterpret_cast<double2 *>(&lds[(gsurf * MT + tl) * XSTRIDE + 2 * (jj * NQ + n)]);
                *dst = make_double2(x0, x1);
            }
            __syncthreads();

            // ---- phase 2: [16 x 64-slice] x [64-slice x nKpad] per wave on the matrix pipe ---
            v4d acc[NCT];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = v4d{0.0, 0.0, 0.0, 0.0};
            {
                const int arow = lane & 15;
                const int kq = lane >> 4;
                const double *xa = &lds[arow * XSTRIDE + wave * KSLICE + kq];
                const double *tb = g.tab + (size_t)(wave * KSLICE + kq) * nKpad + (lane & 15);
#pragma unroll 4
                for (int ks = 0; ks < KSLICE / 4; ++ks) {
                    const double av = xa[4 * ks];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const double bv = tb[(size_t)(4 * ks) * nKpad + 16 * ct];
                        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ct], 0, 0, 0);
                    }
                }
            }
            __syncthreads();                  // every wave is done reading the operand block
            {
                // D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4*reg
                double *pw = &lds[wave * ROWS * nKpad];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        pw[((lane >> 4) + 4 * i) * nKpad + 16 * ct + (lane & 15)] = acc[ct][i];
            }
            __syncthreads();
            for (int o = tid; o < ROWS * nKpad; o += THREADS) {
                const int row = o / nKpad;
                const int col = o - row * nKpad;
                const int gs = row / MT;
                const int t = tile * MT + (row - gs * MT);
                if (t < g.nT && col < g.nK && s0 + gs < N) {
                    double sum = 0.0;
#pragma unroll
                    for (int w = 0; w < WAVES; ++w) sum += lds[(w * ROWS + row) * nKpad + col];
                    const double Dq = g.Dq[t], Dr = g.Dr[t], K = g.Kmult[col];
                    double call = (0.5 * Dq - 0.5 * (K * Dr)) + sum;
                    if (!(g.term[t] > 0.0)) call = 0.0;          // expired option: NPV 0
                    ratio_out[((s0 + gs) * g.nT + t) * (int64_t)g.nK + col] = call / g.disc_tau[t];
                }
            }
            __syncthreads();                  // partial tiles consumed before the next phase 1
        }
    }
}

static_assert(WAVES * ROWS * AFIV_MAX_NKPAD <= ROWS * XSTRIDE, "partial tiles must fit in the operand block");

}  // namespace

int afiv_launch_price(const GridView &g, const double *params, int64_t N, double *ratio_out, hipStream_t s)
{
    if (N <= 0) return 0;
    const int nct = g.nKpad / 16;
    const size_t lds_bytes = 0;   // LDS is static
    int64_t pairs = (N + SURF_PER_BLOCK - 1) / SURF_PER_BLOCK;
    int64_t blocks = pairs < 256 * 64 ? pairs : 256 * 64;
    dim3 grid((unsigned)blocks), block(THREADS);
    switch (nct) {
        case 1: hipLaunchKernelGGL(price_kernel<1>, grid, block, lds_bytes, s, g, params, N, ratio_out); break;
        case 2: hipLaunchKernelGGL(price_kernel<2>, grid, block, lds_bytes, s, g, params, N, ratio_out); break;
        case 3: hipLaunchKernelGGL(price_kernel<3>, grid, block, lds_bytes, s, g, params, N, ratio_out); break;
        case 4: hipLaunchKernelGGL(price_kernel<4>, grid, block, lds_bytes, s, g, params, N, ratio_out); break;
        default: return -2;
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
