// Host side of libafiv.so: the extern "C" entry points declared in include/afiv.h.
// Creation calls allocate and fill device memory; execution calls only validate their
// arguments and enqueue kernels on the caller's stream.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "afiv_internal.h"
#include "laguerre144.inc"

// Price ratios of afiv_surface_static pass through this array instead of caller or grid memory.
// Every GPU the library runs on holds its own copy (128 MiB) at its own address until the process
// ends.  In host code the name is only a handle for hipGetSymbolAddress: kernels are given the
// address that call returns for the current device, never the name itself.
static __device__ __align__(256) double scratch[AFIV_STATIC_SCRATCH_DOUBLES];

namespace {

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Device address of `scratch` on each GPU, resolved when the first grid is created on that GPU (and
// again on demand by afiv_surface_static, should the table be empty).  Racing first calls resolve
// the same address twice and store the same value.
constexpr int MAX_DEVICES = 64;
std::atomic<double *> scratch_addr[MAX_DEVICES];

int scratch_for_current_device(double **out)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return AFIV_E_HIP;
    if (dev < 0 || dev >= MAX_DEVICES) return AFIV_E_ARG;
    double *p = scratch_addr[dev].load(std::memory_order_acquire);
    if (!p) {
        if (hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return AFIV_E_HIP;
        scratch_addr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return AFIV_OK;
}

// Owns a grid under construction: whatever it holds is freed on every way out of afiv_grid_create
// (error returns, a failed host allocation) except the successful one, which releases it.
struct GridHold {
    afiv_grid *g = nullptr;
    ~GridHold()
    {
        if (!g) return;
        if (g->workspace) (void)hipFree(g->workspace);
        if (g->block) (void)hipFree(g->block);
        delete g;
    }
    afiv_grid *release()
    {
        afiv_grid *r = g;
        g = nullptr;
        return r;
    }
};

int check_grid(const afiv_grid *g)
{
    if (!g) return AFIV_E_NULL;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return AFIV_E_HIP;
    if (dev != g->device) return AFIV_E_ARG;   // a grid is bound to the device it was created on
    return AFIV_OK;
}

}  // namespace

extern "C" {

int afiv_version(void) { return AFIV_VERSION; }

const char *afiv_error_string(int code)
{
    switch (code) {
        case AFIV_OK: return "ok";
        case AFIV_E_NULL: return "required pointer is NULL";
        case AFIV_E_SHAPE: return "size out of supported range";
        case AFIV_E_HIP: return "HIP runtime call, kernel launch or host allocation failed";
        case AFIV_E_ARG: return "bad argument (enum value, lattice value that is not finite, or object used on a different device)";
        case AFIV_E_WORKSPACE: return "grid was created without the workspace this call needs";
        case AFIV_E_SCRATCH: return "batch exceeds what afiv_surface_static passes through its static scratch "
                                    "in one call (AFIV_STATIC_MAX_PIECES pieces of AFIV_STATIC_SCRATCH_DOUBLES / "
                                    "(nT * nK) surfaces)";
        default: return "unknown error";
    }
}

int afiv_grid_create(afiv_grid **out, const double *logm, int nK, const double *tau, int nT, double r,
                     double q, int trunc_days, int64_t workspace_surfaces)
try {
    if (!out || !logm || !tau) return AFIV_E_NULL;
    *out = nullptr;
    if (nK < 1 || nK > AFIV_MAX_STRIKES || nT < 1 || nT > AFIV_MAX_MATURITIES) return AFIV_E_SHAPE;
    if (workspace_surfaces < 0) return AFIV_E_SHAPE;
    // year fractions are turned into whole days below: finite, and within the range of that conversion
    if (!std::isfinite(r) || !std::isfinite(q)) return AFIV_E_ARG;
    for (int k = 0; k < nK; ++k)
        if (!std::isfinite(logm[k])) return AFIV_E_ARG;
    for (int t = 0; t < nT; ++t)
        if (!std::isfinite(tau[t]) || std::fabs(tau[t]) > 1e6) return AFIV_E_ARG;
    const int nKpad = (nK + 15) / 16 * 16;

    // Host-side constants use the host libm so that discount factors, forwards and strikes are
    // the very doubles NumPy produces in the Python reference.
    // strikes symmetric about their midpoint (to a few ulp) -- any linspace(a, b, n) is: half of them
    // carry the contraction.  A midpoint other than 0 is a phase x_n c of the integrand, folded into
    // the per-maturity drift term below.
    const int nH = (nK + 1) / 2, nHpad = (nH + 15) / 16 * 16;
    bool sym = nK >= 4;
    double scale = 0.0;
    for (int k = 0; k < nK; ++k) scale = std::fmax(scale, std::fabs(logm[k]));
    double center = 0.5 * (logm[0] + logm[nK - 1]);
    if (!(std::fabs(center) > 8.9e-16 * scale)) center = 0.0;
    for (int k = 0; k < nK && sym; ++k)
        if (!(std::fabs((logm[k] - center) + (logm[nK - 1 - k] - center)) <= 8.9e-16 * scale)) sym = false;
    if (!sym) center = 0.0;
    const size_t n_sym = sym ? (size_t)2 * AFIV_NQ_NODES * nHpad : 0;
    const size_t n_const = ((size_t)8 * nT + (size_t)2 * nK + (size_t)AFIV_KDIM * nKpad + n_sym + 3) / 4 * 4;
    const size_t n_order = ((size_t)nT * nK * sizeof(unsigned short) + 31) / 32 * 4;   // in doubles, 32-byte multiple
    const size_t n_prog = AFIV_MAX_MATURITIES / 8 + AFIV_MAX_MATURITIES;               // one byte per maturity, then one word
    const size_t nd = n_const + n_order + n_prog;
    std::vector<double> h(nd, 0.0);
    double *p_tau = h.data(), *p_term = p_tau + nT, *p_dd = p_term + nT, *p_Dq = p_dd + nT,
           *p_Dr = p_Dq + nT, *p_disc = p_Dr + nT, *p_fwd = p_disc + nT, *p_dq = p_fwd + nT,
           *p_logm = p_dq + nT, *p_K = p_logm + nK, *p_tab = p_K + nK,
           *p_sym = p_tab + (size_t)AFIV_KDIM * nKpad;
    for (int t = 0; t < nT; ++t) {
        p_tau[t] = tau[t];
        // heston_model_ql.py:131: int(tau*365) days, Actual/365 Fixed
        p_term[t] = trunc_days ? (double)((long)(tau[t] * 365.0)) / 365.0 : tau[t];
        p_Dr[t] = 1.0 / std::exp(r * p_term[t]);
        p_Dq[t] = 1.0 / std::exp(q * p_term[t]);
        // (minus the strike grid's midpoint when the half-strike contraction is used: the pricing
        // kernel is this array's only reader)
        p_dd[t] = std::log(1.0) - std::log(p_Dr[t] / p_Dq[t]) - center;
        p_disc[t] = std::exp(-r * tau[t]);
        p_fwd[t] = std::exp((r - q) * tau[t]);
        p_dq[t] = std::exp(-q * tau[t]);
    }
    for (int k = 0; k < nK; ++k) {
        p_logm[k] = logm[k];
        p_K[k] = 1.0 * std::exp(logm[k]);       // heston_model_ql.py:171 with spot 1
    }
    // Strike tables for the contraction index m = 2*(j*144 + n) + c (c = 0: sin-alpha operand,
    // c = 1: cos-alpha operand).  Row values per strike k, with sx = log K_k:
    //   j = 0 (P1):   cos(x_n sx),      -sin(x_n sx)
    //   j = 1 (P2): -K cos(x_n sx),    K sin(x_n sx)
    // so that sum_m X[m] tab[m][k] = Dq P1 - K Dr P2 when X carries Dq / Dr.
    for (int k = 0; k < nK; ++k) {
        const double K = p_K[k];
        const double sx = std::log(K);
        for (int n = 0; n < AFIV_NQ_NODES; ++n) {
            const double ang = AFIV_LAG_X[n] * sx;
            const double c = std::cos(ang), s = std::sin(ang);
            p_tab[(size_t)(2 * n + 0) * nKpad + k] = c;
            p_tab[(size_t)(2 * n + 1) * nKpad + k] = -s;
            p_tab[(size_t)(2 * (AFIV_NQ_NODES + n) + 0) * nKpad + k] = -K * c;
            p_tab[(size_t)(2 * (AFIV_NQ_NODES + n) + 1) * nKpad + k] = K * s;
        }
    }

    if (sym) {
        for (int k = 0; k < nH; ++k) {
            const double sx = std::log(p_K[k]) - center;
            for (int n = 0; n < AFIV_NQ_NODES; ++n) {
                const double ang = AFIV_LAG_X[n] * sx;
                p_sym[(size_t)n * nHpad + k] = std::cos(ang);
                p_sym[(size_t)(AFIV_NQ_NODES + n) * nHpad + k] = std::sin(ang);
            }
        }
    }

    {
        // service order of the implied-vol kernel's queue
        const int per = nT * nK;
        std::vector<double> score(per);
        std::vector<unsigned short> idx(per);
        for (int t = 0; t < nT; ++t)
            for (int k = 0; k < nK; ++k) {
                score[t * nK + k] = tau[t] > 0.0 ? std::fabs((r - q) * tau[t] - logm[k]) / std::sqrt(tau[t]) : 0.0;
                idx[t * nK + k] = (unsigned short)(t * nK + k);
            }
        std::stable_sort(idx.begin(), idx.end(),
                         [&](unsigned short a, unsigned short b) { return score[a] > score[b]; });
        std::memcpy(h.data() + n_const, idx.data(), per * sizeof(unsigned short));
    }

    GridHold hold;
    hold.g = new (std::nothrow) afiv_grid();
    afiv_grid *const g = hold.g;
    if (!g) return AFIV_E_HIP;
    g->block = nullptr;
    g->workspace = nullptr;
    g->workspace_surfaces = 0;
    if (hipGetDevice(&g->device) != hipSuccess) return AFIV_E_HIP;
    {
        // Resolve this device's static scratch here, at set-up: afiv_surface_static needs a grid of
        // the current device, so its own look-up then never runs first inside a stream capture.
        double *unused = nullptr;
        const int rc = scratch_for_current_device(&unused);
        if (rc) return rc;
    }
    double *dev = nullptr;
    if (hipMalloc((void **)&dev, nd * sizeof(double)) != hipSuccess) return AFIV_E_HIP;
    g->block = dev;
    if (hipMemcpy(dev, h.data(), nd * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return AFIV_E_HIP;
    if (workspace_surfaces > 0) {
        const size_t bytes = (size_t)workspace_surfaces * nT * nK * sizeof(double);
        if (hipMalloc((void **)&g->workspace, bytes) != hipSuccess) {
            g->workspace = nullptr;
            return AFIV_E_HIP;
        }
        g->workspace_surfaces = workspace_surfaces;
    }
    GridView &v = g->v;
    v.nT = nT; v.nK = nK; v.nKpad = nKpad; v.r = r; v.q = q;
    v.tau = dev; v.term = v.tau + nT; v.dd = v.term + nT; v.Dq = v.dd + nT; v.Dr = v.Dq + nT;
    v.disc_tau = v.Dr + nT; v.fwd_mult = v.disc_tau + nT; v.dq_tau = v.fwd_mult + nT;
    v.logm = v.dq_tau + nT; v.Kmult = v.logm + nK; v.tab = v.Kmult + nK;
    v.sym = sym ? 1 : 0;
    v.nH = nH;
    v.nHpad = nHpad;
    v.tab_sym = v.tab + (size_t)AFIV_KDIM * nKpad;
    v.order = reinterpret_cast<const unsigned short *>(dev + n_const);
    // whole-day year fractions with at most two distinct increments: see price_kernel's NFAC
    v.step_count = 0;
    v.step_dt[0] = v.step_dt[1] = 0.0;
    v.step_sel = 0;
    if (trunc_days && nT >= 4 && nT <= 32) {
        long lo = 0, hi = 0;
        bool ok = (long)(tau[0] * 365.0) >= 0;
        for (int t = 1; t < nT && ok; ++t) {
            const long d = (long)(tau[t] * 365.0) - (long)(tau[t - 1] * 365.0);
            if (d <= 0) ok = false;
            if (t == 1 || d < lo) lo = d;
            if (t == 1 || d > hi) hi = d;
        }
        if (ok && hi - lo <= 3) {
            // base step + k extra days, k = 0..3.  When only two step lengths occur and the longer one
            // is the more frequent (round(linspace(...)) days: mostly 37, some 36), take the longer as
            // the base and MINUS one day as the extra: fewer extra factors to multiply in.
            long n_lo = 0;
            for (int t = 1; t < nT; ++t)
                if ((long)(tau[t] * 365.0) - (long)(tau[t - 1] * 365.0) == lo) ++n_lo;
            const bool down = hi - lo == 1 && n_lo < (nT - 1) - n_lo;
            const long base = down ? hi : lo;
            unsigned long long sel = 0;
            for (int t = 1; t < nT; ++t) {
                const long d = (long)(tau[t] * 365.0) - (long)(tau[t - 1] * 365.0);
                sel |= (unsigned long long)(down ? base - d : d - base) << (2 * t);
            }
            v.step_count = hi == lo ? 1 : 2;
            v.step_dt[0] = (double)base / 365.0;
            v.step_dt[1] = (down ? -1.0 : 1.0) / 365.0;
            v.step_sel = sel;
        }
    }
    unsigned char prog[AFIV_MAX_MATURITIES] = {0};
    unsigned long long need[AFIV_MAX_MATURITIES] = {0};
    v.prog = reinterpret_cast<const unsigned char *>(dev + n_const + n_order);
    v.need = reinterpret_cast<const unsigned long long *>(dev + n_const + n_order + AFIV_MAX_MATURITIES / 8);
    if (trunc_days && v.step_count == 0 && nT >= 2) {
        // Maturities that are sums of earlier ones: 1 - exp(-d (Tx + Ty)) = ox + oy - ox oy costs six
        // instructions where a direct evaluation costs an exp and a sincos.  A thread can keep the
        // previous maturity's value (P) and two saved ones (A, B); which values to save is chosen
        // here, by dynamic programming over (A, B) = indices of the saved maturities, so that the
        // number of direct evaluations is smallest.  Exact in whole days, so the result does not
        // depend on rounding of the year fractions.
        std::vector<long> days(nT);
        bool ok = true;
        for (int t = 0; t < nT; ++t) {
            days[t] = (long)(tau[t] * 365.0);
            if (days[t] <= 0) ok = false;            // expired or same-day maturities: leave the lattice to the direct path
        }
        if (ok) {
            const int S = nT + 1;                      // slot states: 0 = empty, i + 1 = holds maturity i
            const int INF = 1 << 20;
            std::vector<int> cost((size_t)(nT + 1) * S * S, INF);
            std::vector<unsigned char> how((size_t)(nT + 1) * S * S, 0);
            std::vector<int> from((size_t)(nT + 1) * S * S, -1);
            auto at = [&](int t, int a, int b) { return ((size_t)t * S + a) * S + b; };
            cost[at(0, 0, 0)] = 0;
            for (int t = 0; t < nT; ++t)
                for (int a = 0; a < S; ++a)
                    for (int b = 0; b < S; ++b) {
                        const int c0 = cost[at(t, a, b)];
                        if (c0 >= INF) continue;
                        const long vp = t > 0 ? days[t - 1] : -1, va = a ? days[a - 1] : -1, vb = b ? days[b - 1] : -1;
                        int op = 0;
                        const long w = days[t];
                        if (vp > 0 && vp + vp == w) op = 1;
                        else if (vp > 0 && va > 0 && vp + va == w) op = 2;
                        else if (vp > 0 && vb > 0 && vp + vb == w) op = 3;
                        else if (va > 0 && va + va == w) op = 4;
                        else if (va > 0 && vb > 0 && va + vb == w) op = 5;
                        else if (vb > 0 && vb + vb == w) op = 6;
                        const int c1 = c0 + (op == 0 ? 1 : 0);
                        for (int save = 0; save < 3; ++save) {
                            const int na = save == 1 ? t + 1 : a, nb = save == 2 ? t + 1 : b;
                            const size_t j = at(t + 1, na, nb);
                            if (c1 < cost[j]) {
                                cost[j] = c1;
                                how[j] = (unsigned char)(op | (save << 4));
                                from[j] = a * S + b;
                            }
                        }
                    }
            int best = INF, ba = 0, bb = 0;
            for (int a = 0; a < S; ++a)
                for (int b = 0; b < S; ++b)
                    if (cost[at(nT, a, b)] < best) { best = cost[at(nT, a, b)]; ba = a; bb = b; }
            if (best < INF) {
                int a = ba, b = bb;
                for (int t = nT; t > 0; --t) {
                    const size_t j = at(t, a, b);
                    prog[t - 1] = how[j];
                    const int f = from[j];
                    a = f / S;
                    b = f % S;
                }
                // A product roughly adds the relative errors of its two factors exp(-d T), so a chain of
                // k doublings carries 2^k roundings: values deeper than five products below a direct
                // evaluation (32 roundings, 4e-15) are evaluated directly instead.  Turning a step into
                // a direct evaluation leaves every later step valid (they use its value, not its origin).
                std::vector<int> depth(nT, 0);
                int la = -1, lb = -1;
                best = 0;
                for (int t = 0; t < nT; ++t) {
                    const int op = prog[t] & 15, pv = t - 1;
                    const int pick[7][2] = {{-1, -1}, {pv, pv}, {pv, la}, {pv, lb}, {la, la}, {la, lb}, {lb, lb}};
                    if (op) {
                        depth[t] = 1 + std::max(depth[pick[op][0]], depth[pick[op][1]]);
                        if (depth[t] > 5) {
                            prog[t] &= 0xf0;
                            depth[t] = 0;
                        }
                    }
                    if (!(prog[t] & 15)) ++best;
                    if ((prog[t] >> 4) == 1) la = t;
                    if ((prog[t] >> 4) == 2) lb = t;
                }
            }
            // worth a kernel of its own when at least a quarter of the direct evaluations go
            if (best < INF && 4 * best <= 3 * nT) {
                // Which earlier values does a walk that starts at maturity t0 descend from?  Sources of
                // every step in the full walk, then the transitive closure below t0.
                {
                    std::vector<int> src0(nT, -1), src1(nT, -1);
                    int la = -1, lb = -1;                 // maturities whose values sit in A and B
                    for (int t = 0; t < nT; ++t) {
                        const int op = prog[t] & 15, pv = t - 1;
                        const int pick[7][2] = {{-1, -1}, {pv, pv}, {pv, la}, {pv, lb}, {la, la}, {la, lb}, {lb, lb}};
                        src0[t] = pick[op][0];
                        src1[t] = pick[op][1];
                        if ((prog[t] >> 4) == 1) la = t;
                        if ((prog[t] >> 4) == 2) lb = t;
                    }
                    for (int t0 = 1; t0 < nT; ++t0) {
                        unsigned long long mask = 0;
                        for (int t = t0; t < nT; ++t)
                            for (int sidx : {src0[t], src1[t]})
                                if (sidx >= 0 && sidx < t0) mask |= 1ull << sidx;
                        for (int t = t0 - 1; t >= 0; --t)          // sources lie below their users
                            if ((mask >> t) & 1ull)
                                for (int sidx : {src0[t], src1[t]})
                                    if (sidx >= 0) mask |= 1ull << sidx;
                        need[t0] = mask;
                    }
                }
                // (the block is already on the device: these small pieces follow it)
                if (hipMemcpy(dev + n_const + n_order, prog, sizeof(prog), hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(dev + n_const + n_order + AFIV_MAX_MATURITIES / 8, need, sizeof(need),
                              hipMemcpyHostToDevice) != hipSuccess)
                    return AFIV_E_HIP;
                v.step_count = 3;
            }
        }
    }
    if (!trunc_days && nT >= 4 && nT <= 32) {
        // exact year fractions, evenly spaced to a few ulp (linspace): one step factor
        const double step = (tau[nT - 1] - tau[0]) / (double)(nT - 1);
        bool ok = tau[0] >= 0.0 && step > 0.0;
        for (int t = 1; t < nT && ok; ++t)
            if (!(std::fabs((tau[t] - tau[t - 1]) - step) <= 8.9e-16 * std::fabs(tau[nT - 1]))) ok = false;
        if (ok) {
            v.step_count = 1;
            v.step_dt[0] = step;
        }
    }
    *out = hold.release();
    return AFIV_OK;
} catch (...) {
    return AFIV_E_HIP;   // a host allocation failed; what the grid held by then went with its holder
}

int afiv_grid_destroy(afiv_grid *g)
{
    if (!g) return AFIV_OK;
    if (g->workspace) (void)hipFree(g->workspace);
    if (g->block) (void)hipFree(g->block);
    delete g;
    return AFIV_OK;
}

int afiv_grid_traits(const afiv_grid *g, int *half_strike, int *stepped)
{
    if (!g) return AFIV_E_NULL;
    if (half_strike) *half_strike = g->v.sym;
    if (stepped) *stepped = g->v.step_count;
    return AFIV_OK;
}

int afiv_price(const afiv_grid *g, const double *params, int64_t N, double *ratio_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (N == 0) return AFIV_OK;
    if (!params || !ratio_out) return AFIV_E_NULL;
    return afiv_launch_price(g->v, params, N, ratio_out, as_stream(stream));
}

int afiv_implied_vol(const afiv_grid *g, const double *ratio, int64_t N, double spot, int variant,
                     double *iv_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (variant < AFIV_VARIANT_A || variant > AFIV_VARIANT_A_BRENT) return AFIV_E_ARG;
    if (N == 0) return AFIV_OK;
    if (!ratio || !iv_out) return AFIV_E_NULL;
    return afiv_launch_implied_vol(g->v, ratio, N, spot, variant, iv_out, as_stream(stream));
}

int afiv_arb_check(const afiv_grid *g, const double *iv, const double *ratio, int64_t N, double spot,
                   int bs_form, int32_t *flags_out, uint8_t *mask_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (bs_form != AFIV_VARIANT_A && bs_form != AFIV_VARIANT_B) return AFIV_E_ARG;
    if (N == 0) return AFIV_OK;
    if (!iv || !flags_out) return AFIV_E_NULL;
    return afiv_launch_arb_check(g->v, iv, ratio, N, spot, bs_form, flags_out, mask_out, as_stream(stream));
}

int afiv_implied_vol_checked(const afiv_grid *g, const double *ratio, int64_t N, double spot, int variant,
                             double *iv_out, int32_t *flags_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (variant < AFIV_VARIANT_A || variant > AFIV_VARIANT_A_BRENT) return AFIV_E_ARG;
    if (N == 0) return AFIV_OK;
    if (!ratio || !iv_out || !flags_out) return AFIV_E_NULL;
    return afiv_launch_surface_iv(g->v, ratio, N, spot, variant, iv_out, flags_out, as_stream(stream));
}

int afiv_surface(const afiv_grid *g, const double *params, int64_t N, double spot, int variant,
                 double *ratio_out, double *iv_out, int32_t *flags_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (variant < AFIV_VARIANT_A || variant > AFIV_VARIANT_A_BRENT) return AFIV_E_ARG;
    if (N == 0) return AFIV_OK;
    if (!params || !iv_out || !flags_out) return AFIV_E_NULL;
    hipStream_t s = as_stream(stream);
    const int64_t per = (int64_t)g->v.nT * g->v.nK;
    if (ratio_out) {
        if ((rc = afiv_launch_price(g->v, params, N, ratio_out, s))) return rc;
        return afiv_launch_surface_iv(g->v, ratio_out, N, spot, variant, iv_out, flags_out, s);
    }
    if (!g->workspace) return AFIV_E_WORKSPACE;
    // Stream order makes the reuse of the workspace by consecutive chunks safe.
    for (int64_t off = 0; off < N; off += g->workspace_surfaces) {
        const int64_t n = (N - off < g->workspace_surfaces) ? N - off : g->workspace_surfaces;
        if ((rc = afiv_launch_price(g->v, params + 5 * off, n, g->workspace, s))) return rc;
        if ((rc = afiv_launch_surface_iv(g->v, g->workspace, n, spot, variant, iv_out + off * per,
                                         flags_out + 4 * off, s))) return rc;
    }
    return AFIV_OK;
}

int afiv_surface_static(const afiv_grid *g, const double *params, int64_t N, double spot, int variant,
                        double *iv_out, int32_t *flags_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (N < 0) return AFIV_E_SHAPE;
    if (variant < AFIV_VARIANT_A || variant > AFIV_VARIANT_A_BRENT) return AFIV_E_ARG;
    if (N == 0) return AFIV_OK;
    if (!params || !iv_out || !flags_out) return AFIV_E_NULL;
    const int64_t per = (int64_t)g->v.nT * g->v.nK;
    // Batches beyond the array go through it in pieces, one pair of launches each.  A piece is a
    // multiple of 7 x 256 surfaces where the array allows (the pricing kernel's workgroups take seven
    // surfaces at a time and one workgroup fits a CU, so every CU gets the same share), and the
    // pieces are evened out so that the last one is not a sliver.
    const int64_t fit = (int64_t)AFIV_STATIC_SCRATCH_DOUBLES / per, wave = 7 * 256;
    int64_t piece = fit >= wave ? fit / wave * wave : fit;
    const int64_t pieces = (N + piece - 1) / piece;
    if (pieces > AFIV_STATIC_MAX_PIECES) return AFIV_E_SCRATCH;
    if (pieces > 1 && piece >= wave) {
        const int64_t even = ((N + pieces - 1) / pieces + wave - 1) / wave * wave;
        if (even < piece) piece = even;
    }
    double *ratio = nullptr;
    if ((rc = scratch_for_current_device(&ratio))) return rc;
    hipStream_t s = as_stream(stream);
    // Every ratio the second kernel reads has been written by the first: nothing to clear.  Stream
    // order makes the reuse of the array by consecutive pieces safe.
    for (int64_t off = 0; off < N; off += piece) {
        const int64_t n = N - off < piece ? N - off : piece;
        if ((rc = afiv_launch_price(g->v, params + 5 * off, n, ratio, s))) return rc;
        if ((rc = afiv_launch_surface_iv(g->v, ratio, n, spot, variant, iv_out + off * per, flags_out + 4 * off, s)))
            return rc;
    }
    return AFIV_OK;
}

int afiv_decoder_create(afiv_decoder **out, const afiv_decoder_weights *w)
try {
    if (!out || !w) return AFIV_E_NULL;
    *out = nullptr;
    static const int IN[4] = {12, 32, 64, 128}, OUT[4] = {32, 64, 128, 5};
    for (int l = 0; l < 4; ++l)
        if (!w->lin_w[l] || !w->lin_b[l]) return AFIV_E_NULL;
    for (int l = 0; l < 3; ++l)
        if (!w->bn_w[l] || !w->bn_b[l] || !w->bn_mean[l] || !w->bn_var[l]) return AFIV_E_NULL;
    if (!w->param_mean || !w->param_std) return AFIV_E_NULL;
    std::vector<float> h;
    for (int l = 0; l < 4; ++l) {
        const int in = IN[l], o = OUT[l];
        for (int i = 0; i < in; ++i)
            for (int j = 0; j < o; ++j) h.push_back(w->lin_w[l][(size_t)j * in + i]);   // transpose to input-major
        for (int j = 0; j < o; ++j) h.push_back(w->lin_b[l][j]);
        if (l < 3) {
            // eval-mode BatchNorm1d as y = x*scale + shift (float arithmetic, as torch's CPU kernel)
            std::vector<float> scale(o), shift(o);
            for (int j = 0; j < o; ++j) {
                const float invstd = 1.0f / std::sqrt(w->bn_var[l][j] + w->bn_eps);
                scale[j] = invstd * w->bn_w[l][j];
                shift[j] = w->bn_b[l][j] - w->bn_mean[l][j] * scale[j];
            }
            h.insert(h.end(), scale.begin(), scale.end());
            h.insert(h.end(), shift.begin(), shift.end());
        }
    }
    const size_t npacked = h.size();
    for (int j = 0; j < 5; ++j) h.push_back(w->param_mean[j]);
    for (int j = 0; j < 5; ++j) h.push_back(w->param_std[j]);
    afiv_decoder *d = new (std::nothrow) afiv_decoder();
    if (!d) return AFIV_E_HIP;
    if (hipGetDevice(&d->device) != hipSuccess) { delete d; return AFIV_E_HIP; }
    float *dev = nullptr;
    if (hipMalloc((void **)&dev, h.size() * sizeof(float)) != hipSuccess) { delete d; return AFIV_E_HIP; }
    if (hipMemcpy(dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(dev); delete d; return AFIV_E_HIP;
    }
    d->block = dev;
    d->v.packed = dev;
    d->v.pmean = dev + npacked;
    d->v.pstd = dev + npacked + 5;
    *out = d;
    return AFIV_OK;
} catch (...) {
    return AFIV_E_HIP;   // host allocation of the packed weights failed; no device memory is held yet
}

int afiv_decoder_destroy(afiv_decoder *d)
{
    if (!d) return AFIV_OK;
    if (d->block) (void)hipFree(d->block);
    delete d;
    return AFIV_OK;
}

int afiv_decode(const afiv_decoder *d, const float *z, const float *c, int64_t N, int64_t c_rows,
                float *normalized_out, float *params32_out, double *params64_out, void *stream)
{
    if (!d) return AFIV_E_NULL;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return AFIV_E_HIP;
    if (dev != d->device) return AFIV_E_ARG;
    if (N < 0) return AFIV_E_SHAPE;
    if (N == 0) return AFIV_OK;
    if (!z || !c) return AFIV_E_NULL;
    if (c_rows != 1 && c_rows != N) return AFIV_E_SHAPE;
    if (!normalized_out && !params32_out && !params64_out) return AFIV_E_NULL;
    return afiv_launch_decode(d->v, z, c, N, c_rows, normalized_out, params32_out, params64_out,
                              as_stream(stream));
}

int afiv_arbitrage_penalty(const afiv_grid *g, const double *params, int64_t B, double *scratch,
                           double *acc_out, void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (B < 0) return AFIV_E_SHAPE;
    if (!acc_out) return AFIV_E_NULL;
    if (B > 0 && (!params || !scratch)) return AFIV_E_NULL;
    hipStream_t s = as_stream(stream);
    if (B > 0 && (rc = afiv_launch_price(g->v, params, B, scratch, s))) return rc;
    return afiv_launch_penalty(g->v, scratch, B, acc_out, s);
}

int afiv_stage1_objective(const afiv_grid *g, const double *model_ratio, const double *params, int64_t C,
                          const double *market_ratio, int64_t D, double drift_weight, double *loss_out,
                          void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (C < 0 || D < 0) return AFIV_E_SHAPE;
    if (C == 0 || D == 0) return AFIV_OK;
    if (!model_ratio || !params || !market_ratio || !loss_out) return AFIV_E_NULL;
    return afiv_launch_stage1(g->v, model_ratio, params, C, market_ratio, D, drift_weight, loss_out,
                              as_stream(stream));
}

int afiv_density_wasserstein(const afiv_grid *g, const double *model_ratio, int64_t C,
                             const double *market_ratio, int64_t D, double *scratch, double *dist_out,
                             void *stream)
{
    int rc = check_grid(g);
    if (rc) return rc;
    if (C < 0 || D < 0 || g->v.nK < 3) return AFIV_E_SHAPE;
    if (C == 0 || D == 0) return AFIV_OK;
    if (!model_ratio || !market_ratio || !scratch || !dist_out) return AFIV_E_NULL;
    return afiv_launch_density_wasserstein(g->v, model_ratio, C, market_ratio, D, scratch, dist_out,
                                           as_stream(stream));
}

int afiv_ensemble_stats(const double *iv, const uint8_t *select, int64_t N, int64_t points,
                        double *stats_out, void *stream)
{
    if (!iv || !stats_out) return AFIV_E_NULL;
    if (N < 1 || N > (int64_t)1 << 31 || points < 1 || points > (1 << 24)) return AFIV_E_SHAPE;
    return afiv_launch_ensemble(iv, select, N, (int)points, 1, stats_out, as_stream(stream));
}

int afiv_ensemble_stats_groups(const double *iv, const uint8_t *select, int64_t groups, int64_t N, int64_t points,
                               double *stats_out, void *stream)
{
    if (!iv || !stats_out) return AFIV_E_NULL;
    if (groups < 0 || N < 1 || N > (int64_t)1 << 31 || points < 1 || points > (1 << 24)) return AFIV_E_SHAPE;
    if (groups == 0) return AFIV_OK;
    return afiv_launch_ensemble(iv, select, N, (int)points, groups, stats_out, as_stream(stream));
}

int afiv_debug_math(int op, const double *x, const double *y, int64_t n, double *out0, double *out1,
                    void *stream)
{
    if (n < 0) return AFIV_E_SHAPE;
    if (op < 0 || op > AFIV_MATH_EXPM1_TAB) return AFIV_E_ARG;
    if (n == 0) return AFIV_OK;
    if (!x || !out0) return AFIV_E_NULL;
    if (op == AFIV_MATH_ATAN2 && !y) return AFIV_E_NULL;
    if (op == AFIV_MATH_NDTR) return afiv_launch_debug_ndtr(x, n, out0, as_stream(stream));
    return afiv_launch_debug_math(op, x, y, n, out0, out1, as_stream(stream));
}

}  // extern "C"
