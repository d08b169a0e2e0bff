#!/usr/bin/env python3
"""Prints parity statistics and rough kernel timings on the GPU box (diagnostic, not a test)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap, oracle
afiv = afiv_bootstrap.load()
from test_gpu_parity import sample_params, T_GRID, LOGM_GRID, SPOT, R, Q, dev, price_tol

def stats(name, got, ref):
    err = np.abs(got - ref); ex = err / price_tol(ref)
    print(f"{name}: max abs {np.nanmax(err):.3e}  max err/tol {np.nanmax(ex):.3e}  frac>tol {np.mean(ex>1):.3e}", flush=True)

print(torch.cuda.get_device_name(0), flush=True)
grid = afiv.Grid(LOGM_GRID, T_GRID, R, Q)
g = np.load(os.path.join(ROOT, "tests/golden/golden_prices.npz"))
got = grid.price(dev(g["params"])).cpu().numpy()
stats("golden 8x21", got, g["ratio"])
i = np.unravel_index(np.nanargmax(np.abs(got-g["ratio"])/price_tol(g["ratio"])), got.shape)
print("  worst at", i, "params", g["params"][i[0]], "got", got[i], "ref", g["ratio"][i])
p = sample_params(256, 11)
got = grid.price(dev(p)).cpu().numpy(); ref = oracle.price_ratio_grid(p, LOGM_GRID, T_GRID, R, Q)
stats("random 8x21", got, ref)
for lo, hi in ((0, 1e-3), (1e-3, 5e-3), (5e-3, 2e-2), (2e-2, 10)):
    m = (p[:, 2] >= lo) & (p[:, 2] < hi)
    if m.any(): stats(f"  sigma in [{lo},{hi}) n={m.sum()}", got[m], ref[m])
logm = np.linspace(-0.2, 0.2, 50); tau = np.round(np.linspace(30, 730, 20)) / 365.0
grid3 = afiv.Grid(logm, tau, R, Q)
p3 = sample_params(64, 3)
got = grid3.price(dev(p3)).cpu().numpy(); ref = oracle.price_ratio_grid(p3, logm, tau, R, Q)
stats("random 20x50", got, ref)
# IV
gi = np.load(os.path.join(ROOT, "tests/golden/golden_iv.npz"))
for v, key in (("A", "iv_a"), ("B", "iv_b")):
    iv = grid.implied_vol(dev(gi["ratio"]), float(gi["spot"]), v).cpu().numpy()
    r = gi[key]; m = ~np.isnan(r) & ~np.isnan(iv)
    print(f"IV {v}: nan mismatch {np.sum(np.isnan(iv) != np.isnan(r))}  max abs {np.abs(iv[m]-r[m]).max():.3e}", flush=True)
# timing
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
for name, gr, N in (("8x21", grid, 65536), ("20x50", grid3, 16384)):
    P = dev(sample_params(N, 3))
    ratio = gr.price(P); iv = gr.implied_vol(ratio, SPOT, "A")
    tp = timeit(lambda: gr.price(P, out=ratio))
    ti = timeit(lambda: gr.implied_vol(ratio, SPOT, "A", out=iv))
    ta = timeit(lambda: gr.arb_check(iv, ratio, SPOT, "A"))
    print(f"{name} N={N}: price {tp*1e3:.2f} ms ({N/tp:.0f} surf/s)  iv {ti*1e3:.2f} ms  arb {ta*1e3:.2f} ms  total {N/(tp+ti+ta):.0f} surf/s", flush=True)
