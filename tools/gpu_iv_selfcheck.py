#!/usr/bin/env python3
"""How close the forward-form kernel's vols are to the ROOT (diagnostic): every returned vol is
re-priced with SciPy's ndtr in NumPy and the residual divided by the vega, i.e. the Newton distance
to the root of the very equation the reference solves, on the benchmark lattice."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
afiv = afiv_bootstrap.load()
import bench
from test_gpu_parity import reprice_a, vega
logm, tau = bench.grids()
grid = afiv.Grid(logm, tau, bench.R, bench.Q)
n = 4096
p = bench.synthetic_params(n, 3, "cpu").cuda()
ratio = grid.price(p)
iv = grid.implied_vol(ratio, bench.SPOT, "A").cpu().numpy()
target = ratio.cpu().numpy() * bench.SPOT * np.exp(-bench.R * tau)[None, :, None]
price, unit = reprice_a(iv, logm, tau, bench.SPOT, bench.R, bench.Q)
v = vega(iv, logm, tau, bench.SPOT, bench.R, bench.Q)
ok = ~np.isnan(iv)
with np.errstate(all="ignore"):
    dist = np.abs(price - target) / v
for lo, hi in ((1e-3, np.inf), (1e-6, 1e-3), (0, 1e-6)):
    m = ok & (v >= lo) & (v < hi)
    if m.any():
        flat = np.abs(price - target)[m] <= unit[m]
        print(f"vega in [{lo:g},{hi:g}): n={m.sum()} Newton distance to the root: max {np.nanmax(np.where(flat, 0, dist[m])):.3g} "
              f"p99.9 {np.nanpercentile(dist[m], 99.9):.3g} median {np.nanmedian(dist[m]):.3g}; residual within the objective's rounding at {flat.mean()*100:.1f} %")
