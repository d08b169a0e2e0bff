#!/usr/bin/env python3
"""Kernel times of the two stages on the reference's 8 x 21 lattice and the 20 x 50 benchmark
lattice (diagnostic)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
afiv = afiv_bootstrap.load()
from test_gpu_parity import sample_params, T_GRID, LOGM_GRID, SPOT, R, Q, dev

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

grids = {"8x21": (LOGM_GRID, T_GRID, 131072),
         "20x50": (np.linspace(-0.2, 0.2, 50), np.round(np.linspace(30, 730, 20)) / 365.0, 131072)}
for name, (logm, tau, n) in grids.items():
    g = afiv.Grid(logm, tau, R, Q)
    p = dev(sample_params(n, 3))
    ratio = g.price(p)
    iv, fl = g.implied_vol_checked(ratio, SPOT, "A")
    tp = timeit(lambda: g.price(p, out=ratio))
    ti = timeit(lambda: g.implied_vol_checked(ratio, SPOT, "A", iv_out=iv, flags_out=fl))
    pts = len(logm) * len(tau)
    print(f"{name}: price {tp:.2f} ms ({tp*1e6/n/len(tau):.1f} ns per maturity), iv+checks {ti:.2f} ms "
          f"({ti*1e6/n/pts:.2f} ns per point), {n/(tp+ti)*1e3/1e6:.2f} M surfaces/s", flush=True)

# small batches on the reference's lattice: latency of the whole path (one afiv_surface call)
g = afiv.Grid(LOGM_GRID, T_GRID, R, Q)
for n in (1, 100, 1000, 10000):
    p = dev(sample_params(n, 5))
    t = timeit(lambda: g.surface(p, SPOT, "A"), n=20)
    print(f"8x21, N={n}: {t*1e3:.0f} us per call ({n/t*1e3:.0f} surfaces/s)", flush=True)
