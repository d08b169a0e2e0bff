#!/usr/bin/env python3
"""Pricing-kernel time on 50 strikes: centred linspace, off-centre linspace (both take the
half-strike contraction) and an unevenly spaced grid (general contraction) (diagnostic)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
afiv = afiv_bootstrap.load()
import bench
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
_, tau = bench.grids()
p = bench.synthetic_params(131072, 1, "cuda")
u = np.linspace(0, 1, 50)
for name, logm in (("centred linspace", np.linspace(-0.2, 0.2, 50)), ("off-centre linspace", np.linspace(-0.3, 0.2, 50)),
                   ("uneven", -0.3 + 0.5 * (0.7 * u + 0.3 * u * u))):
    g = afiv.Grid(logm, tau, bench.R, bench.Q)
    out = g.price(p)
    print(name, g.traits, f"{timeit(lambda: g.price(p, out=out)):.2f} ms", flush=True)
