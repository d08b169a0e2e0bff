#!/usr/bin/env python3
"""Times the pricing kernel alone (GPU box) on the reference's 8 x 21 lattice at the batch of 500
dates x 100 samples, on the 20 x 50 benchmark lattice and on an unevenly spaced 20 x 50 lattice.
usage: python tools/gpu_time_price_lattices.py [reps]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afiv_bootstrap
import bench

afiv = afiv_bootstrap.load()
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(5)
cases = (("8x21 x 50000", np.linspace(-0.2, 0.2, 21), np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0]), 50000),
         ("8x21 x 4096", np.linspace(-0.2, 0.2, 21), np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0]), 4096),
         ("20x50 x 131072",) + bench.grids() + (131072,),
         ("20x50 uneven x 131072", np.sort(rng.uniform(-0.2, 0.2, 50)), np.sort(rng.uniform(30, 730, 20)) / 365.0, 131072))
for name, lm, tau, n in cases:
    grid = afiv.Grid(lm, tau, bench.R, bench.Q, device=dev)
    p = bench.synthetic_params(n, 3, dev)
    out = torch.empty((n, len(tau), len(lm)), dtype=torch.float64, device=dev)
    for _ in range(5):
        grid.price(p, out=out)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(reps):
        grid.price(p, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    t = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)])
    fl = bench.pricing_flops(len(tau), len(lm)) * n
    print(f"{name}: median {np.median(t):.4f} ms, min {t.min():.4f} ms, "
          f"{fl / np.median(t) / 1e9:.1f} TFLOP/s by the contract's count = {fl / np.median(t) / 1e9 / bench.FP64_VECTOR_PEAK_TFLOPS:.3f} "
          f"of the FP64 peak; checksum {float(torch.nan_to_num(out).sum()):.12e}", flush=True)
