#!/usr/bin/env python3
"""Times the calibration loss-matrix kernel alone (GPU box): 4096 candidates x 500 days on the
reference's 8 x 21 lattice, and a 20 x 50 lattice.  usage: python tools/gpu_time_stage1.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afiv_bootstrap
import bench

afiv = afiv_bootstrap.load()
dev = torch.device("cuda", 0)
for name, lm, tau in (("8x21", np.linspace(-0.2, 0.2, 21), np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])),
                      ("20x50",) + bench.grids()):
    grid = afiv.Grid(lm, tau, bench.R, bench.Q, device=dev)
    cand = bench.synthetic_params(4096, 4, dev)
    model = grid.price(cand)
    market = grid.price(bench.synthetic_params(500, 5, dev))
    for label, mk in (("no holes", market), ("one NaN per 64 days", market.clone())):
        if label != "no holes":
            mk[::64, 0, 0] = float("nan")
        for _ in range(3):
            loss = grid.stage1_objective(model, cand, mk, 1.0)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(50):
            loss = grid.stage1_objective(model, cand, mk, 1.0)
        ev[1].record()
        torch.cuda.synchronize()
        print(f"{name} {label}: {ev[0].elapsed_time(ev[1]) / 50 * 1e3:.1f} us per 4096 x 500 loss matrix "
              f"(includes the output allocation); finite {float(torch.isfinite(loss).float().mean()):.3f}", flush=True)
