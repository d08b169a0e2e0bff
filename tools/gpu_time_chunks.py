#!/usr/bin/env python3
"""Does the size of a launch matter (diagnostic)?  The benchmark batch, resident, processed as one
launch pair or in chunks whose price ratios fit the last-level cache, timed over enough repetitions
for the clocks to settle.

usage: gpu_time_chunks.py [surfaces] [seconds per setting]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import afiv_bootstrap
afiv = afiv_bootstrap.load()
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
logm, tau = bench.grids()
nT, nK = len(tau), len(logm)
dev = torch.device("cuda", 0)
grid = afiv.Grid(logm, tau, bench.R, bench.Q, device=dev)
p = bench.synthetic_params(n, 3, "cpu").contiguous().to(dev)
iv = torch.empty((n, nT, nK), dtype=torch.float64, device=dev)
fl = torch.empty((n, 4), dtype=torch.int32, device=dev)
for chunks in (1, 4, 8, 16, 32, 64, 1):
    rows = n // chunks
    ratio = torch.empty((rows, nT, nK), dtype=torch.float64, device=dev)

    def step():
        for c in range(chunks):
            sl = slice(c * rows, (c + 1) * rows)
            grid.price(p[sl], out=ratio)
            grid.implied_vol_checked(ratio, bench.SPOT, "A", iv_out=iv[sl], flags_out=fl[sl])
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 0; first = None
    while time.perf_counter() - t0 < budget:
        t1 = time.perf_counter()
        step(); torch.cuda.synchronize()
        if first is None:
            first = time.perf_counter() - t1
        reps += 1
    t = (time.perf_counter() - t0) / reps
    print(f"{chunks:3d} chunk(s) of {rows} surfaces ({rows*nT*nK*8/1e6:.0f} MB of ratios): {t*1e3:.2f} ms per step "
          f"over {reps} steps (first {first*1e3:.2f} ms) = {n/t/1e6:.2f} M surfaces/s", flush=True)
    del ratio

# the same batch through the entry points that keep no ratios: the library's 128 MiB static array
# (afiv_surface_static, pieces of 16128 surfaces) and a grid workspace of 131 072 surfaces
gw = afiv.Grid(logm, tau, bench.R, bench.Q, device=dev, workspace_surfaces=131072)
for name, fn in (("afiv_surface_static (128 MiB array)", lambda: grid.surface_static(p, bench.SPOT, "A", iv_out=iv, flags_out=fl)),
                 ("afiv_surface, workspace of 131072 surfaces", lambda: gw.surface(p, bench.SPOT, "A", keep_ratio=False, iv_out=iv, flags_out=fl))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < budget:
        fn(); torch.cuda.synchronize()
        reps += 1
    t = (time.perf_counter() - t0) / reps
    print(f"{name}: {t*1e3:.2f} ms per step over {reps} steps = {n/t/1e6:.2f} M surfaces/s", flush=True)
