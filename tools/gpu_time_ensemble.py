#!/usr/bin/env python3
"""Times afiv_ensemble_stats (GPU box): the one-sweep LDS path up to 16 384 surfaces and the
multi-sweep selection beyond, on 8 x 21 and 20 x 50 stacks.  usage: python tools/gpu_time_ensemble.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import afiv_bootstrap

afiv = afiv_bootstrap.load()
g = torch.Generator(device="cuda").manual_seed(1)
for n, shape in ((100, (8, 21)), (16384, (8, 21)), (50000, (8, 21)), (131072, (20, 50)), (1048576, (20, 50))):
    s = 0.2 + 0.05 * torch.randn((n,) + shape, generator=g, device="cuda", dtype=torch.float64)
    s[torch.rand(s.shape, generator=g, device="cuda") < 0.1] = float("nan")
    keep = torch.rand(n, generator=g, device="cuda") < 0.8
    for label, sel in (("all", None), ("mask", keep)):
        afiv.ensemble_stats(s, sel)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        reps = 3 if n > 200000 else 10
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            afiv.ensemble_stats(s, sel)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / reps
        gb = s.numel() * 8 / 1e9
        print(f"{n} x {shape[0]} x {shape[1]} ({label}): {ms:.3f} ms per call, {gb / ms * 1e3:.0f} GB/s of the stack's size "
              f"({gb:.2f} GB)", flush=True)
    del s

# the per-date statistics of the all-dates batch: 500 groups of 100 surfaces in one call
s = 0.2 + 0.05 * torch.randn((50000, 8, 21), generator=g, device="cuda", dtype=torch.float64)
s[torch.rand(s.shape, generator=g, device="cuda") < 0.1] = float("nan")
keep = torch.rand(50000, generator=g, device="cuda") < 0.8
for label, sel in (("all", None), ("mask", keep)):
    afiv.ensemble_stats(s, sel, groups=500)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(10):
        afiv.ensemble_stats(s, sel, groups=500)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"500 groups of 100 x 8 x 21 ({label}): {ev[0].elapsed_time(ev[1]) / 10:.3f} ms per call", flush=True)
