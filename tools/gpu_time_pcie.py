#!/usr/bin/env python3
"""What a host-resident consumer sees (diagnostic): the benchmark step with its parameters coming
from pinned host memory and its implied vols and flag words going back to pinned host memory.

The C ABI takes device pointers, so these copies are the caller's; this script times them the way
a caller would issue them -- the batch in chunks, each chunk's device-to-host copy on a second
stream behind that chunk's kernels, overlapping the next chunk's kernels -- next to the same step
with everything resident.

usage: gpu_time_pcie.py [surfaces] [chunks]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import afiv_bootstrap
afiv = afiv_bootstrap.load()
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
assert n % chunks == 0
rows = n // chunks
logm, tau = bench.grids()
nT, nK = len(tau), len(logm)
dev = torch.device("cuda", 0)
grid = afiv.Grid(logm, tau, bench.R, bench.Q, device=dev)
p_host = bench.synthetic_params(n, 3, "cpu").contiguous().pin_memory()
iv_host = torch.empty((n, nT, nK), dtype=torch.float64).pin_memory()
fl_host = torch.empty((n, 4), dtype=torch.int32).pin_memory()
p = p_host.to(dev)
ratio = torch.empty((rows, nT, nK), dtype=torch.float64, device=dev)
iv = torch.empty((2, rows, nT, nK), dtype=torch.float64, device=dev)       # double-buffered chunk
fl = torch.empty((2, rows, 4), dtype=torch.int32, device=dev)
copy = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
done = [torch.cuda.Event(), torch.cuda.Event()]


def step(transfers):
    if transfers:
        p.copy_(p_host, non_blocking=True)
    for c in range(chunks):
        b = c & 1
        sl = slice(c * rows, (c + 1) * rows)
        if transfers and c >= 2:
            main.wait_event(done[b])                 # the copy that last read this buffer
        grid.price(p[sl], out=ratio)
        grid.implied_vol_checked(ratio, bench.SPOT, "A", iv_out=iv[b], flags_out=fl[b])
        if transfers:
            copy.wait_stream(main)
            with torch.cuda.stream(copy):
                iv_host[sl].copy_(iv[b], non_blocking=True)
                fl_host[sl].copy_(fl[b], non_blocking=True)
                done[b].record(copy)
    if transfers:
        main.wait_stream(copy)


def timeit(transfers, seconds=2.0):
    step(transfers); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < seconds:
        step(transfers); torch.cuda.synchronize()
        reps += 1
    return (time.perf_counter() - t0) / reps


t_res = timeit(False)
t_pcie = timeit(True)
out_bytes = iv_host.numel() * 8 + fl_host.numel() * 4
print(f"{n} surfaces x {nT} x {nK}, {chunks} chunks: resident {t_res*1e3:.2f} ms = {n/t_res/1e6:.2f} M surfaces/s; "
      f"host in / host out {t_pcie*1e3:.2f} ms = {n/t_pcie/1e6:.2f} M surfaces/s "
      f"({out_bytes/1e9:.2f} GB back per step = {out_bytes/t_pcie/1e9:.1f} GB/s if the copies alone set the time)", flush=True)
# check against the resident path
ref_iv, ref_fl = grid.implied_vol_checked(grid.price(p[-rows:]), bench.SPOT, "A")
assert torch.equal(torch.nan_to_num(ref_iv.cpu()), torch.nan_to_num(iv_host[-rows:])) and torch.equal(ref_fl.cpu(), fl_host[-rows:])
print("host copies equal the resident results", flush=True)
# the copies alone
torch.cuda.synchronize()
t0 = time.perf_counter()
for c in range(chunks):
    iv_host[c * rows:(c + 1) * rows].copy_(iv[c & 1], non_blocking=True)
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f"device-to-host copy of the vols alone: {iv_host.numel()*8/t/1e9:.1f} GB/s", flush=True)
