#!/usr/bin/env python3
"""Benchmark of the surface-generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (Heston pricing -> implied vol -> arbitrage checks) over one
batch of synthetic parameter sets already resident in HBM, on the workload the headline metric
is quoted on: 50 strikes x 20 maturities, FP64 (BASELINE.json configs[2]; the per-GPU batch is
1 048 576 / 8 surfaces so that eight GPUs process the configured 1 M surfaces per step).
Surfaces are independent, so ranks hold disjoint row blocks (weak scaling) and the data path
has no collective; at N > 1 the 16-byte flag words are all-gathered once per step.

Prints ONE JSON line (rank 0).  ``roofline`` describes the dominant kernel (pricing), timed
with events on the launch stream inside the timed region; ``cpu_baseline`` times the CPU oracle
(test infrastructure, used here only as the measured baseline) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import afiv_bootstrap  # noqa: E402

FP64_VECTOR_PEAK_TFLOPS = 78.6     # MI355X: half of the 157.3 TFLOP/s FP32 vector peak
N_STRIKES, N_MATURITIES = 50, 20
SPOT, R, Q = 21000.0, 0.067, 0.0
SURFACES_PER_GPU = 131072          # x 8 GPUs = 1 048 576 surfaces per step


def pricing_flops(n_t: int, n_k: int) -> float:
    """Algorithmic FLOPs of the pricing kernel per surface: 288 characteristic-function
    evaluations per maturity at 400 FLOPs each (complex sqrt/exp/log/div weighted add/mul 1,
    fma 2, div/sqrt 10, exp/log 30, sincos 40, atan2 50) plus 2 integrals x 144 nodes x 2 FMA
    per strike."""
    return n_t * 288 * 400.0 + n_t * n_k * 288 * 4.0


def iv_checks_flops(n_t: int, n_k: int) -> float:
    """Algorithmic FLOPs of the implied-vol + checks kernel per surface: 12 root-finder iterations
    x 120 per point, plus 125 per point for the three checks."""
    return n_t * n_k * (1440.0 + 125.0)


def surface_flops(n_t: int, n_k: int) -> float:
    return pricing_flops(n_t, n_k) + iv_checks_flops(n_t, n_k)


def synthetic_params(n: int, seed: int, device) -> torch.Tensor:
    """BASELINE config 3 sampler: exp/tanh(param_mean + param_std * eps), eps ~ N(0,1) float32,
    clamped to the calibration bounds (run_single_heston_calibration.py:237-243)."""
    mean = torch.tensor([-0.2858, -4.4085, -3.7188, 1.1875, -3.7454])
    std = torch.tensor([1.6307, 1.4661, 2.1172, 1.4222, 1.1040])
    g = torch.Generator().manual_seed(seed)
    t = mean + std * torch.randn(n, 5, generator=g)
    p = torch.exp(t)
    p[:, 3] = torch.tanh(t[:, 3])
    lo = torch.tensor([1e-3, 1e-3, 1e-3, -0.999, 1e-3])
    hi = torch.tensor([20.0, 2.0, 5.0, 0.999, 2.0])
    p = torch.minimum(torch.maximum(p.double(), lo.double()), hi.double())
    return p.to(device)


def grids():
    logm = np.linspace(-0.2, 0.2, N_STRIKES)
    tau = np.round(np.linspace(30, 730, N_MATURITIES)) / 365.0
    return logm, tau


def cpu_baseline(logm, tau, seconds_target: float = 20.0):
    """Times the CPU oracle (one option at a time: 2 x 144-node quadrature per price, Brent per
    implied vol, then the checks) on a bounded sample of the same workload."""
    import oracle
    threads = oracle.set_threads(min(16, os.cpu_count() or 1))
    p = synthetic_params(2048, 3, "cpu").numpy()
    t0 = time.perf_counter()
    oracle.surface_batch(p[:threads], logm, tau, SPOT, R, Q, "A")
    per_surface_cpu = (time.perf_counter() - t0)            # seconds of one core per surface
    n = int(max(threads, min(2048, seconds_target / max(per_surface_cpu, 1e-6))))
    n -= n % threads
    t0 = time.perf_counter()
    oracle.surface_batch(p[:n], logm, tau, SPOT, R, Q, "A")
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "surfaces/s", "cores": threads, "kind": "port",
            "sample": f"{n} surfaces of the same 20x50 workload, C oracle "
                      f"(per-option quadrature + Brent + checks), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--surfaces-per-gpu", type=int, default=SURFACES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo rehearses the multi-rank path on a box with fewer GPUs than ranks "
                         "(ranks share devices, the flag words are gathered through host memory)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with "
                         "python -m torch.distributed.run --nproc-per-node N)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    afiv = afiv_bootstrap.load()
    logm, tau = grids()
    grid = afiv.Grid(logm, tau, R, Q, device=device)
    n_local = args.surfaces_per_gpu
    params = synthetic_params(n_local, 3 + rank, device)
    ratio = torch.empty((n_local, N_MATURITIES, N_STRIKES), dtype=torch.float64, device=device)
    iv = torch.empty_like(ratio)
    flags_all = torch.empty((world * n_local, 4), dtype=torch.int32, device=device) if world > 1 else None

    flags_local = torch.empty((n_local, 4), dtype=torch.int32, device=device)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    def step(events=None):
        # the two launches afiv_surface makes, issued separately so that each can be timed
        if events:
            events[0].record()
        grid.price(params, out=ratio)
        if events:
            events[1].record()
        grid.implied_vol_checked(ratio, SPOT, "A", iv_out=iv, flags_out=flags_local)
        if events:
            events[2].record()
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(flags_all, flags_local)
            else:
                host = flags_local.cpu()
                gathered = torch.empty((world * n_local, 4), dtype=torch.int32)
                dist.all_gather_into_tensor(gathered, host)
                flags_all.copy_(gathered)
        return flags_local

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        flags = step(ev[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The surfaces themselves stay sharded while the job runs; if one consumer wants them all, the
    # vols are gathered ONCE at the end (not part of a step): time that exchange separately.
    final_gather = None
    if world > 1 and args.backend == "nccl":
        try:
            iv_all = torch.empty((world * n_local, N_MATURITIES, N_STRIKES), dtype=torch.float64, device=device)
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            dist.all_gather_into_tensor(iv_all, iv)
            torch.cuda.synchronize()
            dist.barrier()
            final_gather = {"what": "all-gather of the implied vols after the last step (outside the timed steps)",
                            "bytes_per_rank": iv.numel() * 8, "ms": (time.perf_counter() - tg) * 1e3}
            del iv_all
        except Exception as exc:   # the headline number must not depend on this optional exchange
            final_gather = {"error": repr(exc)}

    k_price = float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e-3
    k_iv = float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) * 1e-3

    if rank == 0:
        total_surfaces = world * n_local * args.steps
        value = total_surfaces / elapsed
        achieved = pricing_flops(N_MATURITIES, N_STRIKES) * n_local / k_price / 1e12
        fl = flags.cpu().numpy()
        traffic = traffic_iv = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as fh:
                pmc = json.load(fh)
            if pmc.get("surfaces_per_launch") == n_local:
                traffic = pmc.get("price_kernel_hbm_bytes_per_launch")
                traffic_iv = pmc.get("surface_kernel_hbm_bytes_per_launch")
        out = {
            "metric": "arb-checked IV surfaces/sec", "value": value, "unit": "surfaces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic Heston parameter sets x 50 strikes x 20 maturities, "
                                   "FP64 pricing + implied vol (variant A, Brent) + static/"
                                   "calendar/butterfly checks (BASELINE.json configs[2])",
                       "surfaces_per_gpu_per_step": n_local, "strikes": N_STRIKES,
                       "maturities": N_MATURITIES, "spot": SPOT, "r": R, "q": Q,
                       "parallelism": f"rows sharded over {world} rank(s), flags all-gather only"},
            "roofline": {"bound": "mfma", "pipe": "FP64 vector + v_mfma_f64 (same 78.6 TFLOP/s peak)",
                         "kernel": "price_kernel", "achieved": achieved,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                         "flops_per_surface": pricing_flops(N_MATURITIES, N_STRIKES),
                         "kernel_ms": {"price": k_price * 1e3, "implied_vol_and_checks": k_iv * 1e3},
                         "note": "the kernel the utilisation target names (and the larger share of the "
                                 "algorithmic FLOPs); by time the implied-vol kernel is the longer of the "
                                 "two, its roofline follows"},
            # the other kernel of a step, same conventions
            "roofline_iv_checks": {"bound": "mfma", "pipe": "FP64 vector (no matrix work in this kernel)",
                                   "kernel": "surface_kernel",
                                   "achieved": iv_checks_flops(N_MATURITIES, N_STRIKES) * n_local / k_iv / 1e12,
                                   "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": iv_checks_flops(N_MATURITIES, N_STRIKES) * n_local / k_iv / 1e12
                                   / FP64_VECTOR_PEAK_TFLOPS,
                                   "traffic": traffic_iv,
                                   "flops_per_surface": iv_checks_flops(N_MATURITIES, N_STRIKES)},
            "whole_path_tflops": surface_flops(N_MATURITIES, N_STRIKES) * value / 1e12,
            "check": {"valid_surfaces_frac": float(np.mean((fl[:, 3] & 4) != 0)),
                      "arbitrage_free_frac": float(np.mean((fl[:, 3] & 1) != 0)),
                      "success_frac": float(np.mean((fl[:, 3] & 8) != 0))},
        }
        if final_gather is not None:
            out["final_gather"] = final_gather
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(logm, tau)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
