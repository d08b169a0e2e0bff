#!/usr/bin/env python3
"""HBM bytes per launch of the two kernels of a bench.py step, from the FETCH_SIZE / WRITE_SIZE
passes summarised by tools/pmc_summary.py.  Usage: pmc_traffic.py summary.json bench_line.json > pmc_traffic.json
FETCH_SIZE is doubled: gfx950 tallies 128-byte read requests at 64 bytes; WRITE_SIZE as reported (KiB)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

summary = json.load(open(sys.argv[1]))
line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
raw, out = {}, {}
for kernel in ("price_kernel", "iv_fast_kernel", "surface_kernel"):
    vals = {}
    for pas, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for name, cs in summary.get(pas, {}).get("counters_mean_per_dispatch", {}).items():
            if kernel in name and counter in cs:
                vals[counter] = cs[counter]
    if len(vals) == 2:
        raw[kernel] = vals
        out[kernel + "_hbm_bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    # share of the SIMDs' issue slots in use: vector instructions (SQ_ACTIVE_INST_VALU counts in units of
    # four cycles) plus the cycles the matrix pipe is busy, over 1024 SIMDs x the kernel's cycles
    # (SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles)
    sq = {}
    for pas in ("pmc_sq", "pmc_sq2"):
        for name, cs in summary.get(pas, {}).get("counters_mean_per_dispatch", {}).items():
            if kernel in name:
                sq.update(cs)
    if "SQ_ACTIVE_INST_VALU" in sq and "SQ_BUSY_CYCLES" in sq:
        out[kernel + "_issue_slots_busy"] = (4.0 * sq["SQ_ACTIVE_INST_VALU"] + sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)) / \
            (32.0 * sq["SQ_BUSY_CYCLES"])
json.dump({"kernel_sources_sha16": bench.kernel_sources_sha(),
           "surfaces_per_launch": line["roofline"]["surfaces_per_launch"], **out,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 "
                  "--warmup 1 (tools/profile_round.sh); KiB per dispatch averaged over dispatches; FETCH_SIZE "
                  "doubled (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE as reported",
           "raw_kib": raw}, sys.stdout, indent=1)
