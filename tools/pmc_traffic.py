#!/usr/bin/env python3
"""HBM bytes per launch of the two kernels of a bench.py step, from the FETCH_SIZE / WRITE_SIZE
passes summarised by tools/pmc_summary.py.  Usage: pmc_traffic.py summary.json bench_line.json > pmc_traffic.json
FETCH_SIZE is doubled: gfx950 tallies 128-byte read requests at 64 bytes; WRITE_SIZE as reported (KiB)."""
import json
import sys

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
json.dump({"surfaces_per_launch": line["roofline"]["surfaces_per_launch"], **out,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 "
                  "--warmup 1 (tools/profile_round.sh); KiB per dispatch averaged over dispatches; FETCH_SIZE "
                  "doubled (gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE as reported",
           "raw_kib": raw}, sys.stdout, indent=1)
