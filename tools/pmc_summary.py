#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output: per-kernel mean counter values per dispatch (counter_collection)
and per-kernel duration statistics (kernel_stats).  Usage: pmc_summary.py <dir> [<dir> ...] > out.json"""
import csv, glob, json, os, sys
from collections import defaultdict

out = {}
for d in sys.argv[1:]:
    entry = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        entry["counters_mean_per_dispatch"] = {
            k: {c: sum(v) / len(v) for c, v in cs.items()} | {"dispatches": len(next(iter(cs.values())))}
            for k, cs in acc.items()}
    for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        with open(path) as fh:
            entry["kernel_stats"] = [row for row in csv.DictReader(fh)]
    out[os.path.basename(os.path.normpath(d))] = entry
json.dump(out, sys.stdout, indent=1)
