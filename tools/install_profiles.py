#!/usr/bin/env python3
"""Copies the summaries tools/profile_round.sh and tools/profile_configs.sh left under gpurun_out/
into profiles/<round>/ (and profiles/pmc_traffic.json, which bench.py reads).
usage: install_profiles.py <round dir name, e.g. r02> <prof_round tag> [<prof_configs tag> [<prof_configs_pmc tag>]]"""
import json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
cfg_tag = sys.argv[3] if len(sys.argv) > 3 else None
pmc_tag = sys.argv[4] if len(sys.argv) > 4 else None
O = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
P = os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)
s = json.load(open(os.path.join(O, "summary.json")))
out = {"command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-general-lattice (1 048 576 surfaces per launch), one "
                  "rocprofv3 --pmc pass per counter group (tools/profile_round.sh)", "mean_per_dispatch": {}}
for grp in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for name, vals in s[grp]["counters_mean_per_dispatch"].items():
        key = "price_kernel" if "price_kernel" in name else "iv_fast_kernel" if "iv_fast_kernel" in name else None
        if key is None:
            continue
        d = out["mean_per_dispatch"].setdefault(key, {})
        for k, v in vals.items():
            if k == "dispatches" and k in d:
                continue
            d[k] = v
json.dump(out, open(os.path.join(P, "pmc_counters_bench.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "stats", f"{tag}_kernel_stats.csv"), os.path.join(P, "kernel_stats_bench.csv"))
shutil.copy(os.path.join(O, "bench_stats.json"), os.path.join(P, "bench_under_rocprof.json"))
shutil.copy(os.path.join(O, "pmc_traffic.json"), os.path.join(ROOT, "profiles", "pmc_traffic.json"))
if cfg_tag:
    C = os.path.join(ROOT, "gpurun_out", f"prof_{cfg_tag}")
    with open(os.path.join(P, "bench_configs_under_rocprof.jsonl"), "w") as fh:
        for c in (1, 2, 4):
            shutil.copy(os.path.join(C, f"kernel_stats_config{c}.csv"), os.path.join(P, f"kernel_stats_config{c}.csv"))
            fh.write(open(os.path.join(C, f"bench_config{c}.json")).read().strip().splitlines()[-1] + "\n")
OURS = ("price_kernel", "iv_fast_kernel", "surface_kernel", "decode_kernel", "stage1_kernel", "arb_kernel",
        "ensemble_", "penalty_kernel", "density_cdf_kernel", "wasserstein_kernel")
if pmc_tag:
    # SQ counters of bench.py --config 2 / 4 (tools/profile_configs_pmc.sh): the library's own kernels only
    C = os.path.join(ROOT, "gpurun_out", f"prof_{pmc_tag}")
    for c in (2, 4):
        s = json.load(open(os.path.join(C, f"summary_config{c}.json")))
        res = {"command": f"python3 bench.py --config {c} --steps 3 --warmup 1, one rocprofv3 --pmc pass per counter "
                          "group (tools/profile_configs_pmc.sh); durations from the --kernel-trace --stats pass "
                          f"of --config {c} --steps 20 --warmup 3", "mean_per_dispatch": {}, "kernel_stats": []}
        for grp, entry in s.items():
            for name, vals in entry.get("counters_mean_per_dispatch", {}).items():
                if not any(k in name for k in OURS):
                    continue
                short = name.split("::")[-1].split("(")[0]
                res["mean_per_dispatch"].setdefault(short, {}).update(vals)
            for row in entry.get("kernel_stats", []):
                if any(k in row["Name"] for k in OURS):
                    res["kernel_stats"].append({"kernel": row["Name"].split("::")[-1].split("(")[0], "calls": int(row["Calls"]),
                                                "average_ns": float(row["AverageNs"]), "min_ns": int(row["MinNs"]),
                                                "max_ns": int(row["MaxNs"])})
        json.dump(res, open(os.path.join(P, f"pmc_counters_config{c}.json"), "w"), indent=1)
d = json.loads(open(os.path.join(P, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"],
      d["roofline_other_kernel"]["kernel_ms_per_step"])
print(open(os.path.join(P, "kernel_stats_bench.csv")).read()[:700])
