#!/usr/bin/env python3
"""Re-runs trials of tools/gpu_fuzz.py by seed and lists the parameter rows whose prices leave the
tolerance, with the GPU and oracle values of the worst point (diagnostic).
usage: gpu_fuzz_diag.py seed [seed ...]; writes gpurun_out/fuzz_diag.npz"""
import os, sys, runpy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seeds = [int(s) for s in sys.argv[1:]]
sys.argv = [sys.argv[0], "0"]                      # zero budget: import the helpers only
ns = runpy.run_path(os.path.join(ROOT, "tools", "gpu_fuzz.py"))
afiv, oracle, dev, price_tol = ns["afiv"], ns["oracle"], ns["dev"], ns["price_tol"]
out = {}
for seed in seeds:
    c = ns["make_case"](seed)
    p = c["p"]
    g = afiv.Grid(c["logm"], c["tau"], c["r"], c["q"], trunc_days=c["trunc"])
    ref = oracle.price_ratio_grid(p, c["logm"], c["tau"], c["r"], c["q"], c["trunc"])
    got = g.price(dev(p)).cpu().numpy()
    bad = ns["price_bad"](got, ref, p, c["tau"], c["logm"], c["r"], c["q"], c["trunc"])
    ex = np.abs(got - ref) * bad
    print(f"seed {seed}: {c['desc']}")
    one = np.isnan(got) != np.isnan(ref)
    for i, t, k in list(zip(*np.nonzero(one)))[:6]:
        print(f"   one-sided NaN: row {i} params {p[i].tolist()} tau={c['tau'][t]!r} logm={c['logm'][k]!r} gpu {got[i, t, k]!r} oracle {ref[i, t, k]!r}")
    for i in np.argsort(-np.nanmax(ex.reshape(len(p), -1), axis=1))[:4]:
        t, k = np.unravel_index(np.nanargmax(ex[i]), ex[i].shape)
        print(f"   row {i} params {p[i].tolist()} |diff| outside the allowance {ex[i, t, k]:.3g} at tau={c['tau'][t]:.6g} logm={c['logm'][k]:.4g}"
              f" gpu {got[i, t, k]!r} oracle {ref[i, t, k]!r}")
    iv, fl = g.implied_vol_checked(dev(ref), c["spot"], c["variant"])
    iv, fl = iv.cpu().numpy(), fl.cpu().numpy()
    fl_o = oracle.arb_check(iv, ref, c["logm"], c["tau"], c["spot"], c["r"], c["q"], c["variant"])
    for i in np.nonzero(np.any(fl != fl_o, axis=1))[0][:4]:
        print(f"   flags differ: row {i} params {p[i].tolist()} gpu {fl[i].tolist()} oracle {fl_o[i].tolist()}"
              f" nan vols {int(np.isnan(iv[i]).sum())} non-finite ratios {int((~np.isfinite(ref[i])).sum())}")
    out[f"iv{seed}"] = iv
    out[f"p{seed}"] = p; out[f"got{seed}"] = got; out[f"ref{seed}"] = ref
    out[f"logm{seed}"] = c["logm"]; out[f"tau{seed}"] = c["tau"]
    out[f"rq{seed}"] = np.array([c["r"], c["q"], float(c["trunc"])])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "fuzz_diag.npz"), **out)
