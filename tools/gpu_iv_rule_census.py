#!/usr/bin/env python3
"""How many implied vols fall under each clause of the tests' implied-vol rule (GPU box).

The rule (tests/test_gpu_parity.py::assert_iv_close): where the Black-Scholes vega is at least 1e-3
the kernel's vol must be within 1e-6 of the oracle's brentq and the NaN masks must agree; below
that a vol may differ if it re-prices to the oracle vol's value within 64 eps F exp(-r tau), and a
small share of NaN-mask mismatches is accepted.  This tool counts, per workload and solver
variant: points, well- and ill-conditioned ones, NaN-mask mismatches of either kind,
ill-conditioned vols within 1e-6, those beyond it that re-price alike, and those that do neither.

usage: python tools/gpu_iv_rule_census.py > profiles/<round>/iv_rule_census.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
import oracle
from test_gpu_parity import LOGM_GRID, T_GRID, R, Q, SPOT, sample_params, vega, reprice_a, dev
from test_gpu_configs import STRESS_RHO, STRESS_SIGMA, STRESS_KAPPA, STRESS_LEVEL

afiv = afiv_bootstrap.load()


def census(got, ref, logm, tau, spot):
    v_ref, v_got = vega(ref, logm, tau, spot, R, Q), vega(got, logm, tau, spot, R, Q)
    v = np.where(np.isnan(v_ref), v_got, v_ref)
    well = v >= 1e-3
    both_nan = np.isnan(got) & np.isnan(ref)
    mism = np.isnan(got) != np.isnan(ref)
    live = ~both_nan & ~mism
    d = np.abs(got - ref)
    p_got, unit = reprice_a(got, logm, tau, spot, R, Q)
    p_ref, _ = reprice_a(ref, logm, tau, spot, R, Q)
    alike = np.abs(p_got - p_ref) <= unit
    ill = ~well & live
    with np.errstate(invalid="ignore"):
        out = {"points": int(got.size), "nan_on_both_sides": int(both_nan.sum()),
               "well_conditioned": int((well & live).sum()), "ill_conditioned": int(ill.sum()),
               "nan_mask_mismatch_well": int((mism & well).sum()), "nan_mask_mismatch_ill": int((mism & ~well).sum()),
               "well_beyond_1e-6": int((well & live & (d > 1e-6)).sum()),
               "well_max_diff": float(np.nanmax(np.where(well & live, d, 0.0))),
               "ill_within_1e-6": int((ill & (d <= 1e-6)).sum()),
               "ill_beyond_1e-6_repricing_alike": int((ill & (d > 1e-6) & alike).sum()),
               "ill_beyond_1e-6_not_alike": int((ill & (d > 1e-6) & ~alike).sum()),
               "ill_beyond_1e-4": int((ill & (d > 1e-4)).sum()),
               "ill_max_diff": float(np.nanmax(np.where(ill, d, 0.0)))}
    return out


def run(name, logm, tau, ratio, spot, variants, out):
    grid = afiv.Grid(logm, tau, R, Q)
    for variant in variants:
        iv = grid.implied_vol(dev(ratio), spot, variant).cpu().numpy()
        ref = oracle.implied_vol_grid(ratio, logm, tau, spot, R, Q, "A" if variant == "A_BRENT" else variant)
        if variant == "B":
            mism = np.isnan(iv) != np.isnan(ref)
            m = ~np.isnan(iv) & ~np.isnan(ref)
            d = np.abs(iv - ref)
            out[f"{name} / {variant}"] = {"points": int(iv.size), "nan_mask_mismatch": int(mism.sum()),
                                          "roots_compared": int(m.sum()), "beyond_1e-6": int((m & (d > 1e-6)).sum()),
                                          "beyond_1e-10": int((m & (d > 1e-10)).sum()),
                                          "max_diff": float(np.nanmax(np.where(m, d, 0.0)))}
        else:
            out[f"{name} / {variant}"] = census(iv, ref, logm, tau, spot)
        print(name, variant, json.dumps(out[f"{name} / {variant}"]), file=sys.stderr, flush=True)


out = {}
# (a) the benchmark lattice: 256 parameter sets x 20 x 50 (the set of
#     test_householder_solver_against_brentq_on_the_benchmark_lattice)
logm, tau = np.linspace(-0.2, 0.2, 50), np.round(np.linspace(30, 730, 20)) / 365.0
ratio = afiv.Grid(logm, tau, R, Q).price(dev(sample_params(256, 77))).cpu().numpy()
run("benchmark lattice 20x50, 256 sets", logm, tau, ratio, SPOT, ("A", "A_BRENT", "B"), out)
# (b) the reference's lattice: 1000 parameter sets x 8 x 21
ratio = afiv.Grid(LOGM_GRID, T_GRID, R, Q).price(dev(sample_params(1000, 78))).cpu().numpy()
run("reference lattice 8x21, 1000 sets", LOGM_GRID, T_GRID, ratio, SPOT, ("A", "A_BRENT", "B"), out)
# (c) the stress grid (270 sets, 7..60 days, 21 strikes)
sets = np.array([[k, lv, s, rho, lv] for rho in STRESS_RHO for s in STRESS_SIGMA for k in STRESS_KAPPA
                 for lv in STRESS_LEVEL])
tau5 = np.array([7, 14, 30, 60]) / 365.0
ratio = afiv.Grid(LOGM_GRID, tau5, R, Q).price(dev(sets)).cpu().numpy()
run("stress grid 4x21, 270 sets", LOGM_GRID, tau5, ratio, SPOT, ("A", "A_BRENT", "B"), out)
json.dump(out, sys.stdout, indent=1)
