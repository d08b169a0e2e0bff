#!/usr/bin/env python3
"""Where the GPU pipeline and the oracle disagree on a regime surface's validity (diagnostic):
for every point whose NaN status differs, the two price ratios, their difference in ulps, and how
far the price is from the guard and from the bracket ends of the variant-B objective."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
afiv = afiv_bootstrap.load()
import oracle
g = np.load(os.path.join(ROOT, "tests", "golden", "regime_fixture.npz"))
spot, r, q = float(g["spot"]), float(g["r"]), float(g["q"])
T, LM = g["T_grid"], g["logm_grid"]
grid = afiv.Grid(LM, T, r, q)
for name, p32 in zip(g["names"], g["params"]):
    p = p32.astype(np.float64)
    ratio_o, iv_o, _ = oracle.surface_batch(p, LM, T, spot, r, q, "B")
    pd = torch.as_tensor(p).cuda()
    ratio_g = grid.price(pd)
    iv_g = grid.implied_vol(ratio_g, spot, "B").cpu().numpy()
    iv_go = grid.implied_vol(torch.as_tensor(ratio_o).cuda(), spot, "B").cpu().numpy()    # GPU solver on oracle prices
    ratio_g = ratio_g.cpu().numpy()
    diff = np.isnan(iv_g) != np.isnan(iv_o)
    if not diff.any():
        continue
    print(f"{name}: {diff.sum()} points differ; valid surfaces gpu {np.sum(~np.isnan(iv_g).any(axis=(1,2)))} oracle {np.sum(~np.isnan(iv_o).any(axis=(1,2)))}"
          f"; GPU solver on the oracle's prices differs at {np.sum(np.isnan(iv_go) != np.isnan(iv_o))} points")
    for i, t, k in zip(*np.nonzero(diff)):
        ro, rg = ratio_o[i, t, k], ratio_g[i, t, k]
        hp = ro * spot * np.exp(-r * T[t])
        intrinsic = max(spot - spot * np.exp(LM[k]) * np.exp(-r * T[t]), 0.0)
        print(f"   surface {i} t={t} k={k} params {p[i].tolist()}\n      ratio oracle {ro!r} gpu {rg!r} diff {abs(ro-rg)/np.spacing(abs(ro)):.1f} ulp; "
              f"price {hp!r} 0.99*intrinsic {0.99*intrinsic!r} iv oracle {iv_o[i,t,k]!r} gpu {iv_g[i,t,k]!r}")
