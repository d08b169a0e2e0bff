#!/usr/bin/env python3
"""Prints the GPU kernel's and the oracle's distance to the 50-digit prices (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import afiv_bootstrap, oracle
afiv = afiv_bootstrap.load()
g = np.load(os.path.join(ROOT, "tests/golden/exact_prices.npz"))
grid = afiv.Grid(g["logm"], g["tau"], float(g["r"]), float(g["q"]))
got = grid.price(torch.as_tensor(g["params"], device="cuda")).cpu().numpy()
orc = oracle.price_ratio_grid(g["params"], g["logm"], g["tau"], float(g["r"]), float(g["q"]))
ref = g["ratio"]
plain = 1e-8 * np.maximum(np.abs(ref), 1e-4)
for s in np.unique(g["params"][:, 2]):
    m = g["params"][:, 2] == s
    print(f"sigma_v={s:g}: gpu max err {np.abs(got-ref)[m].max():.3e} (x plain tol {(np.abs(got-ref)/plain)[m].max():.3f})  "
          f"oracle max err {np.abs(orc-ref)[m].max():.3e} (x plain tol {(np.abs(orc-ref)/plain)[m].max():.3f})")
