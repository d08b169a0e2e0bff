import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import afiv_bootstrap, oracle
afiv = afiv_bootstrap.load()
from test_gpu_parity import sample_params, SPOT, R, Q, dev
logm = np.linspace(-0.2, 0.2, 50); tau = np.round(np.linspace(30, 730, 20)) / 365.0
g = afiv.Grid(logm, tau, R, Q)
p = sample_params(256, 77)
ratio = g.price(dev(p))
for v in ("A", "B"):
    iv = g.implied_vol(ratio, SPOT, v).cpu().numpy()
    ref = oracle.implied_vol_grid(ratio.cpu().numpy(), logm, tau, SPOT, R, Q, v)
    both = ~np.isnan(iv) & ~np.isnan(ref)
    d = np.abs(iv - ref)[both]
    print(v, "points", both.sum(), "nan mismatch", int((np.isnan(iv) != np.isnan(ref)).sum()),
          "frac <=1e-12:", np.mean(d <= 1e-12), "<=1e-9:", np.mean(d <= 1e-9), "<=1e-7:", np.mean(d <= 1e-7), "max", d.max())
