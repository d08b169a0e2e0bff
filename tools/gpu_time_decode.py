#!/usr/bin/env python3
"""Time of the CVAE decode kernel for large batches (diagnostic)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import afiv_bootstrap
afiv = afiv_bootstrap.load()
d = np.load(os.path.join(ROOT, "tests", "golden", "decoder.npz"))
dec = afiv.Decoder({k: d[k] for k in d.files if k.startswith("decoder.")}, d["param_mean"], d["param_std"])
for n in (100, 131072, 1048576):
    z = torch.randn(n, 4, device="cuda"); c = torch.randn(1, 8, device="cuda")
    dec.decode(z, c); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dec.decode(z, c)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"decode N={n}: {t*1e3:.0f} us ({n*22528/t/1e9:.2f} TFLOP/s fp32, {n/t/1e3:.1f} M samples/s)", flush=True)
