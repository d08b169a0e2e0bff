"""Variant A (closed-form guess + Householder steps) against AFIV_VARIANT_A_BRENT (brentq's path) on
the benchmark lattice: NaN masks, vol differences by conditioning, flag words, kernel times."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import afiv_bootstrap
from test_gpu_parity import sample_params, vega
afiv = afiv_bootstrap.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
logm = np.linspace(-0.2, 0.2, 50); tau = np.round(np.linspace(30, 730, 20)) / 365.0
SPOT, R, Q = 21000.0, 0.067, 0.0
g = afiv.Grid(logm, tau, R, Q)
p = torch.as_tensor(sample_params(n, 3)).cuda()
ratio = g.price(p)
out = {}
for v in ("A", "A_BRENT", "B"):
    iv, fl = g.implied_vol_checked(ratio, SPOT, v)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        iv, fl = g.implied_vol_checked(ratio, SPOT, v)
    torch.cuda.synchronize()
    print(v, "ms per call", (time.perf_counter() - t0) / 3 * 1e3, flush=True)
    out[v] = (iv, fl)
iva, fla = out["A"]; ivb, flb = out["A_BRENT"]
print("NaN masks equal:", bool(torch.equal(torch.isnan(iva), torch.isnan(ivb))), "nan frac", float(torch.isnan(iva).float().mean()))
m = min(n, 4096)
a = iva[:m].cpu().numpy(); b = ivb[:m].cpu().numpy()
vg = vega(b, logm, tau, SPOT, R, Q)
both = ~np.isnan(a) & ~np.isnan(b)
d = np.abs(a - b)
for lo, hi in ((1e-3, np.inf), (1e-5, 1e-3), (1e-8, 1e-5), (0, 1e-8)):
    s = both & (vg >= lo) & (vg < hi)
    if s.any():
        print("vega in [%g,%g): n=%d max|d|=%.3g p99=%.3g" % (lo, hi, s.sum(), d[s].max(), np.percentile(d[s], 99)))
print("flags equal on", float((fla == flb).all(dim=1).float().mean()), "of surfaces; valid frac", float(((fla[:, 3] & 4) != 0).float().mean()))
diff = (fla != flb).any(dim=1).nonzero().flatten()[:5]
for i in diff.tolist():
    print(i, fla[i].tolist(), flb[i].tolist())
