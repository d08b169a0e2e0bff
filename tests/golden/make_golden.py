#!/usr/bin/env python3
"""Regenerate the committed fixtures under tests/golden/ (run in the build container only).

Reads the reference's DATA files with loaders that execute nothing from them
(``torch.load(weights_only=True)``, CSV text, and ``static_pickle`` for the two ``.pickle``
files) and writes small ``.npz`` fixtures:

  calibration_pin.npz   500 calibrated parameter sets + the QuantLib-evaluated objective values the
                        reference stored for them (``fit_errors``) + the market IV surfaces they
                        were fitted to.  This is the pin for the pricing restatement.
  decoder.npz           decoder weights / BatchNorm statistics, param_mean/std, conditioning
                        normalisation statistics and the 500 conditioning rows.
  golden_decode.npz     seeded (z, c) -> decoded parameters, produced by torch.nn layers loaded
                        with the reference checkpoint's own state-dict entries.
  golden_iv.npz         price ratios -> implied vols produced by SciPy's brentq / norm.cdf
                        with the reference's objective (variants A and B).
  golden_flags.npz      IV surfaces -> arbitrage-check results produced by the reference's own
                        check_static_arbitrage / check_calendar_arbitrage / check_all_arbitrage
                        (function bodies extracted from the reference file with ``ast`` and
                        executed here; only their outputs are stored).
  golden_prices.npz     frozen outputs of the C oracle for 64 parameter sets (regression anchor).

``generated_samples.pt`` and ``all_regimes_heston_params.pt`` are refused by the weights-only
loader and are therefore not used.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import static_pickle  # noqa: E402
import oracle  # noqa: E402

T_GRID = np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
LOGM_GRID = np.linspace(-0.2, 0.2, 21)


def calibration_pin():
    cal = static_pickle.load(f"{REF}/calibration_single_heston/NIFTY_heston_single_params.pickle")
    sur = static_pickle.load(f"{REF}/nifty_filtered_surfaces.pickle")
    dates = list(cal["dates"])
    params = np.array([[p["kappa"], p["theta"], p["sigma_v"], p["rho"], p["v0"]]
                       for p in cal["params"]], dtype=np.float64)
    fit = np.array(cal["fit_errors"], dtype=np.float64)
    feller = np.array(cal["feller_satisfied"], dtype=np.bool_)
    iv = np.stack([np.asarray(sur["surfaces"][d]["IV_surface"], np.float64) for d in dates])
    spots = np.array([float(sur["spot_prices"][d]) for d in dates])
    tg = np.asarray(sur["canonical_grid"]["T_grid"], np.float64)
    lg = np.asarray(sur["canonical_grid"]["logm_grid"], np.float64)
    failed = np.array([d in set(cal["failed_dates"]) for d in dates], dtype=np.bool_)
    cfg = cal["calibration_settings"]
    np.savez_compressed(
        f"{HERE}/calibration_pin.npz", dates=np.array(dates), params=params, fit_errors=fit,
        feller_satisfied=feller, iv_surfaces=iv, spot_prices=spots, T_grid=tg, logm_grid=lg,
        failed=failed, r=cfg["calibration"]["r"], q=cfg["calibration"]["q"],
        drift_weight=cfg["loss_function"]["drift_penalty_weight"],
        wasserstein_weight=cfg["loss_function"]["wasserstein_weight"],
        density_grid_points=cfg["wasserstein"]["density_grid_points"],
        max_fit_error=cfg["calibration"]["max_fit_error"])
    print("calibration_pin:", params.shape, iv.shape, "failed", failed.sum())
    return params


def decoder_fixture():
    ck = torch.load(f"{REF}/llm_options_assistant/best_model_2025/cvae_model.pt",
                    map_location="cpu", weights_only=True)
    st = torch.load(f"{REF}/condtional_vae/conditioning_normalization_stats.pt",
                    map_location="cpu", weights_only=True)
    sd = ck["model_state_dict"]
    out = {k: v.numpy() for k, v in sd.items() if k.startswith("decoder.")}
    out["param_mean"] = ck["param_mean"].numpy()
    out["param_std"] = ck["param_std"].numpy()
    out["raw_mean"] = st["raw_mean"].numpy()
    out["raw_std"] = st["raw_std"].numpy()
    # conditioning rows (already z-scored), CSV text
    import csv
    with open(f"{REF}/conditioning_variables_top8.csv") as fh:
        rows = list(csv.reader(fh))
    header = rows[0]
    dates = [r[0] for r in rows[1:]]
    vals = np.array([[float(x) for x in r[1:]] for r in rows[1:]], dtype=np.float64)
    out["cond_dates"] = np.array(dates)
    out["cond_names"] = np.array(header[1:])
    out["cond_rows"] = vals
    np.savez_compressed(f"{HERE}/decoder.npz", **out)
    print("decoder:", len(out), "arrays; cond rows", vals.shape)
    return ck, out


def torch_decoder(sd):
    """The decoder stack of cvae_model.py:84-105 as plain torch.nn layers, eval mode."""
    import torch.nn as nn
    seq = nn.Sequential(
        nn.Linear(12, 32), nn.BatchNorm1d(32), nn.ReLU(), nn.Dropout(0.15),
        nn.Linear(32, 64), nn.BatchNorm1d(64), nn.ReLU(), nn.Dropout(0.15),
        nn.Linear(64, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Dropout(0.15),
        nn.Linear(128, 5))
    seq.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")})
    return seq.eval()


def golden_decode(ck, dec):
    seq = torch_decoder(ck["model_state_dict"])
    g = torch.Generator().manual_seed(20251110)
    n = 256
    z = torch.randn(n, 4, generator=g)
    rows = torch.tensor(dec["cond_rows"], dtype=torch.float32)
    idx = torch.randint(0, rows.shape[0], (n,), generator=g)
    c = rows[idx]
    c[:100] = rows[-1]          # the config-1 style batch: one conditioning row repeated
    with torch.no_grad():
        norm = seq(torch.cat([z, c], dim=1))
        t = norm * ck["param_std"] + ck["param_mean"]
        p = t.clone()
        for i in (0, 1, 2, 4):
            p[:, i] = torch.exp(t[:, i])
        p[:, 3] = torch.tanh(t[:, 3])
    np.savez_compressed(f"{HERE}/golden_decode.npz", z=z.numpy(), c=c.numpy(),
                        normalized=norm.numpy(), params=p.numpy())
    print("golden_decode:", p.shape, "ranges", p.min(0).values.numpy(), p.max(0).values.numpy())
    return p.numpy()


def golden_prices_iv(cal_params, dec_params):
    params = np.concatenate([cal_params[::13][:40], dec_params[:24].astype(np.float64)])
    spot, r, q = 21000.0, 0.067, 0.0
    ratio = oracle.price_ratio_grid(params, LOGM_GRID, T_GRID, r, q)
    np.savez_compressed(f"{HERE}/golden_prices.npz", params=params, logm=LOGM_GRID, tau=T_GRID,
                        r=r, q=q, ratio=ratio)
    iv_a = np.empty_like(ratio)
    iv_b = np.empty_like(ratio)
    for s in range(ratio.shape[0]):
        for t, tau in enumerate(T_GRID):
            for k, lm in enumerate(LOGM_GRID):
                iv_a[s, t, k] = oracle.scipy_implied_vol_a(ratio[s, t, k], lm, tau, spot, r, q)
                iv_b[s, t, k] = oracle.scipy_implied_vol_b(ratio[s, t, k], lm, tau, spot, r, q)
    np.savez_compressed(f"{HERE}/golden_iv.npz", ratio=ratio, logm=LOGM_GRID, tau=T_GRID,
                        spot=spot, r=r, q=q, iv_a=iv_a, iv_b=iv_b)
    print("golden_iv: NaN A", np.isnan(iv_a).sum(), "NaN B", np.isnan(iv_b).sum(), "of", iv_a.size)
    return ratio, iv_a, iv_b


def reference_check_functions():
    """Extract the three check_* FunctionDefs from the reference file and exec only those."""
    from scipy.stats import norm
    src = open(f"{REF}/condtional_vae/generate_regime_surfaces.py").read()
    tree = ast.parse(src)
    want = {"check_static_arbitrage", "check_calendar_arbitrage", "check_all_arbitrage"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(body) == 3
    ns = {"np": np, "norm": norm}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<reference check functions>", "exec"), ns)
    return ns


def golden_flags(iv_b, cal_iv):
    ns = reference_check_functions()
    spot, r, q = 21000.0, 0.067, 0.0
    rng = np.random.default_rng(7)
    surfaces = []
    # (1) IV surfaces the variant-B inversion produced (some contain NaN -> None)
    surfaces += [iv_b[i] for i in range(iv_b.shape[0])]
    # (2) market surfaces from the calibration data set
    surfaces += [cal_iv[i] for i in range(0, 500, 25)]
    # (3) flat and perturbed surfaces that trip the checks
    surfaces.append(np.full((8, 21), 0.2))
    for _ in range(20):
        base = cal_iv[rng.integers(0, 500)].copy()
        base *= 1.0 + 0.15 * rng.standard_normal(base.shape)
        surfaces.append(np.abs(base) + 1e-3)
    s = np.stack(surfaces)
    out = np.full((s.shape[0], 4), -1, dtype=np.int32)
    for i, iv in enumerate(s):
        res = ns["check_all_arbitrage"](iv, T_GRID, LOGM_GRID, spot, r, q)
        if res is None:
            out[i] = (-1, -1, -1, 0)
        else:
            out[i] = (res["static_violations"], res["calendar_violations"],
                      res["total_violations"], int(res["is_arbitrage_free"]))
    np.savez_compressed(f"{HERE}/golden_flags.npz", iv=s, logm=LOGM_GRID, tau=T_GRID, spot=spot,
                        r=r, q=q, result=out)
    print("golden_flags:", s.shape, "invalid", (out[:, 0] < 0).sum(),
          "arb-free", (out[:, 3] == 1).sum(), "max static", out[:, 0].max(), "max cal", out[:, 1].max())


if __name__ == "__main__":
    oracle.build(force=True)
    cal_params = calibration_pin()
    ck, dec = decoder_fixture()
    dec_params = golden_decode(ck, dec)
    ratio, iv_a, iv_b = golden_prices_iv(cal_params, dec_params)
    cal_iv = np.load(f"{HERE}/calibration_pin.npz")["iv_surfaces"]
    golden_flags(iv_b, cal_iv)
