"""Drop-in for the reference's ``generate_iv_surface_by_date`` module.

Keeps ``T_GRID``, ``LOGM_GRID``, ``DEVICE``, ``generate_iv_surface(params, T_grid, logm_grid,
spot, r, q) -> (iv, success)`` and the ``main()`` command line (generate_iv_surface_by_date.py:
50-59, :267-322, :329-337), and adds ``generate_iv_surfaces`` -- the same computation for a whole
batch of parameter rows in one pass over the GPU (decode -> Heston pricing -> implied vol ->
arbitrage checks), which is what ``main()`` uses instead of the per-sample Python loop of
:464-468.

Market-data download (yfinance) is outside this path.  ``main()`` takes the conditioning row
either from ``--conditioning_csv`` (rows already z-scored, as conditioning_variables_top8.csv)
or, when yfinance is importable, through ``fetch_conditioning``; otherwise it stops with the
reference's own message.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from datetime import datetime

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
for _p in (_ROOT, os.path.dirname(_HERE), _HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import afiv_bootstrap  # noqa: E402
from heston_model_ql import cached_grid  # noqa: E402
from cvae_model import ConditionalVAE_SingleHeston  # noqa: E402

_afiv = afiv_bootstrap.load()

# canonical lattice of the training data: 1M..24M x 21 strikes (:50-51)
T_GRID = np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
LOGM_GRID = np.linspace(-0.2, 0.2, 21)
DEFAULT_UNREST_INDEX = 100.457836
DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def generate_iv_surfaces(params, T_grid, logm_grid, spot, r, q, variant="A", return_flags=False,
                         on_device=False):
    """Batched generate_iv_surface: ``params`` is [N,5] (NumPy or torch, any float dtype; values
    are widened to float64 exactly as ``float(x)`` does at :286).

    Returns ``(iv [N,nT,nK] float64 ndarray with NaN, success [N] bool ndarray)`` and, with
    ``return_flags``, the [N,4] int32 flag words (static, calendar, butterfly, status).  With
    ``on_device`` the three are CUDA tensors and nothing crosses to the host: ``success`` then goes
    straight into ``ensemble_statistics`` as its selection mask."""
    if isinstance(params, torch.Tensor):
        p = params.detach().to("cuda", torch.float64)
    else:
        p = torch.as_tensor(np.asarray(params), device="cuda").to(torch.float64)
    p = p.reshape(-1, 5).contiguous()
    grid = cached_grid(logm_grid, T_grid, r, q)
    iv, flags, _ratio = grid.surface(p, float(spot), variant)
    bit = _afiv.STATUS_SUCCESS_A if variant in ("A", 0, "A_BRENT", 2) else _afiv.STATUS_SUCCESS_B
    success = (flags[:, 3] & bit) != 0
    if not on_device:
        iv, success, flags = iv.cpu().numpy(), success.cpu().numpy(), flags.cpu().numpy()
    if return_flags:
        return iv, success, flags
    return iv, success


def generate_iv_surface(params, T_grid, logm_grid, spot, r, q):
    """One parameter set -> (iv_surface [nT,nK], success).  NaN marks points the reference's
    guard or root finder rejects (:302-306, :316-318); ``success`` is nan_count/points < 0.1
    (:321).  A failure to set the model up returns (None, False) as :285-288."""
    try:
        row = np.array([float(x) for x in params], dtype=np.float64).reshape(1, 5)
    except Exception:
        return None, False
    iv, success = generate_iv_surfaces(row, T_grid, logm_grid, spot, r, q)
    return iv[0], bool(success[0])


def decode_parameters(model, z, conditioning_batch, param_mean, param_std):
    """decode -> de-normalise -> exp / tanh (:438-448) in one kernel on the GPU.  Returns the
    float32 parameters [N,5] the reference stores and the same values widened to float64, both on
    the device: the second goes straight into the pricing kernel."""
    return model.decode_parameters(z, conditioning_batch, param_mean, param_std)


def ensemble_statistics(surfaces, success=None):
    """nan-aware mean / median / 5th / 95th percentile over valid surfaces (:474-478), computed on
    the GPU with NumPy's arithmetic (np.nanmean, np.nanmedian, np.nanpercentile "linear"), for any
    number of surfaces.  ``surfaces`` is the stack of valid surfaces (NumPy, as the reference
    builds it at :470) or a CUDA tensor of ALL surfaces together with their ``success`` mask
    (``generate_iv_surfaces(..., on_device=True)``), in which case the vols never leave the device:
    only the four (nT, nK) result surfaces come back."""
    if isinstance(surfaces, torch.Tensor) and surfaces.is_cuda:
        s = surfaces.to(torch.float64)
    else:
        s = torch.as_tensor(np.ascontiguousarray(np.asarray(surfaces, dtype=np.float64)), device="cuda")
    out = _afiv.ensemble_stats(s, success).cpu().numpy()
    return out[0], out[1], out[2], out[3]


def generate_for_dates(model, conditioning_rows, n_samples, param_mean, param_std, spot, r, q, seeds=None,
                       T_grid=None, logm_grid=None):
    """The script's body for MANY dates in one pass (the notebook runs the script once per date):
    ``conditioning_rows`` is [D, 8] (normalised), every date gets ``n_samples`` latent draws
    (``seeds[d]`` seeds date d's draw, :439), and decode -> pricing -> implied vol -> checks ->
    per-date statistics (:438-478) run as four launches over the D * n_samples surfaces.

    Returns a dict of CUDA tensors: ``params`` [D, n, 5] float32, ``iv`` [D, n, nT, nK], ``success``
    [D, n] bool, ``flags`` [D, n, 4] int32 and ``stats`` [D, 4, nT, nK] = mean, median, 5th and 95th
    percentile over each date's successful surfaces (NaN where a date has none at a grid point)."""
    T_grid = T_GRID if T_grid is None else T_grid
    logm_grid = LOGM_GRID if logm_grid is None else logm_grid
    rows = torch.as_tensor(np.asarray(conditioning_rows), dtype=torch.float32)
    D = rows.shape[0]
    if seeds is None:
        z = torch.randn(D * n_samples, 4)
    else:
        z = torch.cat([torch.randn(n_samples, 4, generator=torch.Generator().manual_seed(int(sd))) for sd in seeds])
    c = rows.repeat_interleave(n_samples, 0)
    params32, params64 = decode_parameters(model, z.to(DEVICE), c.to(DEVICE), param_mean, param_std)
    iv, success, flags = generate_iv_surfaces(params64, T_grid, logm_grid, spot, r, q, return_flags=True,
                                              on_device=True)
    stats = _afiv.ensemble_stats(iv, success, groups=D)
    nT, nK = len(T_grid), len(logm_grid)
    return {"params": params32.view(D, n_samples, 5), "iv": iv.view(D, n_samples, nT, nK),
            "success": success.view(D, n_samples), "flags": flags.view(D, n_samples, 4), "stats": stats}


def conditioning_from_csv(path, date_str):
    """Already-normalised conditioning row for ``date_str`` (nearest earlier row if absent)."""
    import csv
    with open(path) as fh:
        rows = list(csv.reader(fh))
    dates = [r[0] for r in rows[1:]]
    vals = np.array([[float(x) for x in r[1:]] for r in rows[1:]], dtype=np.float32)
    idx = max((i for i, d in enumerate(dates) if d <= date_str), default=len(dates) - 1)
    return vals[idx], dates[idx]


def save_outputs(output_dir, date_str, params_np, surfaces_array, stats, spot, r, q):
    """iv_surfaces.pt + the two CSV matrices with the reference's keys and labels (:495-520)."""
    import pandas as pd
    mean_s, median_s, p5_s, p95_s = stats
    os.makedirs(output_dir, exist_ok=True)
    torch.save({"date": date_str, "params": params_np, "surfaces": surfaces_array,
                "mean_surface": mean_s, "median_surface": median_s, "p5_surface": p5_s,
                "p95_surface": p95_s, "T_grid": T_GRID, "logm_grid": LOGM_GRID, "spot": spot,
                "r": r, "q": q}, os.path.join(output_dir, "iv_surfaces.pt"))
    rows = [f"{int(round(T * 12))}M" for T in T_GRID]
    cols = [f"{lm * 100:.1f}%" for lm in LOGM_GRID]
    pd.DataFrame(mean_s * 100, index=rows, columns=cols).to_csv(os.path.join(output_dir, "mean_iv_surface.csv"))
    pd.DataFrame(median_s * 100, index=rows, columns=cols).to_csv(os.path.join(output_dir, "median_iv_surface.csv"))


def main(argv=None):
    ap = argparse.ArgumentParser(description="Generate IV surface for a specific date using Conditional VAE")
    ap.add_argument("--date", type=str, required=True, help="Target date (YYYY-MM-DD)")
    ap.add_argument("--n_samples", type=int, default=500)
    ap.add_argument("--output_dir", type=str, default="results_date")
    ap.add_argument("--spot", type=float, default=None)
    ap.add_argument("--r", type=float, default=0.067)
    ap.add_argument("--q", type=float, default=0.0)
    ap.add_argument("--checkpoint", type=str, default=None, help="cvae_model.pt (reference layout)")
    ap.add_argument("--conditioning_csv", type=str, default=None,
                    help="z-scored conditioning rows (offline replacement for the market download)")
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)

    target_date = datetime.strptime(args.date, "%Y-%m-%d").date()
    date_str = target_date.strftime("%Y-%m-%d")
    if not torch.cuda.is_available():
        print("Error: this generator needs a GPU")
        return 1
    ckpt_path = args.checkpoint or os.path.join(_HERE, "..", "llm_options_assistant", "best_model_2025",
                                                "cvae_model.pt")
    checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    param_mean = checkpoint["param_mean"].to(DEVICE)
    param_std = checkpoint["param_std"].to(DEVICE)
    model = ConditionalVAE_SingleHeston(param_dim=5, conditioning_dim=8, latent_dim=4,
                                        hidden_dims=[128, 64, 32], encoder_activation="tanh",
                                        decoder_activation="relu", dropout=0.15).to(DEVICE)
    model.load_state_dict(checkpoint["model_state_dict"])
    model.eval()
    print(" Model loaded")

    if args.spot is None:
        args.spot = 21000.0          # the reference's own offline fallback (:406)
        print(f"  → Using default: {args.spot}")
    else:
        print(f"\nUsing provided spot price: {args.spot:.2f}")

    if args.conditioning_csv:
        cond, used = conditioning_from_csv(args.conditioning_csv, date_str)
        print(f" Conditioning row of {used} (already normalised)")
        conditioning = torch.tensor(cond, dtype=torch.float32, device=DEVICE)
    else:
        try:
            import yfinance  # noqa: F401
        except ImportError:
            print("Error: pip install yfinance nsepy")
            return 1
        print("Error: live market download is not part of this build; pass --conditioning_csv")
        return 1
    conditioning_batch = conditioning.unsqueeze(0).repeat(args.n_samples, 1)

    print(f"\nGenerating {args.n_samples} Heston parameter sets...")
    gen = None
    if args.seed is not None:
        gen = torch.Generator().manual_seed(args.seed)
    z = torch.randn(args.n_samples, 4, generator=gen).to(DEVICE)
    params32, params64 = decode_parameters(model, z, conditioning_batch, param_mean, param_std)
    params_np = params32.cpu().numpy()               # only for the saved file: the path stays on the device
    print(f" Generated {len(params_np)} parameter sets")

    print("\nGenerating IV surfaces...")
    iv, success = generate_iv_surfaces(params64, T_GRID, LOGM_GRID, args.spot, args.r, args.q, on_device=True)
    n_ok = int(success.sum())
    if n_ok == 0:
        print(" No valid surfaces generated!")
        return 1
    stats = ensemble_statistics(iv, success)          # on the device, the success bits as the selection mask
    surfaces_array = iv[success].cpu().numpy()        # only for the saved file (:495-508 stores them)
    print(f" Valid surfaces: {n_ok}/{args.n_samples} ({n_ok / args.n_samples * 100:.1f}%)")
    print(f"  Mean IV (overall):   {np.nanmean(stats[0]) * 100:.2f}%")

    output_dir = os.path.join(_HERE, args.output_dir, date_str) if not os.path.isabs(args.output_dir) \
        else os.path.join(args.output_dir, date_str)
    save_outputs(output_dir, date_str, params_np, surfaces_array, stats, args.spot, args.r, args.q)
    print(f"\n Saved to: {os.path.join(output_dir, 'iv_surfaces.pt')}")
    print(f"Spot price:              {args.spot:.2f}")
    print(f"Valid surfaces:          {n_ok}/{args.n_samples} ({n_ok / args.n_samples * 100:.1f}%)")
    print(json.dumps({"output_dir": output_dir}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
