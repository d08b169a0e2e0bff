#!/usr/bin/env python3
"""Builds tests/golden/regime_fixture.npz from the reference's own regime parameter sets
(llm_options_assistant/best_model_2025/regime_samples/all_regimes_heston_params.pt, 8 regimes x
1000 x 5 float32) and the per-regime statistics of generate_regime_surfaces.py:296-362 for them.

The file holds NumPy scalars next to its tensors, so torch.load(weights_only=True) refuses it; it
is read here WITHOUT unpickling: the zip member data.pkl is walked opcode by opcode
(pickletools.genops executes nothing) to find, for each name under 'regimes', the key of its
storage, its shape and its stride, and the float32 values are taken from the raw storage member.

Expected values: the variant-B surfaces come from the C oracle (restated price + restated brentq,
generate_regime_surfaces.py:106-182); the violation counts from the reference's OWN
check_all_arbitrage / check_static_arbitrage / check_calendar_arbitrage, whose function bodies are
extracted from the reference file with ``ast`` and executed here on those surfaces (only their
outputs are stored).  Build container only."""
import ast
import os
import pickletools
import sys
import zipfile

import numpy as np
from scipy.stats import norm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference"
PT = f"{REF}/llm_options_assistant/best_model_2025/regime_samples/all_regimes_heston_params.pt"
T_GRID = np.array([0.08333333, 0.16666667, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0])
LOGM_GRID = np.linspace(-0.2, 0.2, 21)
ROWS_PER_REGIME = 64
NOISE_LEVEL = 1e-12          # price ratio below which a point is the quadrature's rounding noise


def read_regime_tensors(path):
    """{regime name: float32 [rows, 5]} by a static walk of the pickle stream."""
    z = zipfile.ZipFile(path)
    pkl = [n for n in z.namelist() if n.endswith("/data.pkl")][0]
    base = pkl[: -len("data.pkl")]
    assert z.read(base + "byteorder").decode().strip() == "little"
    ops = [(o.name, a) for o, a, _ in pickletools.genops(z.read(pkl))]
    rebuild_memo = None
    out, i = {}, 0
    names = [a for n, a in ops if n == "BINUNICODE"]
    assert names[0] == "regimes"
    last_str = None
    while i < len(ops):
        name, arg = ops[i]
        if name == "BINUNICODE":
            if arg == "param_names":
                break                                   # end of the 'regimes' dictionary
            last_str = arg
        is_rebuild = (name == "GLOBAL" and arg == "torch._utils _rebuild_tensor_v2")
        if is_rebuild and rebuild_memo is None and ops[i + 1][0] == "BINPUT":
            rebuild_memo = ops[i + 1][1]
        if is_rebuild or (name == "BINGET" and arg == rebuild_memo and rebuild_memo is not None):
            regime = last_str
            # persistent id ('storage', FloatStorage, key, 'cpu', numel), then offset, shape, stride
            j = i + 1
            ints, key = [], None
            while ops[j][0] != "REDUCE" or len(ints) < 6:
                n2, a2 = ops[j]
                if n2 == "BINUNICODE" and a2.isdigit() and key is None:
                    key = a2
                if n2 in ("BININT", "BININT1", "BININT2"):
                    ints.append(a2)
                j += 1
                if ops[j][0] == "REDUCE" and len(ints) >= 6:
                    break
            numel, offset, rows, cols, s0, s1 = ints[:6]
            assert (offset, cols, s0, s1) == (0, 5, 5, 1) and numel == rows * cols, (regime, ints)
            raw = np.frombuffer(z.read(f"{base}data/{key}"), dtype="<f4")
            assert raw.size == numel
            out[regime] = raw.reshape(rows, cols).copy()
            i = j
        i += 1
    return out


def reference_check_functions():
    src = open(f"{REF}/condtional_vae/generate_regime_surfaces.py").read()
    want = {"check_static_arbitrage", "check_calendar_arbitrage", "check_all_arbitrage"}
    body = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(body) == 3
    ns = {"np": np, "norm": norm}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<reference check functions>", "exec"), ns)
    return ns


def main():
    regimes = read_regime_tensors(PT)
    assert len(regimes) == 8 and all(v.shape == (1000, 5) for v in regimes.values()), {k: v.shape for k, v in regimes.items()}
    ns = reference_check_functions()
    spot, r, q = 21000.0, 0.067, 0.0                      # the script's defaults (:40-44)
    names, params, stats = [], [], []
    per = {k: [] for k in ("valid", "static", "calendar", "free", "fragile")}
    for name, full in regimes.items():
        p32 = full[:ROWS_PER_REGIME]
        p = p32.astype(np.float64)                         # params[idx] elements widened by float() (:123)
        ratio, iv, _flags = oracle.surface_batch(p, LOGM_GRID, T_GRID, spot, r, q, "B")
        valid = ~np.isnan(iv).any(axis=(1, 2))             # success = no NaN at all (:180)
        static = np.full(len(p), -1)
        calendar = np.full(len(p), -1)
        free = np.zeros(len(p), dtype=bool)
        for i in np.nonzero(valid)[0]:
            res = ns["check_all_arbitrage"](iv[i], T_GRID, LOGM_GRID, spot, r, q)
            if res is not None:
                static[i] = res["static_violations"]
                calendar[i] = res["calendar_violations"]
                free[i] = bool(res["is_arbitrage_free"])
        # A surface is FRAGILE if one of its price ratios is below NOISE_LEVEL: deep out of the money
        # the true price is zero to double precision and the quadrature returns its own rounding
        # noise (a few 1e-15 of either sign), so whether such a point has an implied vol -- and
        # with it whether the surface counts as valid -- is decided by that noise and differs
        # between any two double-precision evaluations of the same formula.
        fragile = (np.abs(ratio) < NOISE_LEVEL).any(axis=(1, 2))
        n_valid = int(valid.sum())
        stats.append([len(p), n_valid, int(free.sum()), float(np.mean(static[valid])) if n_valid else 0.0,
                      float(np.mean(calendar[valid])) if n_valid else 0.0])
        names.append(name)
        params.append(p32)
        for k, v in (("valid", valid), ("static", static), ("calendar", calendar), ("free", free), ("fragile", fragile)):
            per[k].append(v)
        print(name, stats[-1], "fragile", int(fragile.sum()),
              "kappa %.3g..%.3g rho %.3g..%.3g" % (p[:, 0].min(), p[:, 0].max(), p[:, 3].min(), p[:, 3].max()))
    np.savez_compressed(f"{HERE}/regime_fixture.npz", names=np.array(names), params=np.stack(params),
                        stats=np.array(stats), stats_columns=np.array(["n_total", "n_valid", "arbitrage_free_count",
                                                                       "avg_static_violations", "avg_calendar_violations"]),
                        valid=np.stack(per["valid"]), static=np.stack(per["static"]), calendar=np.stack(per["calendar"]),
                        free=np.stack(per["free"]), fragile=np.stack(per["fragile"]), noise_level=NOISE_LEVEL,
                        T_grid=T_GRID, logm_grid=LOGM_GRID, spot=spot, r=r, q=q)


if __name__ == "__main__":
    main()
