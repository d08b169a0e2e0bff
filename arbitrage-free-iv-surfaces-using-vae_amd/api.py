"""Torch-facing wrappers over the C ABI: device memory, streams and shapes only.

``Grid`` owns an ``afiv_grid`` (the maturity x strike lattice, its discount factors and the strike
tables of the 144-node quadrature); ``Decoder`` owns an ``afiv_decoder`` (the CVAE decoder
weights).  Every method takes and returns CUDA tensors, enqueues on torch's current stream and
does not synchronise.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

VARIANTS = {"A": 0, "B": 1, "A_BRENT": 2, 0: 0, 1: 1, 2: 2}

# status bits of the flag word (column 3 of the [N,4] flags tensor)
STATUS_ARB_FREE = 1        # static == 0 and calendar == 0 on a valid surface
STATUS_BUTTERFLY_FREE = 2  # no negative butterfly on the price ratios
STATUS_VALID = 4           # no NaN and no non-positive implied vol
STATUS_SUCCESS_A = 8       # fewer than 10 % NaN points (generate_iv_surface_by_date.py:321)
STATUS_SUCCESS_B = 16      # no NaN point at all (generate_regime_surfaces.py:180)


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _need(t: torch.Tensor, dtype, what: str, device: torch.device, output: bool = False) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{what} must be a torch.Tensor")
    if t.device != device:
        raise ValueError(f"{what} is on {t.device}, the grid/decoder lives on {device}")
    if t.dtype != dtype:
        raise TypeError(f"{what} must be {dtype}, got {t.dtype}")
    if output and not t.is_contiguous():
        # a contiguous copy would receive the results and be dropped
        raise ValueError(f"{what} must be contiguous: the kernels write into it in place")
    return t.contiguous()


class Grid:
    """Maturity x strike lattice bound to one GPU (see afiv_grid_create)."""

    def __init__(self, logm, tau, r: float = 0.067, q: float = 0.0, trunc_days: bool = True,
                 workspace_surfaces: int = 0, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("afiv needs a GPU: torch.cuda.is_available() is False")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.logm = np.ascontiguousarray(np.asarray(logm, dtype=np.float64).ravel())
        self.tau = np.ascontiguousarray(np.asarray(tau, dtype=np.float64).ravel())
        self.nK, self.nT = int(self.logm.size), int(self.tau.size)
        self.r, self.q = float(r), float(q)
        self.trunc_days = bool(trunc_days)
        self._h = ctypes.c_void_p(0)
        dp = ctypes.POINTER(ctypes.c_double)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_grid_create(
                ctypes.byref(self._h), self.logm.ctypes.data_as(dp), self.nK,
                self.tau.ctypes.data_as(dp), self.nT, self.r, self.q, int(self.trunc_days),
                int(workspace_surfaces))
        _lib.check(rc, "afiv_grid_create")

    @property
    def traits(self) -> dict:
        """{"half_strike": bool, "stepped": 0 | 1 | 2 | 3}: the pricing shortcuts this lattice takes
        (see afiv_grid_traits)."""
        hs, st = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(_lib.lib().afiv_grid_traits(self._h, ctypes.byref(hs), ctypes.byref(st)), "afiv_grid_traits")
        return {"half_strike": bool(hs.value), "stepped": st.value}

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            _lib.lib().afiv_grid_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: modules may already be gone
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _params(self, params: torch.Tensor) -> torch.Tensor:
        p = _need(params, torch.float64, "params", self.device)
        if p.dim() != 2 or p.shape[1] != 5:
            raise ValueError(f"params must be [N,5], got {tuple(p.shape)}")
        return p

    def _lattice(self, t: torch.Tensor, what: str) -> torch.Tensor:
        t = _need(t, torch.float64, what, self.device, output=what.endswith("out"))
        if t.dim() != 3 or t.shape[1] != self.nT or t.shape[2] != self.nK:
            raise ValueError(f"{what} must be [N,{self.nT},{self.nK}], got {tuple(t.shape)}")
        return t

    def _flags(self, t: torch.Tensor, n: int) -> torch.Tensor:
        t = _need(t, torch.int32, "flags_out", self.device, output=True)
        if tuple(t.shape) != (n, 4):
            raise ValueError(f"flags_out must be [{n},4], got {tuple(t.shape)}")
        return t

    # -- entry points ----------------------------------------------------------------------
    def price(self, params: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Price ratios C / exp(-r tau) for every (parameter set, maturity, strike): [N,nT,nK]."""
        p = self._params(params)
        n = p.shape[0]
        if out is None:
            out = torch.empty((n, self.nT, self.nK), dtype=torch.float64, device=self.device)
        else:
            out = self._lattice(out, "out")
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_price(self._h, _ptr(p), n, _ptr(out), _stream())
        _lib.check(rc, "afiv_price")
        return out

    def implied_vol(self, ratio: torch.Tensor, spot: float, variant="A",
                    out: torch.Tensor | None = None) -> torch.Tensor:
        ra = self._lattice(ratio, "ratio")
        n = ra.shape[0]
        if out is None:
            out = torch.empty_like(ra)
        else:
            out = self._lattice(out, "out")
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_implied_vol(self._h, _ptr(ra), n, float(spot), VARIANTS[variant],
                                             _ptr(out), _stream())
        _lib.check(rc, "afiv_implied_vol")
        return out

    def arb_check(self, iv: torch.Tensor, ratio: torch.Tensor | None, spot: float,
                  bs_form="A", want_mask: bool = False):
        """Flags [N,4] int32 = (static, calendar, butterfly, status); with ``want_mask`` also the
        [N,nT,nK] uint8 violation mask (bit 1 static, 2 calendar, 4 butterfly)."""
        v = self._lattice(iv, "iv")
        ra = None if ratio is None else self._lattice(ratio, "ratio")
        n = v.shape[0]
        flags = torch.empty((n, 4), dtype=torch.int32, device=self.device)
        mask = torch.empty((n, self.nT, self.nK), dtype=torch.uint8, device=self.device) \
            if want_mask else None
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_arb_check(self._h, _ptr(v), _ptr(ra), n, float(spot),
                                           VARIANTS[bs_form], _ptr(flags), _ptr(mask), _stream())
        _lib.check(rc, "afiv_arb_check")
        return (flags, mask) if want_mask else flags

    def implied_vol_checked(self, ratio: torch.Tensor, spot: float, variant="A",
                            iv_out: torch.Tensor | None = None, flags_out: torch.Tensor | None = None):
        """Implied vols and flag words in one pass (same results as implied_vol + arb_check with
        bs_form = variant).  Returns (iv, flags)."""
        ra = self._lattice(ratio, "ratio")
        n = ra.shape[0]
        iv = torch.empty_like(ra) if iv_out is None else self._lattice(iv_out, "iv_out")
        flags = torch.empty((n, 4), dtype=torch.int32, device=self.device) \
            if flags_out is None else self._flags(flags_out, n)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_implied_vol_checked(self._h, _ptr(ra), n, float(spot), VARIANTS[variant],
                                                     _ptr(iv), _ptr(flags), _stream())
        _lib.check(rc, "afiv_implied_vol_checked")
        return iv, flags

    def surface(self, params: torch.Tensor, spot: float, variant="A", keep_ratio: bool = True,
                iv_out: torch.Tensor | None = None, flags_out: torch.Tensor | None = None,
                ratio_out: torch.Tensor | None = None):
        """price -> implied vol -> checks.  Returns (iv, flags, ratio-or-None).

        With ``keep_ratio=False`` the ratios go through the grid's workspace (the grid must have
        been created with ``workspace_surfaces > 0``) and are not returned."""
        p = self._params(params)
        n = p.shape[0]
        iv = torch.empty((n, self.nT, self.nK), dtype=torch.float64, device=self.device) \
            if iv_out is None else self._lattice(iv_out, "iv_out")
        flags = torch.empty((n, 4), dtype=torch.int32, device=self.device) \
            if flags_out is None else self._flags(flags_out, n)
        ratio = None
        if keep_ratio:
            ratio = torch.empty_like(iv) if ratio_out is None else self._lattice(ratio_out, "ratio_out")
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_surface(self._h, _ptr(p), n, float(spot), VARIANTS[variant],
                                         _ptr(ratio), _ptr(iv), _ptr(flags), _stream())
        _lib.check(rc, "afiv_surface")
        return iv, flags, ratio

    def surface_static(self, params: torch.Tensor, spot: float, variant="A", *, iv_out: torch.Tensor,
                       flags_out: torch.Tensor):
        """``surface`` for callers that cannot allocate (see afiv_surface_static): outputs are the
        caller's, the price ratios pass through a fixed array inside the library -- in pieces of
        ``STATIC_SCRATCH_DOUBLES // (nT * nK)`` surfaces at most when the batch is larger; calls on
        one GPU must not overlap."""
        p = self._params(params)
        n = p.shape[0]
        iv = self._lattice(iv_out, "iv_out")
        flags = self._flags(flags_out, n)
        if iv.shape[0] != n or iv.data_ptr() != iv_out.data_ptr() or flags.data_ptr() != flags_out.data_ptr() \
                or p.data_ptr() != params.data_ptr():
            raise ValueError("surface_static takes contiguous params [N,5], iv_out [N,nT,nK] and flags_out [N,4]")
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_surface_static(self._h, _ptr(p), n, float(spot), VARIANTS[variant],
                                                _ptr(iv), _ptr(flags), _stream())
        _lib.check(rc, "afiv_surface_static")
        return iv, flags

    def arbitrage_penalty(self, params: torch.Tensor) -> torch.Tensor:
        """Device [2] doubles: sum and count of the static + butterfly penalties
        (cvae_model.py:272-285) over the batch, evaluated on this grid."""
        p = self._params(params)
        b = p.shape[0]
        scratch = torch.empty((max(b, 1), self.nT, self.nK), dtype=torch.float64, device=self.device)
        acc = torch.empty(2, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_arbitrage_penalty(self._h, _ptr(p), b, _ptr(scratch), _ptr(acc),
                                                   _stream())
        _lib.check(rc, "afiv_arbitrage_penalty")
        return acc

    def stage1_objective(self, model_ratio: torch.Tensor, params: torch.Tensor,
                         market_ratio: torch.Tensor, drift_weight: float = 1.0) -> torch.Tensor:
        """Calibration Stage-1 loss matrix [C,D] (run_single_heston_calibration.py:252-287)."""
        mr = self._lattice(model_ratio, "model_ratio")
        mk = self._lattice(market_ratio, "market_ratio")
        p = self._params(params)
        if p.shape[0] != mr.shape[0]:
            raise ValueError("params and model_ratio disagree on the number of candidates")
        out = torch.empty((mr.shape[0], mk.shape[0]), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_stage1_objective(self._h, _ptr(mr), _ptr(p), mr.shape[0], _ptr(mk),
                                                  mk.shape[0], float(drift_weight), _ptr(out), _stream())
        _lib.check(rc, "afiv_stage1_objective")
        return out


    def density_wasserstein(self, model_ratio: torch.Tensor, market_ratio: torch.Tensor) -> torch.Tensor:
        """Per-maturity density-Wasserstein distances [C,D,nT] on this (fine) grid
        (run_single_heston_calibration.py:100-151): +inf where the reference returns inf, NaN for
        maturities marked absent by a NaN market row."""
        mr = self._lattice(model_ratio, "model_ratio")
        mk = self._lattice(market_ratio, "market_ratio")
        c, d = mr.shape[0], mk.shape[0]
        out = torch.empty((c, d, self.nT), dtype=torch.float64, device=self.device)
        scratch = torch.empty(((c + d) * self.nT * (self.nK + 1),), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_density_wasserstein(self._h, _ptr(mr), c, _ptr(mk), d, _ptr(scratch),
                                                     _ptr(out), _stream())
        _lib.check(rc, "afiv_density_wasserstein")
        return out

    def stage2_objective(self, stage1_loss: torch.Tensor, dist: torch.Tensor,
                         wasserstein_weight: float = 1.0) -> torch.Tensor:
        """objective_stage2 (:342-406) from the Stage-1 loss matrix [C,D] (on the market grid) and
        the distances of ``density_wasserstein`` [C,D,nT]: base + weight * mean(finite distances),
        1e10 where the base already is."""
        finite = torch.isfinite(dist)
        cnt = finite.sum(-1)
        avg = torch.where(cnt > 0, torch.where(finite, dist, torch.zeros_like(dist)).sum(-1) / cnt.clamp(min=1),
                          torch.zeros_like(stage1_loss))
        total = stage1_loss + wasserstein_weight * avg
        return torch.where(stage1_loss >= 1e10, torch.full_like(total, 1e10), total)


ENSEMBLE_ONE_SWEEP = 16384     # afiv.h: larger ensembles take several sweeps over the stack (no size limit)
STATIC_SCRATCH_DOUBLES = 4 * 1024 * 1024


def ensemble_stats(iv: torch.Tensor, select: torch.Tensor | None = None, groups: int | None = None) -> torch.Tensor:
    """nan-aware mean, median, 5th and 95th percentile over the leading (sample) axis of a stack
    of surfaces [N, ...] -> [4, ...] (generate_iv_surface_by_date.py:474-478; NumPy's default
    percentile method).  ``select`` is an optional [N] bool / uint8 tensor of the surfaces to
    include.  With ``groups`` the stack is that many consecutive ensembles of N / groups surfaces
    (the samples of each date of a batch over dates) and the result is [groups, 4, ...]."""
    v = iv.contiguous()
    if v.dtype != torch.float64 or not v.is_cuda or v.dim() < 2:
        raise TypeError("iv must be a CUDA float64 tensor [N, ...]")
    n = v.shape[0]
    points = v[0].numel()
    if n < 1 or points < 1:
        raise ValueError(f"ensemble_stats needs at least one surface and one grid point, got {tuple(v.shape)}")
    if groups is not None and (groups < 1 or n % groups):
        raise ValueError(f"{n} surfaces do not split into {groups} equal groups")
    sel = None
    if select is not None:
        sel = select.to(device=v.device, dtype=torch.uint8).contiguous()
        if sel.shape != (n,):
            raise ValueError(f"select must be [{n}], got {tuple(sel.shape)}")
    shape = (4,) + tuple(v.shape[1:])
    out = torch.empty(shape if groups is None else (groups,) + shape, dtype=torch.float64, device=v.device)
    with torch.cuda.device(v.device):
        if groups is None:
            rc = _lib.lib().afiv_ensemble_stats(_ptr(v), _ptr(sel), n, points, _ptr(out), _stream())
        else:
            rc = _lib.lib().afiv_ensemble_stats_groups(_ptr(v), _ptr(sel), groups, n // groups, points, _ptr(out),
                                                       _stream())
    _lib.check(rc, "afiv_ensemble_stats")
    return out


MATH_OPS = {"exp": 0, "sincos": 1, "log1p": 2, "atan2": 3, "rcp": 4, "ndtr": 5, "expm1": 6, "cosm1": 7,
            "exp_tab": 8, "expm1_tab": 9}


def debug_math(op: str, x: torch.Tensor, y: torch.Tensor | None = None):
    """Element-wise evaluation of one of the kernels' own elementary functions (unit tests)."""
    x = x.contiguous()
    if x.dtype != torch.float64 or not x.is_cuda:
        raise TypeError("x must be a CUDA float64 tensor")
    out0 = torch.empty_like(x)
    two = op in ("sincos", "expm1", "cosm1", "expm1_tab")
    out1 = torch.empty_like(x) if two else None
    if y is not None:
        y = y.contiguous()
    with torch.cuda.device(x.device):
        rc = _lib.lib().afiv_debug_math(MATH_OPS[op], _ptr(x), _ptr(y), x.numel(), _ptr(out0), _ptr(out1),
                                        _stream())
    _lib.check(rc, "afiv_debug_math")
    return (out0, out1) if two else out0


_DEC_LINEAR = (0, 4, 8, 12)
_DEC_BN = (1, 5, 9)


class Decoder:
    """CVAE decoder weights on one GPU (see afiv_decoder_create)."""

    def __init__(self, state_dict, param_mean, param_std, prefix: str = "decoder.",
                 bn_eps: float = 1e-5, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("afiv needs a GPU: torch.cuda.is_available() is False")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        keep = []   # host arrays must outlive the create call

        def arr(x):
            if isinstance(x, torch.Tensor):
                x = x.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
            keep.append(a)
            return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))

        w = _lib.DecoderWeights()
        shapes = ((32, 12), (64, 32), (128, 64), (5, 128))
        for i, idx in enumerate(_DEC_LINEAR):
            wt = state_dict[f"{prefix}{idx}.weight"]
            if tuple(wt.shape) != shapes[i]:
                raise ValueError(f"{prefix}{idx}.weight has shape {tuple(wt.shape)}, expected {shapes[i]}")
            w.lin_w[i] = arr(wt)
            w.lin_b[i] = arr(state_dict[f"{prefix}{idx}.bias"])
        for i, idx in enumerate(_DEC_BN):
            w.bn_w[i] = arr(state_dict[f"{prefix}{idx}.weight"])
            w.bn_b[i] = arr(state_dict[f"{prefix}{idx}.bias"])
            w.bn_mean[i] = arr(state_dict[f"{prefix}{idx}.running_mean"])
            w.bn_var[i] = arr(state_dict[f"{prefix}{idx}.running_var"])
        w.param_mean = arr(param_mean)
        w.param_std = arr(param_std)
        w.bn_eps = bn_eps
        self._h = ctypes.c_void_p(0)
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_decoder_create(ctypes.byref(self._h), ctypes.byref(w))
        _lib.check(rc, "afiv_decoder_create")

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            _lib.lib().afiv_decoder_destroy(h)
            h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown: modules may already be gone
            pass

    def decode(self, z: torch.Tensor, c: torch.Tensor, want=("normalized", "params32", "params64")):
        """Returns a dict with the requested outputs:
        normalized [N,5] f32 (raw decoder output), params32 [N,5] f32 and params64 [N,5] f64
        (kappa, theta, sigma_v, rho, v0 after de-normalisation and exp/tanh)."""
        z = _need(z, torch.float32, "z", self.device)
        c = _need(c, torch.float32, "c", self.device)
        if z.dim() != 2 or z.shape[1] != 4:
            raise ValueError(f"z must be [N,4], got {tuple(z.shape)}")
        n = z.shape[0]
        if c.dim() != 2 or c.shape[1] != 8 or c.shape[0] not in (1, n):
            raise ValueError(f"c must be [N,8] or [1,8], got {tuple(c.shape)}")
        out = {}
        if "normalized" in want:
            out["normalized"] = torch.empty((n, 5), dtype=torch.float32, device=self.device)
        if "params32" in want:
            out["params32"] = torch.empty((n, 5), dtype=torch.float32, device=self.device)
        if "params64" in want:
            out["params64"] = torch.empty((n, 5), dtype=torch.float64, device=self.device)
        if not out:
            raise ValueError("nothing requested")
        with torch.cuda.device(self.device):
            rc = _lib.lib().afiv_decode(self._h, _ptr(z), _ptr(c), n, c.shape[0],
                                        _ptr(out.get("normalized")), _ptr(out.get("params32")),
                                        _ptr(out.get("params64")), _stream())
        _lib.check(rc, "afiv_decode")
        return out
