"""Drop-in for the reference's ``cvae_model`` module.

``ConditionalVAE_SingleHeston`` keeps the constructor signature, attribute names and state-dict
layout of cvae_model.py:18-400, so ``load_state_dict`` of the reference checkpoints matches all
keys.  What changes is where the hot-path methods run:

* ``decode`` / ``sample`` in inference (no autograd graph needed) run the hand-written decoder
  kernel (``afiv_decode``) on the GPU holding the inputs;
* ``compute_arbitrage_penalty`` prices its 5-strike x 2-maturity lattice with the pricing
  kernel (``afiv_arbitrage_penalty``) instead of one QuantLib call per point.

Encoder, re-parameterisation, Feller penalty and the loss arithmetic are ordinary torch code
(training is outside the accelerated path); ``decode`` also stays differentiable torch code
whenever autograd needs a graph.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
for _p in (_ROOT, os.path.dirname(_HERE)):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import afiv_bootstrap  # noqa: E402
from heston_model_ql import cached_grid  # noqa: E402

_afiv = afiv_bootstrap.load()

_ACTIVATIONS = {"relu": nn.ReLU, "leaky_relu": lambda: nn.LeakyReLU(0.2), "tanh": nn.Tanh}

# lattice of the training-time penalty (cvae_model.py:245-246)
_PENALTY_LOGM = np.array([-0.1, -0.05, 0.0, 0.05, 0.1])
_PENALTY_TAU = np.array([0.5, 1.0])


def _stack(widths, act, dropout):
    """[Linear, BatchNorm1d, activation, Dropout] per consecutive pair of widths."""
    layers = []
    for w_in, w_out in zip(widths[:-1], widths[1:]):
        layers += [nn.Linear(w_in, w_out), nn.BatchNorm1d(w_out), act]
        if dropout > 0:
            layers.append(nn.Dropout(dropout))
    return layers


class ConditionalVAE_SingleHeston(nn.Module):
    def __init__(self, param_dim=5, conditioning_dim=8, latent_dim=4, hidden_dims=[128, 64, 32],
                 encoder_activation="tanh", decoder_activation="relu", dropout=0.15,
                 feller_penalty_weight=1.0, beta=0.1, arbitrage_penalty_weight=2.0):
        super().__init__()
        self.param_dim = param_dim
        self.conditioning_dim = conditioning_dim
        self.latent_dim = latent_dim
        self.beta = beta
        self.feller_penalty_weight = feller_penalty_weight
        self.arbitrage_penalty_weight = arbitrage_penalty_weight
        self.dropout = dropout
        self.encoder_act = self._get_activation(encoder_activation)
        self.decoder_act = self._get_activation(decoder_activation)
        self._decoder_activation_name = decoder_activation
        hidden = list(hidden_dims)

        # q(z | theta, c): 13 -> hidden... -> (mu, logvar)
        self.encoder = nn.Sequential(*_stack([param_dim + conditioning_dim] + hidden,
                                             self.encoder_act, dropout))
        self.fc_mu = nn.Linear(hidden[-1], latent_dim)
        self.fc_logvar = nn.Linear(hidden[-1], latent_dim)
        # p(theta | z, c): 12 -> reversed hidden -> 5
        self.decoder = nn.Sequential(*_stack([latent_dim + conditioning_dim] + hidden[::-1],
                                             self.decoder_act, dropout),
                                     nn.Linear(hidden[0], param_dim))
        self._hip_decoders = {}      # (device, weight versions, de-normalisation constants) -> afiv.Decoder

    def _get_activation(self, name):
        try:
            return _ACTIVATIONS[name]()
        except KeyError:
            raise ValueError(f"Unknown activation: {name}")

    # ---- encoder side: plain torch --------------------------------------------------------
    def encode(self, x, c):
        h = self.encoder(torch.cat([x, c], dim=1))
        return self.fc_mu(h), self.fc_logvar(h)

    def reparameterize(self, mu, logvar):
        std = torch.exp(0.5 * logvar)
        return mu + torch.randn_like(std) * std

    # ---- decoder ---------------------------------------------------------------------------
    def _hip_ready(self):
        """The kernel implements exactly the shipped architecture: 12 -> 32 -> 64 -> 128 -> 5,
        ReLU, eval-mode BatchNorm."""
        shapes = [tuple(m.weight.shape) for m in self.decoder if isinstance(m, nn.Linear)]
        return (shapes == [(32, 12), (64, 32), (128, 64), (5, 128)]
                and self._decoder_activation_name == "relu")

    def _decoder_handle(self, device, param_mean=None, param_std=None):
        """The device-side copy of the decoder weights, rebuilt when a weight, the device or the
        de-normalisation constants change.  Layers are collected by type, not by position: with
        ``dropout=0`` the reference's Sequential has no Dropout modules and its indices shift."""
        versions = tuple(p._version for p in self.decoder.parameters()) + \
            tuple(b._version for b in self.decoder.buffers())
        mean = np.zeros(5, np.float32) if param_mean is None else \
            np.asarray(torch.as_tensor(param_mean).detach().cpu(), dtype=np.float32).reshape(5)
        std = np.ones(5, np.float32) if param_std is None else \
            np.asarray(torch.as_tensor(param_std).detach().cpu(), dtype=np.float32).reshape(5)
        key = (str(device), versions, mean.tobytes(), std.tobytes())
        if key not in self._hip_decoders:
            if len(self._hip_decoders) >= 4:         # stale weight versions / other constants
                self._hip_decoders.clear()
            linear = [m for m in self.decoder if isinstance(m, nn.Linear)]
            norms = [m for m in self.decoder if isinstance(m, nn.BatchNorm1d)]
            sd = {}
            for idx, m in zip((0, 4, 8, 12), linear):            # the checkpoint layout afiv.Decoder reads
                sd[f"decoder.{idx}.weight"], sd[f"decoder.{idx}.bias"] = m.weight, m.bias
            for idx, m in zip((1, 5, 9), norms):
                sd[f"decoder.{idx}.weight"], sd[f"decoder.{idx}.bias"] = m.weight, m.bias
                sd[f"decoder.{idx}.running_mean"], sd[f"decoder.{idx}.running_var"] = m.running_mean, m.running_var
            self._hip_decoders[key] = _afiv.Decoder(sd, mean, std, bn_eps=norms[0].eps, device=device)
        return self._hip_decoders[key]

    def _inference_inputs(self, z, c):
        """Inputs of an inference-mode decode on the model's GPU.  CPU tensors (the reference's
        ``sample(n, c)`` default) are moved there; a model that itself lives on the CPU is an
        error: the decoder kernel has no CPU fallback."""
        where = next(self.decoder.parameters()).device
        if z.is_cuda:
            where = z.device
        if where.type != "cuda":
            raise RuntimeError("decode in inference mode runs the HIP decoder kernel and needs the model "
                               "or its inputs on a GPU (no CPU fallback); call model.to('cuda')")
        if not self._hip_ready():
            raise RuntimeError("the HIP decoder implements hidden_dims=[128, 64, 32] with ReLU only")
        return where, z.to(where).float().contiguous(), c.to(where).float().contiguous()

    def decode_parameters(self, z, c, param_mean, param_std):
        """decode followed by the de-normalisation and exp / tanh transforms of
        generate_iv_surface_by_date.py:441-446, fused in the decoder kernel.  Returns
        (params float32 [N,5], the same values widened to float64 [N,5]) on the GPU -- the second
        is what the pricing kernel takes, as the reference's ``float(x)`` widening at :286."""
        with torch.no_grad():
            where, zf, cf = self._inference_inputs(z, c)
            dec = self._decoder_handle(where, param_mean, param_std)
            out = dec.decode(zf, cf, want=("params32", "params64"))
        return out["params32"], out["params64"]

    def decode(self, z, c):
        """(z, c) -> normalised parameters (cvae_model.py:145-160)."""
        needs_graph = torch.is_grad_enabled() and (
            self.training or z.requires_grad or c.requires_grad
            or any(p.requires_grad for p in self.decoder.parameters()))
        if self.training or needs_graph:
            return self.decoder(torch.cat([z, c], dim=1))          # autograd / batch statistics
        where, zf, cf = self._inference_inputs(z, c)
        out = self._decoder_handle(where).decode(zf, cf, want=("normalized",))
        return out["normalized"].to(device=z.device, dtype=z.dtype)

    def forward(self, x, c):
        mu, logvar = self.encode(x, c)
        z = self.reparameterize(mu, logvar)
        return self.decode(z, c), mu, logvar

    # ---- penalties -------------------------------------------------------------------------
    def compute_feller_penalty(self, params_normalized, norm_mean, norm_std):
        """mean(relu(sigma_v^2 - 2 kappa theta)) on de-normalised parameters
        (cvae_model.py:179-206)."""
        p = params_normalized * norm_std + norm_mean
        kappa, theta, sigma_v = torch.exp(p[:, 0]), torch.exp(p[:, 1]), torch.exp(p[:, 2])
        return F.relu(sigma_v ** 2 - 2 * kappa * theta).mean()

    def compute_arbitrage_penalty(self, params_normalized, norm_mean, norm_std,
                                  spot=7000.0, r=0.067, q=0.0):
        """Mean static + butterfly violation (x100) on price ratios at 5 strikes x 2 maturities
        (cvae_model.py:208-298); a constant (no gradient), in the dtype/device of the input."""
        p = params_normalized * norm_std + norm_mean
        cols = [torch.exp(p[:, 0]), torch.exp(p[:, 1]), torch.exp(p[:, 2]), torch.tanh(p[:, 3]),
                torch.exp(p[:, 4])]
        # the reference passes float(<float32 element>) to the pricer: widen, do not recompute
        orig = torch.stack(cols, dim=1).detach()
        if not torch.cuda.is_available():
            raise RuntimeError("compute_arbitrage_penalty needs a GPU (no CPU fallback)")
        dev = orig.device if orig.is_cuda else torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.device(dev):
            grid = cached_grid(_PENALTY_LOGM, _PENALTY_TAU, r, q)
            acc = grid.arbitrage_penalty(orig.to(dev, torch.float64).contiguous()).cpu()
        total, count = float(acc[0]), float(acc[1])
        value = total / count if count > 0 else 0.0
        return torch.tensor(value, dtype=p.dtype, device=p.device)

    def loss_function(self, recon, x, mu, logvar, norm_mean, norm_std):
        """reconstruction + beta KL + Feller + arbitrage terms (cvae_model.py:300-345)."""
        n = x.size(0)
        recon_loss = F.mse_loss(recon, x, reduction="sum") / n
        kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / n
        feller_loss = self.compute_feller_penalty(recon, norm_mean, norm_std)
        if self.arbitrage_penalty_weight > 0:
            arbitrage_loss = self.compute_arbitrage_penalty(recon, norm_mean, norm_std)
        else:
            arbitrage_loss = torch.tensor(0.0, device=x.device)
        total = (recon_loss + self.beta * kl_loss + self.feller_penalty_weight * feller_loss
                 + self.arbitrage_penalty_weight * arbitrage_loss)
        return total, recon_loss, kl_loss, feller_loss, arbitrage_loss

    # ---- sampling --------------------------------------------------------------------------
    def sample(self, num_samples, conditioning, device="cpu"):
        """z ~ N(0, I), one conditioning row broadcast if needed (cvae_model.py:347-373).  With the
        reference's default ``device="cpu"`` the draw and the result are CPU tensors; the decoder
        itself runs on the model's GPU (see ``_inference_inputs``)."""
        with torch.no_grad():
            z = torch.randn(num_samples, self.latent_dim).to(device)
            if conditioning.shape[0] == 1 and num_samples > 1:
                conditioning = conditioning.repeat(num_samples, 1)
            return self.decode(z, conditioning.to(device))

    def generate_for_conditions(self, conditions_dict, num_samples_per_condition, cond_mean, cond_std,
                                device="cpu"):
        out = {}
        for name, values in conditions_dict.items():
            c = torch.tensor((values - cond_mean) / cond_std, dtype=torch.float32).unsqueeze(0)
            out[name] = self.sample(num_samples_per_condition, c, device)
        return out
