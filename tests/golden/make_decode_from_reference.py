#!/usr/bin/env python3
"""Regenerates the decoder goldens from the reference's own class and checks that they equal the
committed tests/golden/golden_decode.npz (which make_golden.py produced with plain torch.nn layers
carrying the same state dict).  Build container only: imports
/root/reference/condtional_vae/cvae_model.py with an empty stand-in for the absent QuantLib module
(the class only touches QuantLib inside compute_arbitrage_penalty, which is not called here), loads
the checkpoint with torch.load(weights_only=True), and runs ConditionalVAE_SingleHeston.decode
followed by the de-normalisation of generate_iv_surface_by_date.py:441-446."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.modules.setdefault("QuantLib", types.ModuleType("QuantLib"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "condtional_vae"))
    from cvae_model import ConditionalVAE_SingleHeston
    ck = torch.load(os.path.join(REF, "llm_options_assistant", "best_model_2025", "cvae_model.pt"),
                    map_location="cpu", weights_only=True)
    model = ConditionalVAE_SingleHeston(param_dim=5, conditioning_dim=8, latent_dim=4, hidden_dims=[128, 64, 32],
                                        encoder_activation="tanh", decoder_activation="relu", dropout=0.15)
    print(model.load_state_dict(ck["model_state_dict"]))
    model.eval()
    g = np.load(os.path.join(HERE, "golden_decode.npz"))
    z, c = torch.tensor(g["z"]), torch.tensor(g["c"])
    with torch.no_grad():
        norm = model.decode(z, c)
        t = norm * ck["param_std"] + ck["param_mean"]
        p = t.clone()
        p[:, 0], p[:, 1], p[:, 2], p[:, 4] = torch.exp(t[:, 0]), torch.exp(t[:, 1]), torch.exp(t[:, 2]), torch.exp(t[:, 4])
        p[:, 3] = torch.tanh(t[:, 3])
        # model.sample broadcasts one conditioning row (cvae_model.py:365-366)
        torch.manual_seed(7)
        s = model.sample(5, c[:1])
    assert np.array_equal(norm.numpy(), g["normalized"]), np.abs(norm.numpy() - g["normalized"]).max()
    assert np.array_equal(p.numpy(), g["params"]), np.abs(p.numpy() - g["params"]).max()
    assert tuple(s.shape) == (5, 5)
    np.savez_compressed(os.path.join(HERE, "golden_decode.npz"), z=g["z"], c=g["c"], normalized=norm.numpy(),
                        params=p.numpy(), source="ConditionalVAE_SingleHeston.decode of the reference, "
                        "best_model_2025/cvae_model.pt")
    print("golden_decode.npz regenerated from the reference class: identical to the committed values")


if __name__ == "__main__":
    main()
