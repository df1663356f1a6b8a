"""Generate tests/golden/model_pe_golden.npz from the REAL reference (allegro/allRank at /root/reference) on CPU:
models with the options the shipped configs use around the encoder -- fixed and learned positional encodings fed with
``indices`` (allrank/models/positional.py:15-77), FCModel.input_norm (model.py:27), Sigmoid / Tanh output activations
(model.py:106-117) -- scores, ApproxNDCG loss and the gradient of every parameter.

    python tests/golden/make_golden_pe.py        # build container only
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402


def build():
    load_reference(stable_sort=True)
    from allrank.models import losses as RL
    from allrank.models.model import make_model
    from allrank.config import TransformerConfig, PositionalEncoding

    cfgs = [
        dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None,
             pe="fixed", max_indices=40),
        dict(n_features=24, fc_sizes=[48], fc_activation="ReLU", fc_input_norm=True, N=1, d_ff=96, h=2, output_activation="Sigmoid",
             pe="learned", max_indices=60),
        dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=True, N=1, d_ff=64, h=4, output_activation="Tanh",
             pe=None, max_indices=0),
        # ordinal configuration (reproducibility: d_output = number of relevance levels, Sigmoid, loss "ordinal"); no padded slate:
        # torch >= 2 rejects BCELoss targets of -1, so the reference itself only runs this loss on un-padded batches here
        dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=1, d_ff=64, h=4, output_activation="Sigmoid",
             pe=None, max_indices=0, d_output=4, loss="ordinal"),
    ]
    out = {"n_models": np.int64(len(cfgs))}
    for mi, cfg in enumerate(cfgs):
        torch.manual_seed(300 + mi)
        pe = PositionalEncoding(strategy=cfg["pe"], max_indices=cfg["max_indices"]) if cfg["pe"] else None
        tr = TransformerConfig(N=cfg["N"], d_ff=cfg["d_ff"], h=cfg["h"], positional_encoding=pe, dropout=0.0)
        fc = dict(sizes=list(cfg["fc_sizes"]), input_norm=cfg["fc_input_norm"], activation=cfg["fc_activation"], dropout=0.0)
        model = make_model(fc, tr, dict(d_output=cfg.get("d_output", 1), output_activation=cfg["output_activation"]), cfg["n_features"])
        with torch.no_grad():
            for _, p_ in model.named_parameters():
                if p_.dim() == 1:
                    p_.add_(0.1 * torch.randn_like(p_))
        B, L = 5, 48
        rng = np.random.default_rng(400 + mi)
        x = rng.standard_normal((B, L, cfg["n_features"])).astype(np.float32)
        y = rng.integers(0, 5, (B, L)).astype(np.float32)
        # original ranks: a random subset of 0..69 in random order (ranks >= max_indices exercise the clamp to the padding row)
        idx = np.stack([rng.permutation(70)[:L] for _ in range(B)]).astype(np.int64)
        for b in range(1, B if cfg.get("loss") != "ordinal" else 0):
            n = L - 7 * b
            y[b, n:] = -1
            x[b, n:] = 0
            idx[b, n:] = -1
        mask = y == -1
        sc = model(torch.tensor(x), torch.tensor(mask), torch.tensor(idx))
        if cfg.get("loss") == "ordinal":
            loss = RL.ordinal(sc, torch.tensor(y), n=cfg["d_output"])
        else:
            loss = RL.approxNDCGLoss(sc, torch.tensor(y))
        loss.backward()
        pre = "m%d." % mi
        for k_, v_ in cfg.items():
            out[pre + "cfg." + k_] = np.asarray("-1" if v_ is None else v_)
        out[pre + "x"], out[pre + "y"], out[pre + "indices"] = x, y, idx
        out[pre + "scores"] = sc.detach().numpy()
        out[pre + "loss"] = np.float32(loss.item())
        for n_, p_ in model.named_parameters():
            out[pre + "param." + n_] = p_.detach().numpy().copy()
            out[pre + "grad." + n_] = (p_.grad.numpy().copy() if p_.grad is not None else np.zeros_like(p_.detach().numpy()))
        for n_, b_ in model.named_buffers():
            out[pre + "buffer." + n_] = b_.detach().numpy().copy()
    return {"model_pe_golden.npz": out}


def main():
    for f, d in build().items():
        np.savez_compressed(os.path.join(HERE, f), **d)
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
