"""Generate tests/golden/extra_golden.npz from the REAL reference (CPU, build container only):
    python tests/golden/make_golden_extra.py
Covers SURVEY.md section 8f row 4: rankNet (3 weightings), bce, ordinal, pointwise_rmse, binary_listNet, mrr and the
STOCHASTIC neuralNDCG / neuralNDCG_transposed.  The Gumbel noise the reference draws inside sample_gumbel
(loss_utils.py:70-81) is recorded and replayed: sample_gumbel is replaced by a function returning the recorded draw, so the
engine and the oracle can be fed the identical perturbation (``gumbel=`` argument).

bce / ordinal: torch >= 2.x rejects BCELoss targets outside [0, 1], so on this container's torch 2.10 the reference itself
raises as soon as a slate is padded (target -1) -- the reference pins torch 1.13.1 (Dockerfile:15), which accepted them and
masked the result afterwards (bce.py:24-25).  Two fixtures per case therefore: "*.nopad" runs the untouched reference on
the same inputs with the padding turned into ordinary items (real torch BCELoss, including its -100 log clamp and 1e-12
gradient clamp at p in {0, 1}); "*.pad" runs the reference on the padded inputs with torch.nn.functional.
binary_cross_entropy replaced by the formula torch 1.13 evaluated (no target check)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402
from tests.golden.make_inputs import make_inputs  # noqa: E402
from tests.cases import STOCH, N_ORD  # noqa: E402

SHAPES = [(3, 7, 21, False), (5, 40, 22, False), (3, 130, 23, False), (4, 33, 24, True)]
N_SAMPLES = 3


def ref_loss(fn, s, y, **kw):
    sp = torch.tensor(s, requires_grad=True)
    loss = fn(sp, torch.tensor(y), **kw)
    if loss.requires_grad:
        loss.backward()
        g = sp.grad.numpy().copy()
    else:
        g = np.zeros_like(s)
    return np.float32(loss.item()), g


def _bce_torch113(input, target, weight=None, size_average=None, reduce=None, reduction="mean"):
    l = -(target * torch.clamp(torch.log(input), min=-100) + (1 - target) * torch.clamp(torch.log(1 - input), min=-100))
    return l if reduction == "none" else (l.mean() if reduction == "mean" else l.sum())


class _Torch113BCE(object):
    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.orig = F, F.binary_cross_entropy
        F.binary_cross_entropy = _bce_torch113

    def __exit__(self, *a):
        self.F.binary_cross_entropy = self.orig
        return False


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def build():
    load_reference(stable_sort=True)
    from allrank.models import losses as RL, metrics as RM
    from allrank.models.losses import loss_utils as LU
    out = {"n_cases": np.int64(len(SHAPES)), "n_samples": np.int64(N_SAMPLES), "n_ord": np.int64(N_ORD)}
    for ci, (B, L, seed, ties) in enumerate(SHAPES):
        s, y = make_inputs(B, L, seed, ties)
        pre = "c%d." % ci
        out[pre + "s"], out[pre + "y"] = s, y
        for mode, kw in enumerate((dict(), dict(weight_by_diff=True), dict(weight_by_diff_powed=True))):
            out[pre + "ranknet.m%d.loss" % mode], out[pre + "ranknet.m%d.grad" % mode] = ref_loss(RL.rankNet, s, y, **kw)
        p = sigmoid(s)
        out[pre + "p"] = p
        yb = np.where(y == -1, -1, (y >= 2).astype(np.float32)).astype(np.float32)
        out[pre + "yb"] = yb
        ynp, ybnp = np.where(y == -1, 1, y).astype(np.float32), np.where(yb == -1, 1, yb).astype(np.float32)
        out[pre + "ynp"], out[pre + "ybnp"] = ynp, ybnp
        pe = p.copy()
        pe[0, 0], pe[0, 1] = 0.0, 1.0                             # the log clamp at -100 / the 1e-12 gradient clamp
        out[pre + "pe"] = pe
        out[pre + "bce.nopad.loss"], out[pre + "bce.nopad.grad"] = ref_loss(RL.bce, pe, ybnp)
        rng = np.random.default_rng(seed + 7)
        p3 = sigmoid(rng.standard_normal((B, L, N_ORD)).astype(np.float32) * 2)
        out[pre + "p3"] = p3
        p3e = p3.copy()
        p3e[0, 0, 0], p3e[0, 0, 1] = 0.0, 1.0
        out[pre + "p3e"] = p3e
        out[pre + "ordinal.nopad.loss"], out[pre + "ordinal.nopad.grad"] = ref_loss(RL.ordinal, p3e, ynp, n=N_ORD)
        with _Torch113BCE():
            out[pre + "bce.pad.loss"], out[pre + "bce.pad.grad"] = ref_loss(RL.bce, p, yb)
            out[pre + "ordinal.pad.loss"], out[pre + "ordinal.pad.grad"] = ref_loss(RL.ordinal, p3, y, n=N_ORD)
        out[pre + "rmse.loss"], out[pre + "rmse.grad"] = ref_loss(RL.pointwise_rmse, p, y, no_of_levels=4)
        out[pre + "blistnet.loss"], out[pre + "blistnet.grad"] = ref_loss(RL.binary_listNet, s, yb)
        ats = [1, 3, 10, 1000]
        out[pre + "mrr.ats"] = np.asarray(ats, np.int64)
        out[pre + "mrr.val"] = RM.mrr(torch.tensor(s), torch.tensor(y), ats=ats).numpy()
        out[pre + "mrr.none"] = RM.mrr(torch.tensor(s), torch.tensor(y)).numpy()
        yz = np.where(y == -1, -1, 0).astype(np.float32)           # all maxima 0 -> the batch-level zeroing
        out[pre + "mrr.zero"] = RM.mrr(torch.tensor(s), torch.tensor(yz), ats=ats).numpy()
        # ---- stochastic NeuralSort with a recorded Gumbel draw ----
        g = torch.Generator().manual_seed(seed + 99)
        U = torch.rand([N_SAMPLES, B, L, 1], generator=g)
        gum = (-torch.log(-torch.log(U + 1e-10) + 1e-10))
        out[pre + "gumbel"] = gum.numpy()
        orig = LU.sample_gumbel
        LU.sample_gumbel = lambda shape, device, eps=1e-10, _g=gum: _g
        try:
            for si, c in enumerate(STOCH):
                fn = RL.neuralNDCG_transposed if c["tr"] else RL.neuralNDCG
                key = pre + "stoch%d" % si
                out[key + ".loss"], out[key + ".grad"] = ref_loss(fn, s, y, temperature=c["tau"], k=c["k"],
                                                                   powered_relevancies=c["pw"], stochastic=True,
                                                                   n_samples=N_SAMPLES, beta=c["beta"], log_scores=c["log"])
        finally:
            LU.sample_gumbel = orig
    return {"extra_golden.npz": out}


def main():
    for f, d in build().items():
        np.savez_compressed(os.path.join(HERE, f), **d)
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
