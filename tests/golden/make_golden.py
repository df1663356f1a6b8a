"""Generate tests/golden/*.npz from the REAL reference (allegro/allRank at /root/reference) on CPU.

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_golden.py
The fixtures hold seeded inputs plus the reference's own outputs: loss values, autograd gradients
w.r.t. y_pred, NDCG@k, stable sort indices, and for a small model the scores and every parameter
gradient.  tests/test_oracle_pinned.py pins oracle/ against them on CPU; the ``-m gpu`` parity tests
pin the HIP kernels against them on the MI355X.  Reference sorts run with stable=True (tie policy,
SURVEY.md §9.2); listMLE's torch.randperm (listMLE.py:17) is replaced by the recorded permutation.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402
from tests.golden.make_inputs import make_inputs  # noqa: E402

LAMBDA_SCHEMES = [None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme",
                  "rankNet_scheme", "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme"]


def ref_loss(fn, s, y, **kw):
    sp = torch.tensor(s, requires_grad=True)
    loss = fn(sp, torch.tensor(y), **kw)
    if loss.requires_grad:
        loss.backward()
        g = sp.grad.numpy().copy()
    else:
        g = np.zeros_like(s)
    return np.float32(loss.item()), g


def build():
    """{file name: {key: array}} -- what main() writes; tests/test_golden_drift.py regenerates and compares"""
    load_reference(stable_sort=True)
    from allrank.models import losses as RL, metrics as RM
    from allrank.models.model import make_model
    from allrank.config import TransformerConfig

    out = {}
    shapes = [(3, 7, 11, False), (5, 40, 12, False), (4, 240, 13, False), (4, 33, 14, True)]
    out["n_cases"] = np.int64(len(shapes))
    for ci, (B, L, seed, ties) in enumerate(shapes):
        s, y = make_inputs(B, L, seed, ties)
        pre = "c%d." % ci
        out[pre + "s"], out[pre + "y"] = s, y
        out[pre + "listnet.loss"], out[pre + "listnet.grad"] = ref_loss(RL.listNet, s, y)
        for a in (1.0, 2.5):
            out[pre + "approxndcg.a%g.loss" % a], out[pre + "approxndcg.a%g.grad" % a] = ref_loss(RL.approxNDCGLoss, s, y, alpha=a)
        perm = np.random.default_rng(seed + 100).permutation(L).astype(np.int64)
        out[pre + "listmle.perm"] = perm
        orig = torch.randperm
        torch.randperm = lambda n, _p=perm: torch.tensor(_p)
        try:
            out[pre + "listmle.loss"], out[pre + "listmle.grad"] = ref_loss(RL.listMLE, s, y)
        finally:
            torch.randperm = orig
        for si, sch in enumerate(LAMBDA_SCHEMES):
            for kk in (None, 5):
                for red, lg in (("sum", "binary"), ("mean", "natural")):
                    key = pre + "lambda.s%d.k%s.%s.%s" % (si, kk, red, lg)
                    out[key + ".loss"], out[key + ".grad"] = ref_loss(
                        RL.lambdaLoss, s, y, weighing_scheme=sch, k=kk, reduction=red, reduction_log=lg, sigma=1.3, mu=7.0)
        for tr in (False, True):
            fn = RL.neuralNDCG_transposed if tr else RL.neuralNDCG
            for tau in (1.0, 0.1):
                for kk in (None, 5):
                    for pw in (True, False):
                        key = pre + "neural.t%d.tau%g.k%s.p%d" % (int(tr), tau, kk, int(pw))
                        out[key + ".loss"], out[key + ".grad"] = ref_loss(fn, s, y, temperature=tau, k=kk, powered_relevancies=pw)
        ats = [1, 5, 10, 1000]
        out[pre + "ndcg.ats"] = np.asarray(ats, np.int64)
        out[pre + "ndcg.val"] = RM.ndcg(torch.tensor(s), torch.tensor(y), ats=ats).numpy()
        out[pre + "dcg.val"] = RM.dcg(torch.tensor(s), torch.tensor(y), ats=ats).numpy()
        sm = torch.tensor(s).clone()
        sm[torch.tensor(y) == -1] = float("-inf")
        out[pre + "order"] = sm.sort(descending=True, dim=-1)[1].numpy().astype(np.int64)

    # ---- model golden: scores + parameter gradients through approxNDCG for two small configs ----
    mout = {}
    cfgs = [
        dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None),
        dict(n_features=24, fc_sizes=[24, 64], fc_activation="ReLU", fc_input_norm=True, N=1, d_ff=96, h=1, output_activation="Tanh"),
        dict(n_features=20, fc_sizes=[16], fc_activation="Sigmoid", fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None),
    ]
    for mi, cfg in enumerate(cfgs):
        torch.manual_seed(100 + mi)
        tr = TransformerConfig(N=cfg["N"], d_ff=cfg["d_ff"], h=cfg["h"], positional_encoding=None, dropout=0.0) if cfg["N"] else None
        fc = dict(sizes=list(cfg["fc_sizes"]), input_norm=cfg["fc_input_norm"], activation=cfg["fc_activation"], dropout=0.0)
        model = make_model(fc, tr, dict(d_output=1, output_activation=cfg["output_activation"]), cfg["n_features"])
        with torch.no_grad():
            for _, p_ in model.named_parameters():
                if p_.dim() == 1:
                    p_.add_(0.1 * torch.randn_like(p_))
        B, L = 4, 70
        rng = np.random.default_rng(200 + mi)
        x = rng.standard_normal((B, L, cfg["n_features"])).astype(np.float32)
        y = rng.integers(0, 5, (B, L)).astype(np.float32)
        for b in range(1, B):
            y[b, L - 9 * b:] = -1
            x[b, L - 9 * b:] = 0
        mask = y == -1
        sc = model(torch.tensor(x), torch.tensor(mask), None)
        loss = RL.approxNDCGLoss(sc, torch.tensor(y))
        loss.backward()
        pre = "m%d." % mi
        for k_, v_ in cfg.items():
            mout[pre + "cfg." + k_] = np.asarray(-1 if v_ is None else v_)
        mout[pre + "x"], mout[pre + "y"] = x, y
        mout[pre + "scores"] = sc.detach().numpy()
        mout[pre + "loss"] = np.float32(loss.item())
        for n_, p_ in model.named_parameters():
            mout[pre + "param." + n_] = p_.detach().numpy().copy()
            mout[pre + "grad." + n_] = p_.grad.numpy().copy()
    mout["n_models"] = np.int64(len(cfgs))
    return {"losses_golden.npz": out, "model_golden.npz": mout}


def main():
    for f, d in build().items():
        np.savez_compressed(os.path.join(HERE, f), **d)
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
