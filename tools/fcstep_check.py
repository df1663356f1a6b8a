"""Lab check of the slate-resident FC + ListNet step (csrc/ltrx_fcstep.hip) on a GPU box: for a grid of shapes, three training
steps through FusedTrainer(fc_step=True) against (a) the same trainer on the GEMM launch sequence (fc_step=False) and (b) the
fp64 oracle at the engine's weights; then timings of both paths.  Prints one line per case; exits non-zero on a parity miss.

    python tools/fcstep_check.py [--quick]
"""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ltr_oracle as O          # noqa: E402
from oracle import model_oracle as M        # noqa: E402

DEV = "cuda:0"


def build(cfg, seed):
    from allrank_amd.model import make_model
    params = M.init_params(cfg, seed=seed)
    fc = dict(sizes=list(cfg["fc_sizes"]), input_norm=False, activation=cfg.get("fc_activation"), dropout=0.0)
    model = make_model(fc, None, dict(d_output=1, output_activation=None), cfg["n_features"])
    model.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=True)
    return model.to(DEV), params


def batch(rng, B, L, F, ragged):
    x = rng.standard_normal((B, L, F)).astype(np.float32)
    y = rng.choice(5, size=(B, L), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
    for b, n in ragged:
        if b < B and n < L:
            y[b, n:] = -1
            x[b, n:] = 0
    return x, y


def case(B, L, F, H, act, seed, steps=3):
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=F, fc_sizes=[H], fc_activation=act, fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    m1, params = build(cfg, seed)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(seed + 1)
    x, y = batch(rng, B, L, F, [(1, L // 2), (0, 1), (2, L - 1), (5, 3)])
    mask = y == -1
    xt, yt = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    f1 = FusedTrainer(m1, "listNet", {}, B, L, lr=1e-3, use_graph=False, fc_step=True)
    cross = (H % 4 == 0)                      # (the GEMM launch sequence needs H % 4 == 0)
    f2 = FusedTrainer(m2, "listNet", {}, B, L, lr=1e-3, use_graph=False, fc_step=False) if cross else None
    assert f1.fcstep and (f2 is None or not f2.fcstep)
    adam = M.Adam({k: v.astype(np.float64) for k, v in params.items()}, lr=1e-3)
    f1.keep_fc_out = f1.keep_loss_grad = True
    keys = list(params)
    n1, n2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    worst = dict(loss=0.0, score=0.0, grad=0.0, w=0.0, oloss=0.0, oscore=0.0, ograd=0.0, dsc=0.0)
    for st in range(steps):
        w_before = {k: n1[k].detach().cpu().numpy().astype(np.float64) for k in keys}
        l1 = float(f1.step(xt, yt).item())
        s1 = f1.scores.cpu().numpy().astype(np.float64)
        sc = max(1.0, float(np.abs(s1[~mask]).max()))
        # the two launch sequences are compared on the FIRST step only (equal weights); afterwards Adam's lr * sign(g) on the entries
        # whose gradient is below its own round-off sends any two implementations apart
        if cross and st == 0:
            l2 = float(f2.step(xt, yt).item())
            s2 = f2.scores.cpu().numpy().astype(np.float64)
            worst["loss"] = max(worst["loss"], abs(l1 - l2) / (1 + abs(l2)))
            worst["score"] = max(worst["score"], float(np.abs(s1 - s2)[~mask].max()) / sc)
        so, cache = M.forward(w_before, cfg, x.astype(np.float64), mask)
        lo, gs = O.listnet(so, y, dtype=np.float64)[:2]
        fc_pats = [(f1.fc_out[0] > 0).view(B, L, -1).cpu().numpy()] if act == "ReLU" else None
        g_or = M.backward(w_before, cfg, cache, np.asarray(gs, dtype=np.float64), relu_masks=[], fc_relu_masks=fc_pats)
        worst["oloss"] = max(worst["oloss"], abs(l1 - float(lo)) / (1 + abs(float(lo))))
        worst["oscore"] = max(worst["oscore"], float(np.abs(s1 - so)[~mask].max()) / sc)
        gk = f1.loss.grad.cpu().numpy().astype(np.float64)
        gs_e = np.asarray(O.listnet(s1, y, dtype=np.float64)[1], dtype=np.float64)
        worst["dsc"] = max(worst["dsc"], float(np.abs(gk - gs_e).max()) / max(float(np.abs(gs_e).max()), 1e-30))
        g_eng = {}
        for k in keys:
            g1 = n1[k].grad.cpu().numpy().astype(np.float64)
            g_eng[k] = g1
            own = max(float(np.abs(g_or[k]).max()), 1e-30)
            gm = max(float(np.abs(g_or[kk]).max()) for kk in keys)
            if own > 1e-6 * gm:
                if cross and st == 0:
                    g2 = n2[k].grad.cpu().numpy().astype(np.float64)
                    worst["grad"] = max(worst["grad"], float(np.abs(g1 - g2).max()) / own)
                worst["ograd"] = max(worst["ograd"], float(np.abs(g1 - g_or[k]).max()) / own)
            else:
                worst["ograd"] = max(worst["ograd"], float(np.abs(g1 - g_or[k]).max()) / gm)
        # the Adam update applied by the reducing launch vs an fp64 replica driven by the engine's own gradients
        w_pred = {k: v.copy() for k, v in w_before.items()}
        adam.step(w_pred, g_eng)
        worst["w"] = max(worst["w"], max(float(np.abs(w_pred[k] - n1[k].detach().cpu().numpy()).max()) for k in keys))
    ok = (worst["oloss"] <= 1e-5 and worst["oscore"] <= 2e-5 and worst["ograd"] <= 1e-3 and worst["dsc"] <= 1e-4 and
          worst["loss"] <= 1e-5 and worst["score"] <= 2e-5 and worst["grad"] <= 1e-3 and worst["w"] <= 3e-7)
    print("%s B=%d L=%d F=%d H=%d act=%s " % ("ok  " if ok else "FAIL", B, L, F, H, act) +
          " ".join("%s=%.2e" % kv for kv in worst.items()), flush=True)
    return ok


def timing(B, L=240, F=136, H=96, act=None, steps=200):
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=F, fc_sizes=[H], fc_activation=act, fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    rng = np.random.default_rng(7)
    nb = 4
    x, y = batch(rng, nb * B, L, F, [])
    xt, yt = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    out = {}
    for name, kw in (("fcstep", dict(fc_step=True, use_graph=False)), ("collapse", dict(fc_step="collapse", use_graph=False)),
                     ("gemm+graph", dict(fc_step=False, use_graph=True))):
        if name == "collapse" and act is not None:
            continue
        m, _ = build(cfg, 3)
        ft = FusedTrainer(m, "listNet", {}, B, L, lr=1e-3, **kw)
        for i in range(10):
            ft.step(xt[(i % nb) * B:(i % nb + 1) * B], yt[(i % nb) * B:(i % nb + 1) * B])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            ft.step(xt[(i % nb) * B:(i % nb + 1) * B], yt[(i % nb) * B:(i % nb + 1) * B])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # GPU-side time of one step (events), host overhead excluded
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for i in range(20):
            ft.step(xt[:B], yt[:B])
        e1.record()
        torch.cuda.synchronize()
        out[name] = (dt * 1e6, e0.elapsed_time(e1) / 20 * 1e3)
    by = (4 * F + 8) * B * L
    for name, (us, gus) in out.items():
        print("time B=%d %-10s wall %.1f us/step  (events %.1f us)  %.3f G items/s  %.0f GB/s algorithmic = %.1f %% of 8 TB/s" %
              (B, name, us, gus, B * L / us / 1e3, by / us / 1e3, by / us / 1e3 / 80.0), flush=True)


def main():
    quick = "--quick" in sys.argv
    ok = True
    if "--timing-only" in sys.argv:            # e.g. --timing-only 256,2048   (for rocprofv3 runs)
        for B in [int(v) for v in sys.argv[sys.argv.index("--timing-only") + 1].split(",")]:
            timing(B, steps=50)
        return 0
    cases = [(3, 16, 20, 16, None), (5, 100, 64, 48, "ReLU"), (7, 240, 136, 96, None), (7, 240, 136, 96, "ReLU"), (4, 256, 144, 96, "ReLU"),
             (9, 37, 136, 80, None), (6, 129, 128, 33, "ReLU"), (300, 240, 136, 96, None), (600, 240, 136, 96, "ReLU")]
    if quick:
        cases = cases[:4]
    for i, c in enumerate(cases):
        ok = case(*c, seed=100 + i) and ok
    if "--no-timing" not in sys.argv:
        for B in (64, 256, 2048):
            timing(B)
        timing(2048, act="ReLU")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
