"""The slate-resident FC + ListNet training step (csrc/ltrx_fcstep.hip, ltrx_fc_listnet_step) against the fp64 oracle (-m gpu).

What it replaces: FCModel.forward (allrank/models/model.py:35-44), OutputLayer.forward (model.py:111-117), listNet
(allrank/models/losses/listNet.py:8-30), their autograd backward and torch.optim.Adam.step (allrank/training/train_utils.py:18-29),
for BASELINE configs[1].  Checked at every step, at the engine's weights: loss within 1e-5, scores within 2e-5 of their scale,
d loss / d scores within 1e-4, every parameter gradient within 1e-3 of its tensor's largest entry (on the engine's ReLU branch), the
Adam update against an fp64 replica driven by the engine's gradients (3e-7); on the first step also against the GEMM launch sequence
of the same trainer (FusedTrainer(fc_step=False)).  Shapes cover ragged / one-item / fully padded slates, slate lengths and feature
counts at and off the tile boundaries (L = 16 ... 256, F = 20 ... 144, H = 16 ... 96 including H % 16 != 0), several slates per
workgroup (B > 256), and both activations.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from oracle import model_oracle as M

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg, seed):
    from allrank_amd.model import make_model
    params = M.init_params(cfg, seed=seed)
    fc = dict(sizes=list(cfg["fc_sizes"]), input_norm=False, activation=cfg.get("fc_activation"), dropout=0.0)
    model = make_model(fc, None, dict(d_output=1, output_activation=None), cfg["n_features"])
    model.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=True)
    return model.to(DEV), params


def _batch(rng, B, L, F, ragged):
    x = rng.standard_normal((B, L, F)).astype(np.float32)
    y = rng.choice(5, size=(B, L), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
    for b, n in ragged:
        if b < B and n < L:
            y[b, n:] = -1
            x[b, n:] = 0
    return x, y


def _run(B, L, F, H, act, seed, steps=3, optimizer="Adam", weight_decay=0.0, clip=None, global_batch=None, fc_step=True):
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=F, fc_sizes=[H], fc_activation=act, fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    m1, params = _build(cfg, seed)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(seed + 1)
    x, y = _batch(rng, B, L, F, [(1, L // 2), (0, 1), (2, L - 1), (3, 0), (5, 3)])      # slate 3: fully padded
    mask = y == -1
    xt, yt = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    kw = dict(lr=1e-3, use_graph=False, optimizer=optimizer, weight_decay=weight_decay, gradient_clipping_norm=clip)
    f1 = FusedTrainer(m1, "listNet", {}, B, L, fc_step=fc_step, **kw)
    cross = (H % 4 == 0)                       # (the GEMM launch sequence needs H % 4 == 0)
    f2 = FusedTrainer(m2, "listNet", {}, B, L, fc_step=False, **kw) if cross else None
    assert f1.fcstep == fc_step and (f2 is None or not f2.fcstep)
    f1.keep_fc_out = f1.keep_loss_grad = True
    keys = list(params)
    n1, n2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    worst = dict(loss=0.0, score=0.0, grad=0.0, w=0.0, oloss=0.0, oscore=0.0, ograd=0.0, dsc=0.0)
    div = float(global_batch or B)
    for st in range(steps):
        w_before = {k: n1[k].detach().cpu().numpy().astype(np.float64) for k in keys}
        l1 = float(f1.step(xt, yt, global_batch=global_batch).item())
        s1 = f1.scores.cpu().numpy().astype(np.float64)
        sc = max(1.0, float(np.abs(s1[~mask]).max()))
        if cross and st == 0:
            l2 = float(f2.step(xt, yt, global_batch=global_batch).item())
            s2 = f2.scores.cpu().numpy().astype(np.float64)
            worst["loss"] = max(worst["loss"], abs(l1 - l2) / (1 + abs(l2)))
            worst["score"] = max(worst["score"], float(np.abs(s1 - s2)[~mask].max()) / sc)
        so, cache = M.forward(w_before, cfg, x.astype(np.float64), mask)
        # the oracle's fully padded slate is NaN (the reference's too); the engine defines its contribution as 0
        ok_rows = ~mask.all(1)
        lo_rows, gs = O.listnet(so[ok_rows], y[ok_rows], dtype=np.float64)[:2]
        lo = float(lo_rows) * ok_rows.sum() / div
        gfull = np.zeros_like(so)
        gfull[ok_rows] = np.asarray(gs, dtype=np.float64) * ok_rows.sum() / div
        fc_pats = [(f1.fc_out[0] > 0).view(B, L, -1).cpu().numpy()] if act == "ReLU" else None      # (collapse: act is None)
        g_or = M.backward(w_before, cfg, cache, gfull, relu_masks=[], fc_relu_masks=fc_pats)
        worst["oloss"] = max(worst["oloss"], abs(l1 - lo) / (1 + abs(lo)))
        worst["oscore"] = max(worst["oscore"], float(np.abs(s1 - so)[~mask].max()) / sc)
        gk = f1.loss.grad.cpu().numpy().astype(np.float64)
        gs_e = np.zeros_like(so)
        gs_e[ok_rows] = np.asarray(O.listnet(s1[ok_rows], y[ok_rows], dtype=np.float64)[1], dtype=np.float64) * ok_rows.sum() / div
        worst["dsc"] = max(worst["dsc"], float(np.abs(gk - gs_e).max()) / max(float(np.abs(gs_e).max()), 1e-30))
        assert np.all(gk[mask] == 0.0)                       # exactly 0 at padded slots
        g_eng = {}
        gm = max(float(np.abs(g_or[kk]).max()) for kk in keys)
        for k in keys:
            g1 = n1[k].grad.cpu().numpy().astype(np.float64)
            g_eng[k] = g1
            own = max(float(np.abs(g_or[k]).max()), 1e-30)
            if own > 1e-6 * gm:
                if cross and st == 0:
                    g2 = n2[k].grad.cpu().numpy().astype(np.float64)
                    worst["grad"] = max(worst["grad"], float(np.abs(g1 - g2).max()) / own)
                worst["ograd"] = max(worst["ograd"], float(np.abs(g1 - g_or[k]).max()) / own)
            else:
                worst["ograd"] = max(worst["ograd"], float(np.abs(g1 - g_or[k]).max()) / gm)
        # the optimizer update (the reducing launch's fused Adam / AdamW, or -- with clipping -- clip + flat-buffer kernel) against an
        # fp64 replica of torch.optim's rule driven by the ENGINE's gradients (a cross-path comparison of WEIGHTS is meaningless: the
        # first Adam update is lr * sign(g), so entries whose gradient is below its round-off may take either sign)
        if st == 0:
            adam = M.Adam({k: v.astype(np.float64) for k, v in params.items()}, lr=1e-3, weight_decay=weight_decay,
                          decoupled=(optimizer == "AdamW"))
        g_use = g_eng
        if clip:                                               # clip_grad_norm_ (train_utils.py:24-25): g * min(1, c / (||g|| + 1e-6))
            tot = float(np.sqrt(sum(float((g ** 2).sum()) for g in g_eng.values())))
            g_use = {k: g * min(1.0, clip / (tot + 1e-6)) for k, g in g_eng.items()}
        w_pred = {k: v.copy() for k, v in w_before.items()}
        adam.step(w_pred, g_use)
        worst["w"] = max(worst["w"], max(float(np.abs(w_pred[k] - n1[k].detach().cpu().numpy()).max()) for k in keys))
    return worst


def _assert(worst, what):
    assert worst["oloss"] <= 1e-5, (what, worst)
    assert worst["oscore"] <= 2e-5, (what, worst)
    assert worst["ograd"] <= 1e-3, (what, worst)
    assert worst["dsc"] <= 1e-4, (what, worst)
    assert worst["loss"] <= 1e-5 and worst["score"] <= 2e-5 and worst["grad"] <= 1e-3, (what, worst)
    assert worst["w"] <= 3e-7, (what, worst)


CASES = [(6, 16, 20, 16, None), (6, 100, 64, 48, "ReLU"), (7, 240, 136, 96, None), (7, 240, 136, 96, "ReLU"), (6, 256, 144, 96, "ReLU"),
         (9, 37, 136, 80, None), (6, 129, 128, 33, "ReLU"), (6, 240, 132, 96, None), (6, 17, 8, 96, "ReLU"), (8, 200, 48, 64, None)]


@pytest.mark.parametrize("B,L,F,H,act", CASES)
def test_fc_listnet_step_matches_fp64_oracle_and_gemm_path(B, L, F, H, act):
    _assert(_run(B, L, F, H, act, seed=100 + L + F), (B, L, F, H, act))


@pytest.mark.parametrize("B,L,F,H", [(6, 16, 20, 16), (7, 240, 136, 96), (6, 256, 144, 96), (9, 37, 136, 80), (6, 129, 128, 33), (6, 17, 8, 96),
                                     (300, 240, 136, 96), (700, 240, 136, 96)])
def test_fc_linear_listnet_step_matches_fp64_oracle_and_gemm_path(B, L, F, H):
    """the opt-in linear-scorer form (FC activation None; ltrx_fc_linear_listnet_step): one matrix-vector product per slate and the exact
    rank-1 gradients -- same bars as the two-layer evaluation, incl. the fused optimizer update and B > 256 (several slates per
    workgroup, register double-buffering)"""
    _assert(_run(B, L, F, H, None, seed=300 + L + F, fc_step="collapse"), (B, L, F, H, "collapse"))
    if B == 7:
        _assert(_run(B, L, F, H, None, seed=41, fc_step="collapse", optimizer="AdamW", weight_decay=0.01), "collapse AdamW")
        _assert(_run(B, L, F, H, None, seed=42, fc_step="collapse", clip=0.05, global_batch=20), "collapse clip + divisor")


def test_fc_linear_collapse_is_only_taken_for_a_linear_scorer():
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=136, fc_sizes=[96], fc_activation="ReLU", fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    m, _p = _build(cfg, 3)
    ft = FusedTrainer(m, "listNet", {}, 4, 240, lr=1e-3, use_graph=False, fc_step="collapse")
    assert ft.fcstep is True                                   # ReLU: the two-layer MFMA kernel


@pytest.mark.parametrize("B,act", [(300, None), (700, "ReLU")])
def test_fc_listnet_step_several_slates_per_workgroup(B, act):
    """more slates than compute units: a workgroup walks several slates and writes ONE partial gradient"""
    _assert(_run(B, 240, 136, 96, act, seed=7 + B), (B, act))


def test_fc_listnet_step_sharded_divisor_and_other_update_rules():
    """batch_divisor = the global batch (sharded runs, short last batches); AdamW / weight decay in the reducing launch; gradient
    clipping takes the gradients-only form + the flat-buffer optimizer kernels"""
    _assert(_run(6, 240, 136, 96, None, seed=31, global_batch=24), "global_batch")
    _assert(_run(6, 240, 136, 96, "ReLU", seed=32, optimizer="AdamW", weight_decay=0.01), "AdamW")
    _assert(_run(6, 240, 136, 96, None, seed=33, optimizer="Adam", weight_decay=0.01), "Adam+L2")
    _assert(_run(6, 240, 136, 96, None, seed=34, clip=0.05), "clip")


def test_fc_listnet_step_is_deterministic_and_leaves_inputs_untouched():
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=136, fc_sizes=[96], fc_activation="ReLU", fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    rng = np.random.default_rng(5)
    x, y = _batch(rng, 300, 240, 136, [(1, 100), (2, 1)])
    xt, yt = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    res = []
    for _ in range(2):
        m, _p = _build(cfg, 9)
        ft = FusedTrainer(m, "listNet", {}, 300, 240, lr=1e-3, use_graph=False)
        assert ft.fcstep
        for _s in range(3):
            loss = ft.step(xt, yt)
        res.append((loss.clone(), ft.scores.clone(), ft.flat_g.clone(), ft.flat_p.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)                              # fixed-order partial sums: bit-identical runs
    assert torch.equal(xt.cpu(), torch.tensor(x)) and torch.equal(yt.cpu(), torch.tensor(y))


def test_fc_listnet_unsupported_shapes_take_the_gemm_path_or_raise():
    from allrank_amd import _lib as LB
    from allrank_amd.engine import FusedTrainer
    lib = LB.lib()
    assert lib.ltrx_fc_listnet_supported(240, 136, 96) == 1
    assert lib.ltrx_fc_listnet_supported(257, 136, 96) == 0 and lib.ltrx_fc_listnet_supported(240, 148, 96) == 0
    assert lib.ltrx_fc_listnet_supported(240, 136, 112) == 0 and lib.ltrx_fc_listnet_supported(240, 134, 96) == 0
    cfg = dict(n_features=136, fc_sizes=[128], fc_activation=None, fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    m, _p = _build(cfg, 3)
    ft = FusedTrainer(m, "listNet", {}, 4, 240, lr=1e-3, use_graph=False)
    assert not ft.fcstep                                      # H = 128 > 96: the GEMM launch sequence
    x, y = _batch(np.random.default_rng(1), 4, 240, 136, [])
    assert torch.isfinite(ft.step(torch.tensor(x, device=DEV), torch.tensor(y, device=DEV))).all()


def test_score_after_fcstep_uses_the_updated_weights_and_load_state_dict_is_seen():
    """the GEMM forward of score() reads pre-split weight images; they must follow (a) the slate-resident step's own
    updates and (b) an external load_state_dict into a live trainer"""
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=136, fc_sizes=[96], fc_activation="ReLU", fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    m, _p = _build(cfg, 11)
    x, y = _batch(np.random.default_rng(2), 8, 240, 136, [(1, 50)])
    xt, yt = torch.tensor(x, device=DEV), torch.tensor(y, device=DEV)
    ft = FusedTrainer(m, "listNet", {}, 8, 240, lr=1e-2, use_graph=False)
    for _ in range(3):
        ft.step(xt, yt)
    m.eval()
    with torch.no_grad():
        ref = m.score(xt, yt == -1, None)
    got = ft.score(xt, yt).clone()
    assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    m2, _p2 = _build(cfg, 12)
    m.load_state_dict(m2.state_dict())
    with torch.no_grad():
        ref2 = m.score(xt, yt == -1, None)
    got2 = ft.score(xt, yt).clone()
    assert float((got2 - ref2).abs().max()) <= 2e-5 * max(1.0, float(ref2.abs().max()))
    assert float((ref2 - ref).abs().max()) > 1e-2            # (the two weight sets really differ)
