"""-m gpu: the model options of the shipped allRank configs on the explicit training step (FusedTrainer) and on the
nn.Module path -- fixed / learned positional encodings fed with ``indices`` (allrank/models/positional.py:15-77,
transformer.py:51-52), FCModel.input_norm (model.py:27,39), Sigmoid / Tanh output activations (model.py:106-117) -- against
golden vectors produced by the reference itself (tests/golden/make_golden_pe.py, make_golden.py): scores, ApproxNDCG loss and
the gradient of EVERY parameter (incl. the learned table, whose padding row must get exactly 0); model 3 is the ordinal
configuration: OutputLayer(d_output=4, Sigmoid) trained through the ``ordinal`` loss."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pe_golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "model_pe_golden.npz"), allow_pickle=False))


def _none(v):
    v = str(v)
    return None if v == "-1" else v


def _build(g, pre):
    from allrank_amd.model import make_model
    c = lambda k: g[pre + "cfg." + k]  # noqa: E731
    pe = _none(c("pe")) if (pre + "cfg.pe") in g else None
    tr = None
    if int(c("N")):
        tr = dict(N=int(c("N")), d_ff=int(c("d_ff")), h=int(c("h")), dropout=0.0,
                  positional_encoding=dict(strategy=pe, max_indices=int(c("max_indices"))) if pe else None)
    fc = dict(sizes=[int(v) for v in np.atleast_1d(c("fc_sizes"))], input_norm=bool(c("fc_input_norm")),
              activation=_none(c("fc_activation")), dropout=0.0)
    d_out = int(c("d_output")) if (pre + "cfg.d_output") in g else 1
    model = make_model(fc, tr, dict(d_output=d_out, output_activation=_none(c("output_activation"))), int(c("n_features")))
    sd = {k[len(pre + "param."):]: torch.tensor(v) for k, v in g.items() if k.startswith(pre + "param.")}
    sd.update({k[len(pre + "buffer."):]: torch.tensor(v) for k, v in g.items() if k.startswith(pre + "buffer.")})
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return model.to(DEV)


def _loss_of(g, pre):
    """(name, kwargs) of the loss the golden model was differentiated through"""
    if (pre + "cfg.loss") in g and str(g[pre + "cfg.loss"]) == "ordinal":
        return "ordinal", dict(n=int(g[pre + "cfg.d_output"]))
    return "approxNDCGLoss", {}


def _check(g, pre, scores, loss, grads, valid, what):
    serr = float(np.abs(scores - g[pre + "scores"])[valid].max())
    assert serr < 3e-5, (what, "scores", serr)
    assert abs(loss - float(g[pre + "loss"])) <= 1e-5 * (1 + abs(float(g[pre + "loss"]))), (what, loss, float(g[pre + "loss"]))
    ref = {k[len(pre + "grad."):]: v for k, v in g.items() if k.startswith(pre + "grad.")}
    scale = max(float(np.abs(v).max()) for v in ref.values())
    for k, v in ref.items():
        err = float(np.abs(grads[k] - v).max())
        assert err <= 2e-4 * scale + 1e-8, (what, k, err, scale)
    return serr


@pytest.mark.parametrize("mi", [0, 1, 2, 3])
def test_module_path_matches_reference_with_positional_encoding_and_activations(pe_golden, mi):
    from allrank_amd import losses as E
    g, pre = pe_golden, "m%d." % mi
    model = _build(g, pre)
    x, y, idx = (torch.tensor(g[pre + k], device=DEV) for k in ("x", "y", "indices"))
    mask = y == -1
    sc = model(x, mask, idx)
    lname, largs = _loss_of(g, pre)
    loss = getattr(E, lname)(sc, y, **largs)
    loss.backward()
    grads = {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}
    _check(g, pre, sc.detach().cpu().numpy(), float(loss.item()), grads, ~mask.cpu().numpy(), "module m%d" % mi)
    if "encoder.position.pe.weight" in grads:
        assert not grads["encoder.position.pe.weight"][-1].any(), "the padding row of the learned table must get no gradient"


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("gemm", ["split_bf16", "hipblaslt"])
@pytest.mark.parametrize("mi", [0, 1, 2, 3])
def test_fused_step_matches_reference_with_positional_encoding_and_activations(pe_golden, mi, gemm, compact):
    from allrank_amd.engine import FusedTrainer
    g, pre = pe_golden, "m%d." % mi
    model = _build(g, pre)
    x, y, idx = (torch.tensor(g[pre + k], device=DEV) for k in ("x", "y", "indices"))
    B, L = y.shape
    lname, largs = _loss_of(g, pre)
    ft = FusedTrainer(model, lname, largs, B, L, lr=1e-3, use_graph=False, gemm=gemm, compact=compact)
    loss = float(ft.step(x, y, idx).item())
    grads = {n: p.grad.cpu().numpy() for n, p in model.named_parameters()}
    valid = (y != -1).cpu().numpy()
    _check(g, pre, ft.scores_raw.cpu().numpy(), loss, grads, valid, "fused m%d %s compact=%s" % (mi, gemm, compact))
    if ft.n_out > 1:                      # model.score = sum over the output units (model.py:119-128)
        assert torch.allclose(ft.scores, ft.scores_raw.sum(-1), atol=1e-6)
    if "encoder.position.pe.weight" in grads:
        assert not grads["encoder.position.pe.weight"][-1].any()
    # a second and third step run (and, without compact, the third is captured in a hipGraph): finite and decreasing-ish
    ft2 = FusedTrainer(_build(g, pre), lname, largs, B, L, lr=1e-3, use_graph=not compact, gemm=gemm, compact=compact)
    ls = [float(ft2.step(x, y, idx).item()) for _ in range(4)]
    assert np.isfinite(ls).all() and ls[-1] < ls[0] + 1e-3, ls


def test_fused_step_matches_reference_model_with_input_norm_and_tanh(model_golden):
    """model 1 of tests/golden/model_golden.npz: FC [24, 64] ReLU + input_norm + 1 encoder layer + Tanh output"""
    from allrank_amd.engine import FusedTrainer
    g, pre = model_golden, "m1."
    model = _build(g, pre)
    x, y = torch.tensor(g[pre + "x"], device=DEV), torch.tensor(g[pre + "y"], device=DEV)
    B, L = y.shape
    for gemm in ("split_bf16", "hipblaslt"):
        m = _build(g, pre)
        ft = FusedTrainer(m, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=False, gemm=gemm)
        loss = float(ft.step(x, y).item())
        grads = {n: p.grad.cpu().numpy() for n, p in m.named_parameters()}
        _check(g, pre, ft.scores.cpu().numpy(), loss, grads, (y != -1).cpu().numpy(), "model_golden m1 " + gemm)


def test_positional_encoding_requires_indices(pe_golden):
    from allrank_amd.engine import FusedTrainer
    g, pre = pe_golden, "m0."
    model = _build(g, pre)
    x, y = torch.tensor(g[pre + "x"], device=DEV), torch.tensor(g[pre + "y"], device=DEV)
    ft = FusedTrainer(model, "approxNDCGLoss", {}, y.shape[0], y.shape[1], use_graph=False)
    with pytest.raises(ValueError):
        ft.step(x, y)
