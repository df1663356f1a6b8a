"""Pins oracle/ (the CPU restatement) against (1) the literal known-answer constants of the reference's own
tests and (2) golden vectors generated from the reference itself (tests/golden/make_golden.py).  CPU only."""
import math

import numpy as np
import pytest

from oracle import ltr_oracle as O
from oracle import model_oracle as M
from tests.cases import close, grad_close, iter_loss_cases

PAD = -1.0


def _a(x):
    return np.asarray([x], dtype=np.float32)


# ---- (1) KATs copied as numbers from /root/reference/tests/losses/*.py ----------------------------------
def test_kat_approxndcg():          # tests/losses/test_approxndcg.py:10-20
    l0 = O.approxndcg(_a([0.5, 0.3, 0.5]), _a([0.5, 0.3, 0.5]))[0]
    l1 = O.approxndcg(_a([0.5, 0.3, 0.5, 1.0]), _a([0.5, 0.3, 0.5, PAD]))[0]
    assert l0 == pytest.approx(-0.8499219417) and l1 == pytest.approx(l0)


@pytest.mark.parametrize("scheme,log,expected", [
    ("ndcgLoss1_scheme", "binary", 2.9272110462),       # tests/losses/test_lambdaloss.py:10-21
    ("ndcgLoss2PP_scheme", "binary", 1.1244146823),     # :24-34
    ("rankNet_scheme", "natural", 1.1962778568),        # :37-46
])
def test_kat_lambdaloss(scheme, log, expected):
    l0 = O.lambdaloss(_a([0.5, 0.3, 0.5]), _a([0.5, 0.3, 0.5]), weighing_scheme=scheme, reduction_log=log)[0]
    l1 = O.lambdaloss(_a([0.5, 0.3, 0.5, 1.0]), _a([0.5, 0.3, 0.5, PAD]), weighing_scheme=scheme, reduction_log=log)[0]
    assert l0 == pytest.approx(expected) and l1 == pytest.approx(l0)


def test_kat_listmle():             # tests/losses/test_listmle.py:14-22 (result independent of the shuffle here)
    for perm in ([0, 1, 2], [2, 0, 1], [1, 2, 0]):
        l0 = O.listmle(_a([0.5, 0.3, 0.5]), _a([1.0, 0.0, PAD]), perm)[0]
        assert l0 == pytest.approx(0.5981389284133911)


def _softmax(v):
    v = np.asarray(v, np.float64)
    e = np.exp(v - v.max())
    return e / e.sum()


def test_kat_listnet():             # tests/losses/test_listnet.py:16-46
    r = O.listnet(_a([0.5, 0.2]), _a([1.0, 0.0]), eps=0.0)[0]
    assert r == pytest.approx(-np.sum(_softmax([1.0, 0.0]) * np.log(_softmax([0.5, 0.2]))))
    r = O.listnet(_a([0.5, -1e30]), _a([1.0, 0.0]))[0]
    assert math.isfinite(r) and r == pytest.approx(-np.sum(_softmax([1.0, 0.0]) * np.log(_softmax([0.5, -1e30]) + 1e-10)))
    r = O.listnet(_a([0.5, 0.2, 0.5]), _a([1.0, 0.0, PAD]))[0]
    assert r == pytest.approx(-np.sum(_softmax([1.0, 0.0]) * np.log(_softmax([0.5, 0.2]) + 1e-10)))


def test_kat_ndcg():                # tests/losses/test_ndcg.py:14-69 (idcg==0 -> 1.0 per metrics.py:8,24)
    assert O.ndcg(_a([0.5, 0.2]), _a([1.0, 0.0]))[0][0, 0] == 1.0
    assert O.ndcg(_a([0.5, 0.2]), _a([0.0, 1.0]))[0][0, 0] == pytest.approx(1 / math.log2(3))
    assert O.ndcg(_a([0.5, 0.2]), _a([0.0, 0.0]))[0][0, 0] == 1.0
    r = O.ndcg(_a([0.5, 0.2, 0.1]), _a([1.0, 0.0, 1.0]), ats=[1, 2])[0][0]
    assert r == pytest.approx([1.0, 1.0 / (1.0 + 1 / math.log2(3))])
    assert O.ndcg(_a([0.5, 0.2, 1.0]), _a([1.0, 0.0, PAD]))[0][0, 0] == 1.0
    assert O.ndcg(_a([0.5, 0.2, 1.0]), _a([0.0, 1.0, PAD]))[0][0, 0] == pytest.approx(1 / math.log2(3))


@pytest.mark.parametrize("transposed", [False, True])
def test_kat_neuralndcg_equals_ndcg_at_low_temperature(transposed):   # tests/losses/test_neuralndcg.py:16-94
    cases = [
        ([0.5, 0.2], [1.0, 0.0], 1e-4, None),
        ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0], 1e-4, None),
        ([0.5, -1e30], [1.0, 0.0], 1e-4, None),
        ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63, 1., 0.5, 0.3], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0, PAD, PAD, PAD], 1e-3, None),
        ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0], 1e-4, 3),
    ]
    for yp, yt, tau, k in cases:
        r = O.neuralndcg(_a(yp), _a(yt), temperature=tau, k=k, transposed=transposed)[0]
        e = O.ndcg(_a(yp), _a(yt), ats=None if k is None else [k])[0].mean()
        assert math.isfinite(r) and -r == pytest.approx(e)


# ---- (2) golden vectors from the reference itself ---------------------------------------------------------
def _run_oracle(kind, kw, s, y):
    if kind == "listnet":
        return O.listnet(s, y)[:2]
    if kind == "approxndcg":
        return O.approxndcg(s, y, **kw)[:2]
    if kind == "listmle":
        return O.listmle(s, y, kw["perm"])[:2]
    if kind == "lambdaloss":
        return O.lambdaloss(s, y, **kw)[:2]
    if kind == "neuralndcg":
        return O.neuralndcg(s, y, **kw)[:2]
    raise KeyError(kind)


def test_losses_match_reference_golden(losses_golden):
    bad = []
    n = 0
    for name, kind, kw, s, y, rl, rg in iter_loss_cases(losses_golden):
        lo, go = _run_oracle(kind, kw, s, y)
        n += 1
        if not (close(lo, rl) and grad_close(go, rg)):
            bad.append((name, float(lo), float(rl), float(np.abs(go - rg).max())))
    assert n == 208 and not bad, bad[:10]


def test_ndcg_and_sort_indices_match_reference_golden(losses_golden):
    g = losses_golden
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        ats = [int(a) for a in g[pre + "ndcg.ats"]]
        nd, order = O.ndcg(g[pre + "s"], g[pre + "y"], ats=ats)
        dc, _ = O.dcg(g[pre + "s"], g[pre + "y"], ats=ats)
        assert close(nd, g[pre + "ndcg.val"]) and close(dc, g[pre + "dcg.val"])
        nvalid = (g[pre + "y"] != -1).sum(1)
        for b in range(order.shape[0]):     # bit-exact on the valid prefix (tie policy: stable descending)
            assert np.array_equal(order[b, :nvalid[b]], g[pre + "order"][b, :nvalid[b]])


def _cfg_from_golden(g, pre):
    def val(k):
        v = g[pre + "cfg." + k]
        return v
    acts = {"ReLU": "ReLU", "Tanh": "Tanh", "Sigmoid": "Sigmoid", "-1": None}
    return dict(n_features=int(val("n_features")), fc_sizes=[int(v) for v in np.atleast_1d(val("fc_sizes"))],
                fc_activation=acts[str(val("fc_activation"))], fc_input_norm=bool(val("fc_input_norm")),
                N=int(val("N")), d_ff=int(val("d_ff")), h=int(val("h")),
                output_activation=acts[str(val("output_activation"))])


def test_model_forward_backward_match_reference_golden(model_golden):
    g = model_golden
    for mi in range(int(g["n_models"])):
        pre = "m%d." % mi
        cfg = _cfg_from_golden(g, pre)
        params = {k[len(pre + "param."):]: v for k, v in g.items() if k.startswith(pre + "param.")}
        x, y = g[pre + "x"], g[pre + "y"]
        mask = y == -1
        sc, cache = M.forward(params, cfg, x, mask)
        assert close(sc[~mask], g[pre + "scores"][~mask], rtol=1e-5, atol=2e-5)
        lo, gs, _ = O.approxndcg(sc, y)
        assert close(lo, g[pre + "loss"])
        grads = M.backward(params, cfg, cache, gs)
        allg = np.concatenate([g[pre + "grad." + k].ravel() for k in params])
        scale = np.abs(allg).max()
        for k in params:
            assert np.abs(grads[k] - g[pre + "grad." + k]).max() <= 2e-4 * scale + 1e-8, k


# ---- pointwise / pairwise losses, mrr, stochastic NeuralSort ------------------
_EXTRA = dict(ranknet=O.ranknet, bce=O.bce, ordinal=O.ordinal, pointwise_rmse=O.pointwise_rmse, binary_listnet=O.binary_listnet)


def test_extra_losses_match_reference_golden(extra_golden):
    from tests.cases import iter_extra_cases
    n = 0
    for name, kind, kw, yp, yt, rl, rg in iter_extra_cases(extra_golden):
        loss, grad = _EXTRA[kind](yp, yt, **kw)
        assert close(loss, rl), (name, loss, rl)
        assert grad_close(grad, rg), (name, float(np.abs(grad - rg).max()))
        n += 1
    assert n == 36


def test_mrr_matches_reference_golden(extra_golden):
    g = extra_golden
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        s, y = g[pre + "s"], g[pre + "y"]
        ats = [int(a) for a in g[pre + "mrr.ats"]]
        assert np.array_equal(O.mrr(s, y, ats), g[pre + "mrr.val"])
        assert np.array_equal(O.mrr(s, y), g[pre + "mrr.none"])
        assert np.array_equal(O.mrr(s, np.where(y == -1, -1, 0).astype(np.float32), ats), g[pre + "mrr.zero"])


def test_kat_mrr_and_ranknet():
    # metrics.py:80-113 by hand: best label at predicted rank 1 -> 1/2; cut-off 1 -> 0
    r = O.mrr(_a([0.9, 0.5, 0.7]), _a([0.0, 0.0, 1.0]), ats=[1, 2, 3])
    assert np.allclose(r, [[0.0, 0.5, 0.5]])
    # rankNet on one pair: log(1 + exp(-(s_i - s_j)))
    l, g_ = O.ranknet(_a([0.2, 1.0]), _a([1.0, 0.0]))
    assert l == pytest.approx(math.log(1 + math.exp(0.8)), rel=1e-6)
    assert g_[0, 0] == pytest.approx(-1 / (1 + math.exp(-0.8)), rel=1e-6) and g_[0, 1] == pytest.approx(-g_[0, 0])


def test_stochastic_neuralndcg_matches_reference_golden(extra_golden):
    from tests.cases import iter_stochastic_cases
    n = 0
    for name, c, s, y, gum, rl, rg, strict in iter_stochastic_cases(extra_golden):
        loss, grad = O.neuralndcg_stochastic(s, y, gum, temperature=c["tau"], k=c["k"], powered_relevancies=c["pw"],
                                             transposed=c["tr"], beta=c["beta"], log_scores=c["log"])
        assert close(loss, rl), (name, loss, rl)
        assert grad_close(np.where(strict, grad, 0), np.where(strict, rg, 0)), (name,)
        assert np.isfinite(grad).all()
        n += 1
    assert n == 16


def test_row4_kats_from_the_reference_tests():
    from tests.cases import row4_kats, mrr_kats
    for kind, kw, yp, yt, expected in row4_kats():
        got = _EXTRA[kind](np.asarray([yp], np.float32), np.asarray([yt], np.float32), **kw)[0]
        assert math.isfinite(got) and got == pytest.approx(expected, rel=2e-6, abs=1e-9), (kind, kw, yp, got, expected)
    for yp, yt, ats, expected in mrr_kats():
        assert np.array_equal(O.mrr(np.asarray(yp, np.float32), np.asarray(yt, np.float32), ats), np.asarray(expected, np.float32))
    # test_loss_ordinal.py:20-24
    t = (np.asarray([[2.0, 1.0, 0.0]])[:, :, None] >= np.arange(1, 3)[None, None, :]).astype(float).tolist()
    assert t == [[[1.0, 1.0], [1.0, 0.0], [0.0, 0.0]]]


def test_torch_port_matches_numpy_oracle():
    """oracle/torch_port.py (bench.py's cpu_baseline leg: the reference step on CPU torch operators) == the numpy oracle,
    which the tests above pin to the reference's own golden vectors: three training steps, losses and updated weights."""
    import torch
    from oracle import model_oracle as M, torch_port as T
    cfg = dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None)
    for loss, ofn in (("approxNDCGLoss", lambda s, t: O.approxndcg(s, t)), ("listNet", lambda s, t: O.listnet(s, t))):
        p = M.init_params(cfg, seed=3)
        rng = np.random.default_rng(0)
        B, L = 4, 30
        x = rng.standard_normal((B, L, 20)).astype(np.float32)
        y = rng.integers(0, 5, (B, L)).astype(np.float32)
        y[1, 20:] = -1
        x[1, 20:] = 0
        st = T.Stepper(p, cfg, loss)
        opt = M.Adam(p, lr=1e-3)
        for i in range(3):
            lt = st.step(torch.tensor(x), torch.tensor(y))
            lo = float(M.train_step(p, cfg, opt, x, y, ofn)[0])
            assert abs(lt - lo) <= (1e-6 if i == 0 else 1e-4) * (1 + abs(lo)), (loss, i, lt, lo)


def test_oracle_ndcg_with_gain_function_matches_reference_golden():
    """metrics.py:7-8,41-42,67 with a caller-supplied gain (tests/golden/make_golden_gain.py: identity -- what neuralNDCG.py:58 passes --,
    a gain that is not 0 at label 0, a non-monotone gain)"""
    import os
    from tests.golden.make_golden_gain import GAINS
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gain_golden.npz"))
    ats = [int(a) for a in g["ats"]]
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        s, y = g[pre + "s"], g[pre + "y"]
        for name, fn in GAINS.items():
            nd, _ = O.ndcg(s, y, ats=ats, gain_function=fn)
            dc, _ = O.dcg(s, y, ats=ats, gain_function=fn)
            nn, _ = O.ndcg(s, y, gain_function=fn, filler_value=0.25)
            assert np.allclose(nd, g[pre + name + ".ndcg"], rtol=1e-5, atol=1e-6), (ci, name)
            assert np.allclose(dc, g[pre + name + ".dcg"], rtol=1e-5, atol=1e-5), (ci, name)
            assert np.allclose(nn, g[pre + name + ".ndcg_none"], rtol=1e-5, atol=1e-6), (ci, name)


def test_oracle_dropout_backward_is_the_gradient_of_its_forward():
    """oracle/model_oracle.py with injected dropout masks (model.py:43, transformer.py:105,155,227) and the numpy restatement of the
    engine's counter-based masks (oracle/dropout_oracle.py): keep rates ~ 1 - p, multipliers 0 or 1/(1-p), and -- masks frozen -- the
    central difference of the loss along every parameter's gradient equals the gradient norm; drop=None is the old forward."""
    from oracle import dropout_oracle as D
    cfg = dict(n_features=10, fc_sizes=[12, 8], fc_activation="ReLU", fc_input_norm=False, N=2, d_ff=16, h=2, output_activation=None)
    p = {k: v.astype(np.float64) for k, v in M.init_params(cfg, seed=3).items()}
    rng = np.random.default_rng(0)
    B, L = 3, 7
    x = rng.standard_normal((B, L, 10))
    y = rng.integers(0, 5, (B, L)).astype(np.float64)
    y[1, 5:] = -1
    mask = y == -1
    big = D.keep_scale(0.3, 12345, 7, (200, 300))
    assert abs(float((big > 0).mean()) - 0.7) < 0.01 and set(np.unique(big).tolist()) == {0.0, float(np.float32(1) / (np.float32(1) - np.float32(0.3)))}
    att = D.attention_keep_scale(0.2, 99, 3, 4, 2, 64)
    assert att.shape == (4, 2, 64, 64) and abs(float((att > 0).mean()) - 0.8) < 0.02
    assert not np.array_equal(D.keep_scale(0.3, 12345, 8, (200, 300)) > 0, big > 0)          # the step word re-keys the mask
    drop = {"fc": [D.keep_scale(0.2, 11 + i, 1, (B, L, s_)) for i, s_ in enumerate([12, 8])],
            "layers": [{"att": D.attention_keep_scale(0.25, 100 + n, 1, B, 2, L), "ff": D.keep_scale(0.2, 200 + n, 1, (B, L, 16)),
                        "s0": D.keep_scale(0.3, 300 + n, 1, (B, L, 8)), "s1": D.keep_scale(0.1, 400 + n, 1, (B, L, 8))} for n in range(2)]}

    def loss_of(pp):
        s_, c_ = M.forward(pp, cfg, x, mask, drop)
        return O.approxndcg(s_, y, dtype=np.float64)[0], s_, c_
    _, s0, c0 = loss_of(p)
    g = M.backward(p, cfg, c0, O.approxndcg(s0, y, dtype=np.float64)[1])
    for k in p:
        n_ = np.linalg.norm(g[k])
        if n_ < 1e-12:
            continue
        d_ = g[k] / n_
        pp = dict(p)
        pp[k] = p[k] + 1e-6 * d_
        lp = loss_of(pp)[0]
        pp[k] = p[k] - 1e-6 * d_
        lm = loss_of(pp)[0]
        assert abs((lp - lm) / 2e-6 - n_) <= 1e-4 * n_, k
    s_plain, _ = M.forward(p, cfg, x, mask)
    s_none, _ = M.forward(p, cfg, x, mask, None)
    assert np.array_equal(s_plain, s_none) and not np.allclose(s_plain, s0)
