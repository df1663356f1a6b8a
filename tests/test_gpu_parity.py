"""GPU parity tests proper (-m gpu): the HIP kernels, called through the C ABI / Python boundary, against
(1) the golden vectors generated from the reference itself and (2) the numpy oracle on seeded random inputs, plus
size-independent properties at full size.  Tolerance: |a-b| <= 1e-5 (1 + |b|) for loss / NDCG values (BASELINE.json
north_star: fp32 within 1e-5), bit-exact sort indices on the valid prefix, gradients within 2e-4 of the largest
reference gradient entry.  /root/reference is NOT needed here."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O
from oracle import model_oracle as M
from tests.cases import close, grad_close, iter_loss_cases

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _log(name, obj):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_%s.json" % name), "w") as fh:
        json.dump(obj, fh, indent=1, default=float)


def _t(a, rg=False):
    return torch.tensor(np.asarray(a), device=DEV, requires_grad=rg)


def _engine_loss(kind, kw, s, y):
    from allrank_amd import losses as E
    sp = _t(s, True)
    yt = _t(y)
    if kind == "listnet":
        l = E.listNet(sp, yt)
    elif kind == "approxndcg":
        l = E.approxNDCGLoss(sp, yt, **kw)
    elif kind == "listmle":
        l = E.listMLE(sp, yt, perm=torch.tensor(kw["perm"]))
    elif kind == "lambdaloss":
        l = E.lambdaLoss(sp, yt, **kw)
    elif kind == "neuralndcg":
        kw = dict(kw)
        tr = kw.pop("transposed")
        l = (E.neuralNDCG_transposed if tr else E.neuralNDCG)(sp, yt, **kw)
    else:
        raise KeyError(kind)
    l.backward()
    return float(l.item()), sp.grad.cpu().numpy()


def _oracle_loss(kind, kw, s, y):
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


def test_mfma_layout_selftest():
    from allrank_amd import ops
    rng = np.random.default_rng(0)
    A = rng.standard_normal((32, 2)).astype(np.float32)          # asymmetric operands: catches row/col swaps
    Bm = rng.standard_normal((2, 32)).astype(np.float32)
    D = ops.mfma_selftest(_t(A), _t(Bm)).cpu().numpy()
    ref = (A.astype(np.float64) @ Bm.astype(np.float64))
    assert np.abs(D - ref).max() < 1e-5, np.abs(D - ref).max()


def test_losses_match_reference_golden(losses_golden):
    bad, rows = [], []
    n = 0
    for name, kind, kw, s, y, rl, rg in iter_loss_cases(losses_golden):
        lo, go = _engine_loss(kind, kw, s, y)
        n += 1
        ok = close(lo, rl) and grad_close(go, rg) and np.all(go[y == -1] == 0)
        rows.append(dict(name=name, loss=lo, ref=float(rl), gerr=float(np.abs(go - rg).max()), gmax=float(np.abs(rg).max()), ok=bool(ok)))
        if not ok:
            bad.append(rows[-1])
    _log("golden_losses", rows)
    assert n == 208 and not bad, bad[:8]


@pytest.mark.parametrize("B,L,seed", [(64, 240, 1), (7, 33, 2), (3, 1, 3), (5, 257, 4), (2, 1024, 5)])
def test_losses_match_oracle_random(B, L, seed):
    from tests.golden.make_inputs import make_inputs
    s, y = make_inputs(B, L, seed)
    perm = np.random.default_rng(seed).permutation(L).astype(np.int64)
    cases = [("listnet", {}), ("approxndcg", dict(alpha=1.0)), ("listmle", dict(perm=perm)),
             ("lambdaloss", dict(weighing_scheme="lambdaRank_scheme")),
             ("lambdaloss", dict(weighing_scheme="ndcgLoss2PP_scheme", k=10, reduction="mean", reduction_log="natural")),
             ("lambdaloss", dict(weighing_scheme="ndcgLoss1_scheme", k=5))]
    # (every size: L = 1024 runs the general L2-streaming Sinkhorn kernels, L <= 240 the register-resident ones)
    cases += [("neuralndcg", dict(transposed=False, temperature=1.0)),
              ("neuralndcg", dict(transposed=True, temperature=0.5, k=5, powered_relevancies=False))]
    bad, rows = [], []
    for kind, kw in cases:
        lo, go = _engine_loss(kind, kw, s, y)
        ro, rg = _oracle_loss(kind, kw, s, y)
        ok = close(lo, ro, rtol=2e-5) and grad_close(go, rg, rtol=5e-4) and np.all(go[y == -1] == 0)
        rows.append(dict(kind=kind, kw={k: (v if not isinstance(v, np.ndarray) else "perm") for k, v in kw.items()},
                         loss=lo, ref=float(ro), gerr=float(np.abs(go - rg).max()), gmax=float(np.abs(rg).max()), ok=bool(ok)))
        if not ok:
            bad.append(rows[-1])
    _log("random_losses_%dx%d" % (B, L), rows)
    assert not bad, bad


def test_padding_invariance_and_no_input_mutation():
    """the reference's own test idiom (tests/losses/*: *_ignores_padded): an extra padded slot changes nothing."""
    from allrank_amd import losses as E
    yp = torch.tensor([[0.5, 0.3, 0.5]], device=DEV)
    yt = torch.tensor([[0.5, 0.3, 0.5]], device=DEV)
    ypp = torch.tensor([[0.5, 0.3, 0.5, 1.0]], device=DEV)
    ytp = torch.tensor([[0.5, 0.3, 0.5, -1.0]], device=DEV)
    kat = [(lambda a, b: E.approxNDCGLoss(a, b, alpha=1.), -0.8499219417),
           (lambda a, b: E.lambdaLoss(a, b, weighing_scheme="ndcgLoss1_scheme", reduction_log="binary"), 2.9272110462),
           (lambda a, b: E.lambdaLoss(a, b, weighing_scheme="ndcgLoss2PP_scheme", reduction_log="binary"), 1.1244146823),
           (lambda a, b: E.lambdaLoss(a, b, weighing_scheme="rankNet_scheme", reduction_log="natural"), 1.1962778568)]
    for fn, expected in kat:
        c0, c1 = ypp.clone(), ytp.clone()
        r, rp = fn(yp, yt).item(), fn(ypp, ytp).item()
        assert r == pytest.approx(expected, abs=1e-5) and rp == pytest.approx(r, abs=1e-6)
        assert torch.equal(ypp, c0) and torch.equal(ytp, c1)
    r = E.listMLE(torch.tensor([[0.5, 0.3, 0.5]], device=DEV), torch.tensor([[1.0, 0.0, -1.0]], device=DEV)).item()
    assert r == pytest.approx(0.5981389284133911, abs=1e-6)            # tests/losses/test_listmle.py:14-22
    r = E.listNet(torch.tensor([[0.5, -1e30]], device=DEV), torch.tensor([[1.0, 0.0]], device=DEV)).item()
    assert np.isfinite(r)                                              # tests/losses/test_listnet.py:29-35


def test_neuralndcg_equals_ndcg_at_low_temperature():
    """tests/losses/test_neuralndcg.py:16-94 (deterministic variants)."""
    from allrank_amd import losses as E, metrics as EM
    PAD = -1.0
    cases = [([0.5, 0.2], [1.0, 0.0], 1e-4, None),
             ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0], 1e-4, None),
             ([0.5, -1e30], [1.0, 0.0], 1e-4, None),
             ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63, 1., 0.5, 0.3], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0, PAD, PAD, PAD], 1e-3, None),
             ([0.5, 0.2, 0.1, 0.4, 1.0, -1.0, 0.63], [1.0, 2.0, 2.0, 4.0, 1.0, 4.0, 3.0], 1e-4, 3)]
    for yp, yt, tau, k in cases:
        a, b = torch.tensor([yp], device=DEV), torch.tensor([yt], device=DEV)
        e = EM.ndcg(a, b, ats=None if k is None else [k]).mean().item()
        for fn in (E.neuralNDCG, E.neuralNDCG_transposed):
            r = fn(a, b, temperature=tau, k=k).item()
            assert np.isfinite(r) and -r == pytest.approx(e, abs=1e-5), (yp, tau, k, r, e)


def test_ndcg_and_sort_indices(losses_golden):
    from allrank_amd import metrics as EM
    g = losses_golden
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        s, y = g[pre + "s"], g[pre + "y"]
        ats = [int(a) for a in g[pre + "ndcg.ats"]]
        nd, order = EM.ndcg(_t(s), _t(y), ats=ats, return_order=True)
        dc = EM.dcg(_t(s), _t(y), ats=ats)
        assert close(nd.cpu().numpy(), g[pre + "ndcg.val"]) and close(dc.cpu().numpy(), g[pre + "dcg.val"])
        order = order.cpu().numpy()
        nv = (y != -1).sum(1)
        for b in range(s.shape[0]):
            assert np.array_equal(order[b, :nv[b]], g[pre + "order"][b, :nv[b]])      # bit-exact (tie policy)
            assert sorted(order[b].tolist()) == list(range(s.shape[1]))              # a permutation


def test_ndcg_with_a_gain_function_matches_reference_golden():
    """metrics.py:7-8,41-42 ``gain_function``: the identity the reference itself passes (losses/neuralNDCG.py:58), a gain that is
    not zero at label 0 (padded items then count, metrics.py:35,67) and a non-monotone gain (the ideal ranking is by label,
    metrics.py:21) -- reference-generated vectors (tests/golden/make_golden_gain.py), and the oracle on a long slate"""
    from allrank_amd import metrics as EM
    from tests.golden.make_golden_gain import GAINS
    from tests.golden.make_inputs import make_inputs
    g = np.load(os.path.join(ROOT, "tests", "golden", "gain_golden.npz"))
    ats = [int(a) for a in g["ats"]]
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        s, y = g[pre + "s"], g[pre + "y"]
        for name, fn in GAINS.items():
            nd = EM.ndcg(_t(s), _t(y), ats=ats, gain_function=fn).cpu().numpy()
            dc = EM.dcg(_t(s), _t(y), ats=ats, gain_function=fn).cpu().numpy()
            nn = EM.ndcg(_t(s), _t(y), gain_function=fn, filler_value=0.25).cpu().numpy()
            assert close(nd, g[pre + name + ".ndcg"]) and close(dc, g[pre + name + ".dcg"]) and close(nn, g[pre + name + ".ndcg_none"]), (ci, name)
    # the reference's default gain handed over explicitly == the in-kernel default
    s, y = make_inputs(4, 300, 77, tie_scores=True)
    a = EM.ndcg(_t(s), _t(y), ats=[5, 30], gain_function=lambda x: torch.pow(2, x) - 1).cpu().numpy()
    b = EM.ndcg(_t(s), _t(y), ats=[5, 30]).cpu().numpy()
    assert close(a, b)
    s, y = make_inputs(3, 3000, 78)
    nd = EM.ndcg(_t(s), _t(y), ats=[5, 3000], gain_function=GAINS["plus1"]).cpu().numpy()
    assert close(nd, O.ndcg(s, y, ats=[5, 3000], gain_function=GAINS["plus1"])[0])
    with pytest.raises(ValueError):
        EM.ndcg(_t(s), _t(y), gain_function=lambda x: x.sum(1))


def test_metrics_on_validation_slates_longer_than_the_loss_limit():
    """validation sets are padded to their longest slate with no bound (dataset_loading.py:185-194): ndcg / mrr take slates of
    up to LTRX_MAX_METRIC_SLATE_LEN = 8192 items (values and stable sort indices == the oracle), beyond that -- and for a loss
    beyond its limit (LTRX_MAX_SLATE_LEN = 2048; LTRX_MAX_LONG_SLATE_LEN = 16384 for the four hot listwise losses) -- the call raises and
    the message names the limit (no silent fallback)."""
    from allrank_amd import metrics as EM, losses as E, _lib as LB
    from tests.golden.make_inputs import make_inputs
    for (B, L, seed) in [(3, 3000, 5), (2, 8192, 6)]:
        s, y = make_inputs(B, L, seed, tie_scores=True)
        nd, order = EM.ndcg(_t(s), _t(y), ats=[5, 30, L], return_order=True)
        ndo, oo = O.ndcg(s, y, ats=[5, 30, L])
        assert close(nd.cpu().numpy(), ndo, rtol=2e-5), (L, nd.cpu().numpy(), ndo)
        order = order.cpu().numpy()
        nv = (y != -1).sum(1)
        for b in range(B):
            assert np.array_equal(order[b, :nv[b]], oo[b, :nv[b]])
        assert close(EM.mrr(_t(s), _t(y), ats=[1, 10]).cpu().numpy(), O.mrr(s, y, ats=[1, 10]))
    s, y = make_inputs(1, LB.MAX_METRIC_SLATE_LEN + 1, 7)
    with pytest.raises(RuntimeError, match="LTRX_MAX_METRIC_SLATE_LEN"):
        EM.ndcg(_t(s), _t(y), ats=[5])
    s, y = make_inputs(1, LB.MAX_SLATE_LEN + 1, 8)
    with pytest.raises(RuntimeError, match="LTRX_MAX_SLATE_LEN"):
        E.neuralNDCG(_t(s, True), _t(y))
    s, y = make_inputs(1, LB.MAX_LONG_SLATE_LEN + 1, 8)
    with pytest.raises(RuntimeError, match="LTRX_MAX_LONG_SLATE_LEN"):
        E.listNet(_t(s, True), _t(y))


def test_ndcg_full_size_properties():
    """size-independent properties at the bench size: indices are a permutation that sorts the scores (stable), the
    oracle agrees, NDCG of the ideal ranking is 1, all-zero-label slates give the filler."""
    from allrank_amd import metrics as EM
    from tests.golden.make_inputs import make_inputs
    s, y = make_inputs(512, 240, 77, tie_scores=True)
    st, yt = _t(s), _t(y)
    nd, order = EM.ndcg(st, yt, ats=[5, 10, 240], return_order=True)
    ndo, oo = O.ndcg(s, y, ats=[5, 10, 240])
    assert close(nd.cpu().numpy(), ndo)
    order = order.cpu().numpy()
    nv = (y != -1).sum(1)
    for b in range(0, 512, 17):
        assert np.array_equal(order[b, :nv[b]], oo[b, :nv[b]])
        v = s[b, order[b, :nv[b]]]
        assert np.all(v[:-1] >= v[1:])
    ideal = EM.ndcg(yt.clone(), yt, ats=[5, 240]).cpu().numpy()
    assert np.allclose(ideal, 1.0, atol=1e-6)
    assert np.all(nd.cpu().numpy()[1] == 1.0)        # slate 1 of make_inputs has no relevant item -> filler 1.0


def test_layernorm_forward_backward():
    from allrank_amd import ops
    rng = np.random.default_rng(3)
    for rows, D, with_res in [(37, 512, True), (64 * 240, 512, True), (9, 96, False), (5, 20, True)]:
        x = rng.standard_normal((rows, D)).astype(np.float32) * 2 + 0.3
        r = rng.standard_normal((rows, D)).astype(np.float32) if with_res else None
        a = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
        b = (0.1 * rng.standard_normal(D)).astype(np.float32)
        gy = rng.standard_normal((rows, D)).astype(np.float32)
        gx2 = rng.standard_normal((rows, D)).astype(np.float32)
        xt, at, bt = _t(x, True), _t(a, True), _t(b, True)
        rt = _t(r, True) if with_res else None
        y, xs = ops.layer_norm_residual(xt, rt, at, bt, 1e-6)
        xsum = x + r if with_res else x
        yo, cache = M.custom_ln_fwd(xsum.astype(np.float64), a.astype(np.float64), b.astype(np.float64))
        assert np.abs(y.detach().cpu().numpy() - yo).max() < 2e-5
        if with_res:
            (y * _t(gy)).sum().add((xs * _t(gx2)).sum()).backward()
        else:
            (y * _t(gy)).sum().backward()
        grads = {}
        gxo = M.custom_ln_bwd(cache, a.astype(np.float64), gy.astype(np.float64), grads, "n") + (gx2 if with_res else 0)
        sc = np.abs(gxo).max()
        assert np.abs(xt.grad.cpu().numpy() - gxo).max() < 1e-4 * sc
        if with_res:
            assert np.abs(rt.grad.cpu().numpy() - gxo).max() < 1e-4 * sc
        assert np.abs(at.grad.cpu().numpy() - grads["n.a_2"]).max() < 1e-4 * max(1.0, np.abs(grads["n.a_2"]).max())
        assert np.abs(bt.grad.cpu().numpy() - grads["n.b_2"]).max() < 1e-4 * max(1.0, np.abs(grads["n.b_2"]).max())


@pytest.mark.parametrize("mode", [0, 1])      # 0 = exact fp32 MFMA everywhere, 1 = split-bf16 LDS-resident kernels where the shape fits (default)
@pytest.mark.parametrize("B,L,h,dk", [(2, 240, 8, 64), (3, 70, 4, 8), (2, 33, 1, 96), (1, 300, 2, 32), (2, 129, 1, 128),
                                      (2, 64, 2, 72), (3, 256, 2, 64), (4, 37, 3, 48), (2, 5, 1, 64), (1, 300, 2, 64), (2, 1024, 1, 64), (2, 513, 2, 48)])
def test_attention_forward_backward(B, L, h, dk, mode):
    from allrank_amd import ops
    rng = np.random.default_rng(L * 7 + dk)
    d = h * dk
    qkv = rng.standard_normal((B, L, 3 * d)).astype(np.float32)
    mask = np.zeros((B, L), dtype=bool)
    for b in range(B):
        mask[b, L - 1 - 5 * b:] = b > 0
    if B > 1:
        mask[1, min(3, L - 1)] = L > 4         # a padded key in the middle (the reference allows arbitrary masks)
    go = rng.standard_normal((B, L, d)).astype(np.float32)
    t = _t(qkv, True)
    q, k, v = t[:, :, :d], t[:, :, d:2 * d], t[:, :, 2 * d:]
    with ops.arithmetic(attention=mode):          # the mode travels with the calls (forward and its backward), no library state
        o = ops.attention(q, k, v, _t(mask), h)
    (o * _t(go)).sum().backward()

    def heads(x):
        return x.reshape(B, L, h, dk).transpose(0, 2, 1, 3).astype(np.float64)

    qo, ko, vo = heads(qkv[:, :, :d]), heads(qkv[:, :, d:2 * d]), heads(qkv[:, :, 2 * d:])
    oo, p = M.attention_fwd(qo, ko, vo, mask)
    gq, gk, gv = M.attention_bwd(qo, ko, vo, p, heads(go))

    def unheads(x):
        return x.transpose(0, 2, 1, 3).reshape(B, L, d)

    err = dict(o=float(np.abs(o.detach().cpu().numpy() - unheads(oo)).max()))
    g = t.grad.cpu().numpy()
    for name, ref, sl in (("dq", gq, slice(0, d)), ("dk", gk, slice(d, 2 * d)), ("dv", gv, slice(2 * d, 3 * d))):
        err[name] = float(np.abs(g[:, :, sl] - unheads(ref)).max() / max(np.abs(ref).max(), 1e-6))
    _log("attention_%d_%d_%d_%d_mode%d" % (B, L, h, dk, mode), err)
    assert err["o"] < 2e-5 and err["dq"] < 1e-4 and err["dk"] < 1e-4 and err["dv"] < 1e-4, err
    assert np.all(g[:, :, d:][np.broadcast_to(mask[:, :, None], (B, L, 2 * d))] == 0)   # padded keys get exactly 0


@pytest.mark.parametrize("step", [0.6, 5.0, 11.0, 40.0])
def test_attention_lazy_softmax_reference_is_invariant_to_the_key_order(step):
    """Round 6: the LDS-resident forward keeps a LAZY softmax reference (a query's reference maximum only moves when a 32-key tile exceeds
    it by 2^8; until then P is taken relative to the stale reference and the O accumulators are not rescaled).  Random inputs almost never
    move the reference after the first tile, so this test builds logits that CLIMB with the key index -- by `step` natural-log units per
    32-key tile: 0.6 never moves the reference after tile 0 (P grows to ~2^6 against it), 5 moves it every second tile, 11 and 40 every
    tile -- and compares with the SAME keys and values in REVERSED order: there the first tile holds the largest scores, the reference is
    set once and never moves.  Attention is invariant to a joint permutation of keys and values, every (query, key) score is the same dot
    product with the same three-product rounding in both runs, so the two outputs may differ by the summation order of the softmax only --
    any error of the lazy-reference bookkeeping (a missed or doubled rescale, a wrong LSE) shows at O(1).  Gradients likewise (dk, dv
    flipped back).  Logits this large are outside the accuracy bars of the fp32-class arithmetic itself (|s| 2^-17 per score), which is
    why the comparison is between two orders of one arithmetic and not with the fp64 oracle; the oracle only bounds the output loosely."""
    from allrank_amd import ops
    B, L, h, dk = 2, 240, 2, 64
    d = h * dk
    rng = np.random.default_rng(int(step * 10) + 3)
    qkv = (0.05 * rng.standard_normal((B, L, 3 * d))).astype(np.float32)
    ramp = (np.arange(L) // 32).astype(np.float32) * step
    for hh in range(h):
        qkv[:, :, hh * dk] = 2.0                                                     # q . k / sqrt(dk) = ramp(key) + noise
        qkv[:, :, d + hh * dk] = ramp * np.sqrt(dk) / 2.0
    qkv[1, 100:140, 0] = -2.0                                                        # these queries see the ramp falling
    qkv[:, :, 2 * d:] = rng.standard_normal((B, L, d)).astype(np.float32)
    mask = np.zeros((B, L), dtype=bool)
    mask[1, 230:] = True
    go = rng.standard_normal((B, L, d)).astype(np.float32)
    res = {}
    for name, flip in (("climbing", False), ("reversed", True)):
        x, mk = qkv.copy(), mask.copy()
        if flip:                                   # keys and values (and their padding mask) in reversed order, queries as they were
            x[:, :, d:] = qkv[:, ::-1, d:]
            mk = mask[:, ::-1].copy()
        t = _t(x, True)
        with ops.arithmetic(attention=1):
            o = ops.attention(t[:, :, :d], t[:, :, d:2 * d], t[:, :, 2 * d:], _t(mk), h)
        (o * _t(go)).sum().backward()
        g = t.grad.cpu().numpy().astype(np.float64)
        if flip:
            g[:, :, d:] = g[:, ::-1, d:].copy()    # dk, dv back in the original key order
        res[name] = (o.detach().cpu().numpy().astype(np.float64), g)
        assert np.isfinite(res[name][0]).all() and np.isfinite(res[name][1]).all(), name
    (o1, g1), (o2, g2) = res["climbing"], res["reversed"]
    vmax = float(np.abs(qkv[:, :, 2 * d:]).max())
    err = dict(o=float(np.abs(o1 - o2).max()) / vmax)
    kmax = float(np.abs(qkv[:, :, d:2 * d]).max())
    for nm, sl in (("dq", slice(0, d)), ("dk", slice(d, 2 * d)), ("dv", slice(2 * d, 3 * d))):
        # (scale: the largest gradient entry of the three tensors; dq = sum_k dS k is a sum of terms |k| times larger than that -- k is
        #  dominated by the ramp feature, up to 1120 -- whose fp32 accumulation ORDER differs between the two runs: its bar carries |k|max)
        err[nm] = float(np.abs(g1[:, :, sl] - g2[:, :, sl]).max()) / float(np.abs(g1).max()) / (max(kmax, 1.0) if nm == "dq" else 1.0)
    # loose sanity bound against the fp64 oracle (the arithmetic's own error at these logits, not the reference logic)
    heads = lambda x: x.reshape(B, L, h, dk).transpose(0, 2, 1, 3).astype(np.float64)      # noqa: E731
    oo, _p = M.attention_fwd(heads(qkv[:, :, :d]), heads(qkv[:, :, d:2 * d]), heads(qkv[:, :, 2 * d:]), mask)
    err["o_vs_fp64"] = float(np.abs(o1 - oo.transpose(0, 2, 1, 3).reshape(B, L, d)).max()) / vmax
    _log("attention_key_order_%g" % step, err)
    # measured on the MI355X: o <= 2.8e-6, dk <= 2.6e-6, dv <= 1.7e-6, dq <= 1.04e-6 (per unit of |k|max), o vs fp64 <= 1.4e-5
    assert err["o"] < 1e-5 and err["dq"] < 3e-6 and err["dk"] < 1e-5 and err["dv"] < 1e-5, err
    assert err["o_vs_fp64"] < 1e-4, err


@pytest.mark.parametrize("mode", [0, 1])
def test_attention_fully_masked_slate_is_zero_and_finite(mode):
    """A slate whose keys are all padding (dataset.py pads whole slates when a batch is short): output and all three input
    gradients of that slate are exactly 0 and nothing is NaN / inf; the other slate of the batch is unaffected (== the same
    slate run alone).  Both kernel families (the LDS-resident split-bf16 kernels hand dS through their HBM workspace)."""
    from allrank_amd import ops
    rng = np.random.default_rng(77)
    B, L, h, dk = 2, 40, 2, 64
    d = h * dk
    qkv = rng.standard_normal((B, L, 3 * d)).astype(np.float32)
    go = rng.standard_normal((B, L, d)).astype(np.float32)
    mask = np.zeros((B, L), dtype=bool)
    mask[1, :] = True
    mask[0, 33:] = True

    def run(qkv_, go_, mask_):
        t = _t(qkv_, True)
        with ops.arithmetic(attention=mode):
            o = ops.attention(t[:, :, :d], t[:, :, d:2 * d], t[:, :, 2 * d:], _t(mask_), h)
        (o * _t(go_)).sum().backward()
        return o.detach().cpu().numpy(), t.grad.cpu().numpy()

    o, g = run(qkv, go, mask)
    assert np.isfinite(o).all() and np.isfinite(g).all()
    assert (o[1] == 0).all() and (g[1] == 0).all()
    o1, g1 = run(qkv[:1], go[:1], mask[:1])
    assert np.array_equal(o[0], o1[0]) and np.array_equal(g[0], g1[0])


def test_abi_is_reentrant_two_threads_in_different_attention_modes():
    """VERDICT r2 item 2 / SURVEY 8b "Threading / streams": the library keeps no mode.  Two Python threads (the reference's
    DataParallel replicas are threads, main.py:76-78, model_utils.py:40-53) run attention forward + backward concurrently on
    their own streams, one in mode 0 (exact fp32 MFMA) and one in mode 1 (split-bf16), 20 rounds each; every result must be
    BIT-identical to the single-threaded result of its own mode (the kernels are deterministic), and the two modes must
    really differ from each other (so a leak of one thread's mode into the other would show)."""
    import threading
    from allrank_amd import ops
    B, L, h, dk = 4, 240, 8, 64
    d = h * dk
    rng = np.random.default_rng(99)
    qkv = _t(rng.standard_normal((B, L, 3 * d)).astype(np.float32))
    go = _t(rng.standard_normal((B, L, d)).astype(np.float32))
    mask = torch.zeros((B, L), dtype=torch.bool, device=DEV)
    mask[1, 200:] = True

    def run(mode):
        t = qkv.clone().requires_grad_(True)
        with ops.arithmetic(attention=mode):
            o = ops.attention_packed(t, mask, h)
        o.backward(go)                          # the backward carries the mode of its forward
        return o.detach().clone(), t.grad.clone()

    ref = {m: run(m) for m in (0, 1)}
    assert not torch.equal(ref[0][0], ref[1][0])                 # two different arithmetics ...
    assert float((ref[0][0] - ref[1][0]).abs().max()) < 1e-4    # ... of the same function
    bad, errs = [], []

    def worker(mode):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(20):
                    o, g = run(mode)
                    st.synchronize()
                    if not (torch.equal(o, ref[mode][0]) and torch.equal(g, ref[mode][1])):
                        bad.append((mode, it))
        except Exception as e:        # noqa: BLE001 -- surfaced below
            errs.append(repr(e))

    th = [threading.Thread(target=worker, args=(m,)) for m in (0, 1, 1, 0)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    assert not errs, errs
    assert not bad, bad


def _cfg_from_golden(g, pre):
    acts = {"ReLU": "ReLU", "Tanh": "Tanh", "Sigmoid": "Sigmoid", "-1": None}
    v = lambda k: g[pre + "cfg." + k]  # noqa: E731
    return dict(n_features=int(v("n_features")), fc_sizes=[int(x) for x in np.atleast_1d(v("fc_sizes"))],
                fc_activation=acts[str(v("fc_activation"))], fc_input_norm=bool(v("fc_input_norm")), N=int(v("N")),
                d_ff=int(v("d_ff")), h=int(v("h")), output_activation=acts[str(v("output_activation"))])


def _make_engine_model(cfg, params=None):
    from allrank_amd.model import make_model
    tr = dict(N=cfg["N"], d_ff=cfg["d_ff"], h=cfg["h"], positional_encoding=None, dropout=0.0) if cfg["N"] else None
    fc = dict(sizes=list(cfg["fc_sizes"]), input_norm=cfg["fc_input_norm"], activation=cfg["fc_activation"], dropout=0.0)
    model = make_model(fc, tr, dict(d_output=1, output_activation=cfg["output_activation"]), cfg["n_features"])
    if params is not None:
        missing = model.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(DEV)


def test_model_matches_reference_golden(model_golden):
    """scores and every parameter gradient (through approxNDCG) vs the reference's own CPU run; the golden state_dict
    loads with strict=True, i.e. keys and shapes are the reference's (checkpoint contract)."""
    from allrank_amd import losses as E
    g = model_golden
    rows = []
    for mi in range(int(g["n_models"])):
        pre = "m%d." % mi
        cfg = _cfg_from_golden(g, pre)
        params = {k[len(pre + "param."):]: v for k, v in g.items() if k.startswith(pre + "param.")}
        model = _make_engine_model(cfg, params)
        x, y = _t(g[pre + "x"]), _t(g[pre + "y"])
        mask = y == -1
        sc = model(x, mask, None)
        loss = E.approxNDCGLoss(sc, y)
        loss.backward()
        valid = ~mask.cpu().numpy()
        serr = float(np.abs(sc.detach().cpu().numpy() - g[pre + "scores"])[valid].max())
        sscale = max(1.0, float(np.abs(g[pre + "scores"])[valid].max()))      # (scores relative to their scale: the nn.Linear layers run
        #                                                                        the three-product split-bf16 GEMMs, 1.4e-6 of sum |a||b|)
        allg = np.concatenate([g[pre + "grad." + k].ravel() for k in params])
        scale = float(np.abs(allg).max())
        gerr = max(float(np.abs(p.grad.cpu().numpy() - g[pre + "grad." + n]).max()) for n, p in model.named_parameters())
        rows.append(dict(model=mi, score_err=serr, score_scale=sscale, loss=float(loss.item()), ref_loss=float(g[pre + "loss"]), grad_err=gerr,
                         grad_scale=scale))
        assert serr < 2e-5 * sscale and close(loss.item(), g[pre + "loss"]) and gerr <= 2e-4 * scale + 1e-8, rows[-1]
        assert torch.equal(model.score(x, mask, None), model(x, mask, None))
    _log("model_golden", rows)


def _engine_relu_patterns(model, run):
    """the feed-forward ReLU pattern (r > 0, one bool [B, L, d_ff] array per encoder layer) of the ENGINE's forward: the input
    of every PositionwiseFeedForward is captured by a hook while ``run()`` executes the forward, and w_1's GEMM + ReLU is
    evaluated again on it with the same kernel (deterministic, so these are the bits the fused node used)."""
    from allrank_amd import ops
    caught, hooks = [], []
    for lay in model.encoder.layers:
        hooks.append(lay.feed_forward.register_forward_pre_hook(lambda mod, inp: caught.append((mod, inp[0].detach()))))
    try:
        out = run()
    finally:
        for h_ in hooks:
            h_.remove()
    pats = []
    with torch.no_grad():
        for mod, xin in caught:
            pats.append((ops.linear(xin, mod.w_1.weight, mod.w_1.bias, act=1) > 0).cpu().numpy())
    return out, pats


@pytest.mark.parametrize("backend", ["split_bf16", "hipblaslt"])
def test_model_config3_matches_oracle(backend):
    """BASELINE.json config (3): F=136, fc [512], N=2, h=8, d_ff=2048, slate 240 -- forward + backward of the nn.Module path vs
    the numpy oracle, with its nn.Linear layers on the split-bf16 GEMMs (the default, ops.linear) and on hipBLASLt fp32.
    ONE gradient bar for both arithmetics: every one of the 6.4 M gradient entries within 5e-4 of the largest
    gradient.  ReLU has no derivative at 0: an fp32-class forward error (1.4e-6 here, 2.8e-7 with hipBLASLt) puts the few
    feed-forward units whose pre-activation is within round-off of 0 on the other branch, and each such unit moves one row's
    contribution to dW_1 / db_1 by a finite amount in ANY arithmetic.  The oracle therefore differentiates through the
    engine's own ReLU pattern (oracle/model_oracle.py backward(relu_masks=)); the units on which the two patterns differ are
    counted, must be a vanishing fraction, and must all have an oracle pre-activation within round-off of 0."""
    from allrank_amd import losses as E
    from allrank_amd import ops
    cfg = dict(n_features=136, fc_sizes=[512], fc_activation=None, fc_input_norm=False, N=2, d_ff=2048, h=8, output_activation=None)
    params = M.init_params(cfg, seed=5)
    model = _make_engine_model(cfg, params)
    rng = np.random.default_rng(6)
    B, L = 4, 240
    x = rng.standard_normal((B, L, 136)).astype(np.float32)
    y = rng.choice(5, size=(B, L), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
    y[1, 200:] = -1
    x[1, 200:] = 0
    y[3, 17:] = -1
    x[3, 17:] = 0
    mask = y == -1
    with ops.arithmetic(linear=backend):
        sc, pats = _engine_relu_patterns(model, lambda: model(_t(x), _t(mask), None))
        loss = E.approxNDCGLoss(sc, _t(y))
        loss.backward()
    so, cache = M.forward(params, cfg, x, mask)
    lo, gs, _ = O.approxndcg(so, y)
    flips, zmax = 0, 0.0
    for lc, pat in zip(cache["layers"], pats):
        diff = pat != (lc["z"] > 0)
        flips += int(diff.sum())
        if diff.any():
            zmax = max(zmax, float(np.abs(lc["z"][diff]).max()))
    grads = M.backward(params, cfg, cache, gs, relu_masks=pats)
    serr = float(np.abs(sc.detach().cpu().numpy() - so)[~mask].max())
    scale = max(float(np.abs(v).max()) for v in grads.values())
    gerr = max(float(np.abs(p.grad.cpu().numpy() - grads[n]).max()) for n, p in model.named_parameters())
    n_units = sum(int(p_.size) for p_ in pats)
    _log("model_cfg3_%s" % backend, dict(score_err=serr, loss=float(loss.item()), oracle_loss=float(lo), grad_err=gerr, grad_scale=scale,
                                         relu_units=n_units, relu_units_on_other_branch=flips, max_abs_preact_of_those=zmax))
    assert serr < 5e-5 and close(loss.item(), lo)
    assert flips <= 2e-4 * n_units and zmax < 2e-4, (flips, n_units, zmax)
    assert gerr <= 5e-4 * scale, (gerr, scale, backend)


@pytest.mark.parametrize("gemm", ["split_bf16", "hipblaslt", "split_bf16_strict"])
@pytest.mark.parametrize("loss_name,loss_args", [("approxNDCGLoss", {}), ("listNet", {}),
                                                  ("lambdaLoss", dict(weighing_scheme="lambdaRank_scheme", k=10)),
                                                  ("neuralNDCG", dict(temperature=1.0))])
def test_fused_trainer_matches_oracle_and_autograd_path(loss_name, loss_args, gemm):
    """the explicit (hipGraph-captured) training step == the autograd path == the numpy oracle, step after step."""
    import copy
    from allrank_amd import losses as E
    from allrank_amd.engine import FusedTrainer, Trainer
    cfg = dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None)
    params = M.init_params(cfg, seed=11)
    m1 = _make_engine_model(cfg, params)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(12)
    B, L = 4, 70
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[2, 40:] = -1
    x[2, 40:] = 0
    xt, yt = _t(x), _t(y)
    ft = FusedTrainer(m1, loss_name, loss_args, B, L, lr=1e-3, use_graph=True, gemm=gemm)
    lossfn = (lambda s, t: getattr(E, loss_name)(s, t, **loss_args))
    tr = Trainer(m2, lossfn, torch.optim.Adam(m2.parameters(), lr=1e-3))
    ofn = {"approxNDCGLoss": lambda s, t: O.approxndcg(s, t), "listNet": lambda s, t: O.listnet(s, t),
           "lambdaLoss": lambda s, t: O.lambdaloss(s, t, **loss_args), "neuralNDCG": lambda s, t: O.neuralndcg(s, t, **loss_args)}[loss_name]
    oopt = M.Adam(params, lr=1e-3)
    rows = []
    for step in range(5):                       # steps 0,1 eager warm-up, step 2 captures + replays, 3,4 replay
        lf = float(ft.step(xt, yt).item())
        la = float(tr.step(xt, yt).item())
        lo = float(M.train_step(params, cfg, oopt, x, y, ofn)[0])
        rows.append((lf, la, lo))
        tol = 1e-5 if step == 0 else 2e-3       # after the first Adam step round-off of ~0 gradients is amplified to +-lr
        assert abs(lf - la) <= tol * (1 + abs(la)) and abs(lf - lo) <= tol * (1 + abs(lo)), rows
    _log("fused_trainer_%s_%s" % (loss_name, gemm), rows)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:                               # weights live in the flat buffer but are visible through the module
        # 5 Adam steps move a parameter by at most 5*lr; parameters whose true gradient is 0 (output bias under a
        # shift-invariant loss, key bias) follow the SIGN of round-off noise, so the two paths may differ by up to 2*5*lr
        assert (sd1[k] - sd2[k]).abs().max().item() <= 1.01e-2, k
    with torch.no_grad():
        s1 = m1.score(xt, yt == -1, None)
    assert torch.isfinite(s1).all()


def test_fused_trainer_fc_only_relu():
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=20, fc_sizes=[24, 16], fc_activation="ReLU", fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
    params = M.init_params(cfg, seed=3)
    m1 = _make_engine_model(cfg, params)
    rng = np.random.default_rng(4)
    B, L = 8, 24
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    ft = FusedTrainer(m1, "listNet", {}, B, L, lr=1e-3, use_graph=False, gemm="split_bf16")
    oopt = M.Adam(params, lr=1e-3)
    for step in range(2):
        lf = float(ft.step(_t(x), _t(y)).item())
        lo = float(M.train_step(params, cfg, oopt, x, y, lambda s, t: O.listnet(s, t))[0])
        assert abs(lf - lo) <= (1e-5 if step == 0 else 1e-3) * (1 + abs(lo)), (step, lf, lo)


@pytest.mark.parametrize("act", ["Tanh", "Sigmoid"])
def test_fused_trainer_fc_sigmoid_tanh(act):
    """FCModel activations other than ReLU (model.py:28-29 resolves any torch.nn name) on the explicit step: a two-layer FC stack with
    Tanh / Sigmoid in front of a one-layer encoder -- loss, scores and every parameter gradient against the oracle"""
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=20, fc_sizes=[24, 16], fc_activation=act, fc_input_norm=False, N=1, d_ff=32, h=2, output_activation=None)
    params = M.init_params(cfg, seed=13)
    m1 = _make_engine_model(cfg, params)
    rng = np.random.default_rng(14)
    B, L = 8, 24
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[2, 15:] = -1
    x[2, 15:] = 0
    mask = y == -1
    ft = FusedTrainer(m1, "listNet", {}, B, L, lr=1e-3, use_graph=False, gemm="split_bf16")
    so, cache = M.forward(params, cfg, x, mask)
    lo, gs, _ = O.listnet(so, y)
    grads = M.backward(params, cfg, cache, gs)
    lf = float(ft.step(_t(x), _t(y)).item())
    assert abs(lf - float(lo)) <= 1e-5 * (1 + abs(float(lo))), (lf, lo)
    assert float(np.abs(ft.scores.cpu().numpy() - so)[~mask].max()) < 2e-5 * max(1.0, float(np.abs(so).max()))
    scale = max(float(np.abs(v).max()) for v in grads.values())
    for n, p_ in m1.named_parameters():
        assert float(np.abs(p_.grad.cpu().numpy() - grads[n]).max()) <= 2e-4 * scale, (act, n)


@pytest.mark.parametrize("strict", [0, 1])
def test_split_bf16_gemm_matches_fp64(strict):
    """fp32-accurate GEMMs on the bf16 MFMA: error relative to sum_k |a||b| must be fp32-class
    (2-term split: <= 3*2^-18 per product worst case; 3-term 'strict': <= 2^-24-ish)."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(0)
    rows = []
    for (Mm, N, K) in [(300, 200, 136), (1024, 512, 512), (129, 1, 512), (15360, 96, 136)]:
        A = rng.standard_normal((Mm, K)).astype(np.float32)
        Bw = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        At, Bt, bt = _t(A), _t(Bw), _t(bias)
        C = torch.empty((Mm, N), device=DEV)
        LB.check(lib.ltrx_gemm_nt(LB.ptr(At), K, LB.ptr(Bt), K, None, LB.ptr(C), N, Mm, N, K, LB.ptr(bt), 1, None, 0, 0.0, 0, None, strict, 0, None), "gemm_nt")
        aux = rng.standard_normal((Mm, N)).astype(np.float32)
        C2 = torch.empty((Mm, N), device=DEV)
        LB.check(lib.ltrx_gemm_nt(LB.ptr(At), K, LB.ptr(Bt), K, None, LB.ptr(C2), N, Mm, N, K, None, 2, LB.ptr(_t(aux)), N, 0.0, 0, None, strict, 0, None), "gemm_nt(mask)")
        ref2 = (A.astype(np.float64) @ Bw.astype(np.float64).T) * (aux > 0)
        assert float(np.abs(C2.cpu().numpy() - ref2).max()) < 1e-4
        ref = np.maximum(A.astype(np.float64) @ Bw.astype(np.float64).T + bias, 0)
        scale = (np.abs(A).astype(np.float64) @ np.abs(Bw).astype(np.float64).T).max()
        err = float(np.abs(C.cpu().numpy() - ref).max() / scale)
        tref = torch.relu(torch.addmm(bt, At, Bt.t())).cpu().numpy()
        err_blas = float(np.abs(tref - ref).max() / scale)
        rows.append(dict(kind="nt", shape=(Mm, N, K), strict=strict, rel_err=err, rel_err_hipblaslt_fp32=err_blas))
        assert err < (4e-7 if strict else 4e-6), rows[-1]
    for (Mm, NP, KP) in [(1000, 200, 136), (15360, 512, 136), (4096, 1536, 512), (77, 5, 3)]:
        A = rng.standard_normal((Mm, NP)).astype(np.float32)
        Bx = rng.standard_normal((Mm, KP)).astype(np.float32)
        At, Bt = _t(A), _t(Bx)
        C = torch.empty((NP, KP), device=DEV)
        gb = torch.empty(NP, device=DEV)
        ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(Mm, NP, KP), 64), dtype=torch.uint8, device=DEV)
        LB.check(lib.ltrx_gemm_tn(LB.ptr(At), NP, LB.ptr(Bt), KP, LB.ptr(C), LB.ptr(gb), Mm, NP, KP, strict, 0, LB.ptr(ws), None), "gemm_tn")
        bref = A.astype(np.float64).sum(0)
        assert float(np.abs(gb.cpu().numpy() - bref).max()) < 1e-5 * max(1.0, np.abs(A).sum(0).max())
        ref = A.astype(np.float64).T @ Bx.astype(np.float64)
        scale = (np.abs(A).astype(np.float64).T @ np.abs(Bx).astype(np.float64)).max()
        err = float(np.abs(C.cpu().numpy() - ref).max() / scale)
        rows.append(dict(kind="tn", shape=(Mm, NP, KP), strict=strict, rel_err=err))
        assert err < (4e-7 if strict else 4e-6), rows[-1]
    _log("split_gemm_strict%d" % strict, rows)


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def test_bench_two_ranks_on_one_gpu_gloo():
    """the N>1 path of bench.py (torchrun, slate sharding, flat-gradient all-reduce, max-over-ranks timing) end to end:
    two ranks share the single GPU of the test box and exchange gradients over gloo (RCCL needs one GPU per rank)."""
    import subprocess
    import sys
    env = dict(os.environ, LTRX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--slates-per-gpu", "8", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["value"] > 0 and np.isfinite(rec["last_loss"])


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no torchrun around it (the form the driver uses) must start two ranks itself and
    print a line with n_gpus == 2 (gloo here: the test box has one GPU; on a multi-GPU node the backend is RCCL)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["LTRX_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--slates-per-gpu", "8",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 16 and rec["value"] > 0 and np.isfinite(rec["last_loss"])
    assert rec["comm"]["allreduce_bytes_per_step"] > 0 and rec["comm"]["exposed_ms"] >= 0.0


def test_sharded_step_equals_single_rank_step():
    """slate-sharded data parallelism reproduces the single-process step: 2 ranks x 4 slates (gloo, one GPU) vs
    1 rank x 8 slates -- same loss (sum of rank shares) and same updated weights; and the captured sharded step (hipGraph
    segments, collectives between them) == the eager sharded step bit for bit (tests/dist_equiv_worker.py)."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tests", "dist_equiv_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "EQUIV_OK" in out.stdout, out.stdout[-2000:]


def test_sharded_fit_with_uneven_last_batch_equals_one_rank():
    """``fit()`` on 2 ranks (gloo, one GPU) over batches of 16 / 16 / 1 slates -- the last one leaves rank 1 without a
    slate -- reproduces the 1-rank epoch losses and weights (tests/dist_fit_worker.py)."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "tests", "dist_fit_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        ref = os.path.join(tmp, "ref.pt")
        out = subprocess.run([sys.executable, script, "--ref", ref], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "FIT_REF_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), script, "--cmp", ref]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dist_fit_worker.log"), "w") as fh:
        fh.write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "FIT_EQUIV_OK" in out.stdout, out.stdout[-2000:]


def test_neuralndcg_block_resident_path_equals_general_path():
    """the block-resident kernels (L <= 240: 2 x 2 / 4 x 4 / 6 x 6 tiles on 16 waves, 15 x 5 tiles on 12 waves) and the general
    L2-streaming kernels agree with each other and the oracle; L = 250 runs the general kernels on both paths; the 3-item slates
    converge below tol before max_iter (batch-global early exit: the rewind of the backward kernel runs)."""
    from allrank_amd import losses as E, _lib as LB
    from tests.golden.make_inputs import make_inputs
    lib = LB.lib()
    for (B, L, seed) in [(16, 240, 21), (9, 100, 22), (5, 64, 23), (4, 130, 24), (3, 190, 25), (6, 33, 26), (2, 250, 27), (4, 3, 28), (7, 1, 29),
                         (3, 200, 30)]:
        s, y = make_inputs(B, L, seed)
        out = {}
        for force in (0, 1):
            sp = _t(s, True)
            with E.neural_kernel_path(force):
                l = E.neuralNDCG(sp, _t(y), temperature=0.7, k=20)
            l.backward()
            out[force] = (float(l.item()), sp.grad.cpu().numpy())
        ro, rg = O.neuralndcg(s, y, temperature=0.7, k=20)[:2]
        for force in (0, 1):
            assert close(out[force][0], ro, rtol=2e-5) and grad_close(out[force][1], rg, rtol=5e-4), (B, L, force, out[force][0], ro)
        assert abs(out[0][0] - out[1][0]) < 1e-6


def test_attention_dropout_statistics_and_gradient_consistency():
    """in-kernel dropout on the attention probabilities: (a) p = 0 is the plain kernel; (b) the keep rate is 1 - p and
    E[out] is the undropped output; (c) forward and backward regenerate the SAME mask: the analytic gradient matches a
    finite difference of the (fixed-seed) dropped forward."""
    from allrank_amd import ops
    rng = np.random.default_rng(0)
    B, L, h, dk = 2, 96, 2, 32
    d = h * dk
    qkv = (rng.standard_normal((B, L, 3 * d)) * 0.5).astype(np.float32)
    mask = torch.zeros(B, L, dtype=torch.bool, device=DEV)
    t0 = _t(qkv)
    base = ops.attention(t0[:, :, :d], t0[:, :, d:2 * d], t0[:, :, 2 * d:], mask, h, 0.0)
    same = ops.attention(t0[:, :, :d], t0[:, :, d:2 * d], t0[:, :, 2 * d:], mask, h, 0.0, seed=123)
    assert torch.equal(base, same)
    # (b) v = ones  =>  out = (1/(1-p)) * sum_j keep_ij P_ij : its mean over rows is ~1 and the kept mass fraction ~ 1-p
    ones = qkv.copy()
    ones[:, :, 2 * d:] = 1.0
    t1 = _t(ones)
    p = 0.3
    outs = [ops.attention(t1[:, :, :d], t1[:, :, d:2 * d], t1[:, :, 2 * d:], mask, h, p, seed=s).mean().item() for s in range(8)]
    assert abs(np.mean(outs) - 1.0) < 0.02, outs
    o1 = ops.attention(t1[:, :, :d], t1[:, :, d:2 * d], t1[:, :, 2 * d:], mask, h, p, seed=7)
    o2 = ops.attention(t1[:, :, :d], t1[:, :, d:2 * d], t1[:, :, 2 * d:], mask, h, p, seed=7)
    o3 = ops.attention(t1[:, :, :d], t1[:, :, d:2 * d], t1[:, :, 2 * d:], mask, h, p, seed=8)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # (c) directional derivative check with a fixed seed (float64 reference is not available: use central differences)
    tq = _t(qkv, True)
    go = _t(rng.standard_normal((B, L, d)).astype(np.float32))
    o = ops.attention(tq[:, :, :d], tq[:, :, d:2 * d], tq[:, :, 2 * d:], mask, h, p, seed=99)
    (o * go).sum().backward()
    direction = rng.standard_normal(qkv.shape).astype(np.float32)
    eps = 1e-2

    def f(x):
        t = _t(x)
        return float((ops.attention(t[:, :, :d], t[:, :, d:2 * d], t[:, :, 2 * d:], mask, h, p, seed=99) * go).sum().item())

    fd = (f(qkv + eps * direction) - f(qkv - eps * direction)) / (2 * eps)
    an = float((tq.grad.cpu().numpy() * direction).sum())
    assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (fd, an)


def test_model_trains_with_reference_dropout_config():
    """the reference's WEB30K ranker (fc 96, N=2, h=1, d_ff=384, dropout 0.1 -- approxndcg.json) runs train() steps through
    the plugin surface and reduces the loss on a fixed batch."""
    from allrank_amd import losses as E
    from allrank_amd.model import make_model
    from allrank_amd.engine import Trainer
    torch.manual_seed(0)
    model = make_model(dict(sizes=[96], input_norm=False, activation=None, dropout=0.0),
                       dict(N=2, d_ff=384, h=1, positional_encoding=dict(strategy="fixed", max_indices=240), dropout=0.1),
                       dict(d_output=1, output_activation=None), 136).to(DEV)
    model.train()
    rng = np.random.default_rng(1)
    x = _t(rng.standard_normal((16, 60, 136)).astype(np.float32))
    y = _t(rng.integers(0, 5, (16, 60)).astype(np.float32))
    idx = torch.arange(60, device=DEV).expand(16, 60).contiguous()
    tr = Trainer(model, E.approxNDCGLoss, torch.optim.Adam(model.parameters(), lr=1e-3))
    losses = [tr.step(x, y, idx).item() for _ in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.02, losses


def test_end_to_end_training_on_device_resident_libsvm_data(tmp_path):
    """libsvm file -> DeviceSlates (HBM) -> on-device FixLength batches -> FusedTrainer epochs -> NDCG@5 on a held-out set
    improves.  Data: the reference's dummy-data recipe (generate_dummy_data.py:10-18: N(0,1) features, label =
    clip(int(mean((X+1)/2) * num_labels)))."""
    from sklearn.datasets import dump_svmlight_file
    from allrank_amd.data import DeviceSlates, evaluate
    from allrank_amd.engine import fit_device
    from allrank_amd.model import make_model
    rng = np.random.default_rng(0)

    def dummy(n_q, seed):
        r = np.random.default_rng(seed)
        lens = r.integers(8, 30, n_q)
        X = r.standard_normal((lens.sum(), 20)).astype(np.float32)
        yy = np.clip((np.mean((X + 1) / 2, axis=1) * 5).astype(int), 0, 4).astype(np.float32)
        return X, yy, np.repeat(np.arange(n_q), lens)

    paths = {}
    for role, (n_q, seed) in dict(train=(256, 1), vali=(64, 2)).items():
        X, yy, qid = dummy(n_q, seed)
        paths[role] = str(tmp_path / (role + ".txt"))
        dump_svmlight_file(X, yy, paths[role], query_id=qid)
    train = DeviceSlates.from_svm_file(paths["train"], DEV)
    vali = DeviceSlates.from_svm_file(paths["vali"], DEV)
    torch.manual_seed(42)
    model = make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                       dict(N=1, d_ff=64, h=2, positional_encoding=None, dropout=0.0),
                       dict(d_output=1, output_activation=None), 20).to(DEV)
    before = evaluate(model, vali, {"ndcg": [5]})["ndcg_5"]
    g = torch.Generator(device=DEV).manual_seed(0)
    res = fit_device(model, "approxNDCGLoss", {}, train, vali, epochs=6, batch_size=32, slate_length=24, metrics={"ndcg": [5, 10]},
                     lr=2e-3, generator=g, gradient_clipping_norm=5.0, lr_schedule=lambda e: 2e-3 * (0.5 ** (e // 4)))
    after = res["val_metrics"]["ndcg_5"]
    _log("fit_device", dict(before=before, after=after, history=res["history"], fused=res["fused"]))
    assert res["fused"] and np.isfinite(after) and after > before + 0.03, (before, res["history"])
    h = res["history"]
    assert "train_ndcg_5" in h[-1] and 0.0 < h[-1]["train_ndcg_5"] <= 1.0 and h[-1]["train_ndcg_5"] > h[0]["train_ndcg_5"]
    # the same run with variable-length execution (padded slots skipped): same data order (same generator seed), same result
    torch.manual_seed(42)
    model_c = make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                         dict(N=1, d_ff=64, h=2, positional_encoding=None, dropout=0.0),
                         dict(d_output=1, output_activation=None), 20).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    res_c = fit_device(model_c, "approxNDCGLoss", {}, train, vali, epochs=6, batch_size=32, slate_length=24, metrics={"ndcg": [5, 10]},
                       lr=2e-3, generator=g, gradient_clipping_norm=5.0, lr_schedule=lambda e: 2e-3 * (0.5 ** (e // 4)), compact=True)
    assert res_c["fused"]
    assert abs(res_c["history"][0]["train_loss"] - h[0]["train_loss"]) < 2e-3        # epoch 0: round-off level differences only
    assert abs(res_c["val_metrics"]["ndcg_5"] - after) < 0.03, (res_c["val_metrics"], after)


# ------------------------------------------------------------------------------------------------------------------
# dropout inside the explicit step (counter-based masks regenerated in the backward; device-side step word)
# ------------------------------------------------------------------------------------------------------------------
def test_dropout_sites_share_one_counter_based_mask():
    """the GEMM epilogue, the LayerNorm residual add and ltrx_dropout_apply key the SAME mask off (seed, step word,
    element index): epilogue dropout == plain output * apply-mask bit for bit; keep rate = 1 - p; the step word re-keys."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(0)
    Mm, N, K, p, seed = 700, 192, 136, 0.3, 0xC0FFEE
    A, Bw, bias = _t(rng.standard_normal((Mm, K)).astype(np.float32)), _t(rng.standard_normal((N, K)).astype(np.float32)), _t(rng.standard_normal(N).astype(np.float32))
    step = torch.tensor([7], dtype=torch.int32, device=DEV)
    C0, C1, Mk = (torch.empty((Mm, N), device=DEV) for _ in range(3))
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, None, LB.ptr(C0), N, Mm, N, K, LB.ptr(bias), 1, None, 0, 0.0, 0, None, 0, 0, None), "nt")
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, None, LB.ptr(C1), N, Mm, N, K, LB.ptr(bias), 1, None, 0, p, seed, LB.ptr(step), 0, 0, None), "nt")
    ones = torch.ones((Mm, N), device=DEV)
    LB.check(lib.ltrx_dropout_apply(LB.ptr(ones), LB.ptr(Mk), ones.numel(), p, seed, LB.ptr(step), None), "apply")
    torch.cuda.synchronize()
    keep = (Mk > 0).float().mean().item()
    assert abs(keep - (1 - p)) < 0.01, keep
    vals = torch.unique(Mk).cpu().numpy()
    assert len(vals) == 2 and vals[0] == 0 and abs(vals[1] - 1 / (1 - p)) < 1e-6, vals
    assert torch.equal(C1, C0 * Mk)
    # act == 2 (ReLU+dropout backward): mask carried by aux, scale 1/(1-p)
    G2 = torch.empty((Mm, N), device=DEV)
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, None, LB.ptr(G2), N, Mm, N, K, None, 2, LB.ptr(C1), N, p, 0, None, 0, 0, None), "nt")
    G0 = torch.empty((Mm, N), device=DEV)
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, None, LB.ptr(G0), N, Mm, N, K, None, 0, None, 0, 0.0, 0, None, 0, 0, None), "nt")
    torch.cuda.synchronize()
    assert torch.equal(G2, torch.where(C1 > 0, G0 * np.float32(1 / (1 - p)), torch.zeros_like(G0)))
    # a different step word -> a different, equally dense mask; NULL step word == step word 0
    step2 = torch.tensor([8], dtype=torch.int32, device=DEV)
    Mk2, Mk0, Mkn = (torch.empty((Mm, N), device=DEV) for _ in range(3))
    LB.check(lib.ltrx_dropout_apply(LB.ptr(ones), LB.ptr(Mk2), ones.numel(), p, seed, LB.ptr(step2), None), "apply")
    LB.check(lib.ltrx_dropout_apply(LB.ptr(ones), LB.ptr(Mk0), ones.numel(), p, seed, LB.ptr(torch.zeros(1, dtype=torch.int32, device=DEV)), None), "apply")
    LB.check(lib.ltrx_dropout_apply(LB.ptr(ones), LB.ptr(Mkn), ones.numel(), p, seed, None, None), "apply")
    torch.cuda.synchronize()
    agree = ((Mk2 > 0) == (Mk > 0)).float().mean().item()
    assert abs(agree - ((1 - p) ** 2 + p ** 2)) < 0.01, agree          # independent masks
    assert torch.equal(Mk0, Mkn)
    # LayerNorm residual add: xsum = x + drop(res) with the same mask (D = 256 -> vectorised kernel; D = 96 -> generic)
    for D in (256, 96):
        rows = 300
        x, res = _t(rng.standard_normal((rows, D)).astype(np.float32)), _t(rng.standard_normal((rows, D)).astype(np.float32))
        a, b = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
        xs, y, mean, rstd, mk = torch.empty_like(x), torch.empty_like(x), torch.empty(rows, device=DEV), torch.empty(rows, device=DEV), torch.empty_like(x)
        LB.check(lib.ltrx_layernorm_fwd(LB.ptr(x), LB.ptr(res), LB.ptr(a), LB.ptr(b), rows, D, 1e-6, LB.ptr(xs), LB.ptr(y), LB.ptr(mean),
                                        LB.ptr(rstd), p, seed, LB.ptr(step), None), "ln")
        LB.check(lib.ltrx_dropout_apply(LB.ptr(res), LB.ptr(mk), res.numel(), p, seed, LB.ptr(step), None), "apply")
        torch.cuda.synchronize()
        assert torch.equal(xs, x + mk), D
        ref = xs.double()
        mu = ref.mean(1, keepdim=True)
        yr = (ref - mu) / (ref.std(1, keepdim=True) + 1e-6)
        assert (y.double() - yr).abs().max().item() < 2e-5


def test_numpy_dropout_masks_equal_the_kernels_bit_for_bit():
    """oracle/dropout_oracle.py restates the engine's counter-based masks: the element mask of the GEMM epilogues / LayerNorm residual
    / ltrx_dropout_apply, and the attention mask -- read back from an attention call whose probabilities are uniform (q = k = 0) and
    whose value rows are one-hot, so that O[query, key] = mask[query, key] / L -- for the resident three-product kernels (mode 1) and
    the exact-fp32 kernels (mode 0)."""
    from allrank_amd import _lib as LB
    from oracle import dropout_oracle as D
    lib = LB.lib()
    for (shape, p, seed, word) in [((700, 192), 0.3, 0xC0FFEE, 7), ((1000, 2048), 0.1, 123456789, 1), ((33, 40), 0.5, 5, 0)]:
        ones = torch.ones(shape, device=DEV)
        mk = torch.empty(shape, device=DEV)
        step = torch.tensor([word], dtype=torch.int32, device=DEV)
        LB.check(lib.ltrx_dropout_apply(LB.ptr(ones), LB.ptr(mk), ones.numel(), p, seed, LB.ptr(step), None), "apply")
        assert np.array_equal(mk.cpu().numpy(), D.keep_scale(p, seed, word, shape)), (shape, p)
    B, L, h, dk, p, seed, word = 3, 64, 2, 64, 0.25, 0xABCDE, 5
    d = h * dk
    qkv = torch.zeros(B * L, 3 * d, device=DEV)
    qkv[:, 2 * d:] = torch.eye(L, device=DEV).repeat(B, h)              # V[b, j, head, :] = one-hot(j)
    mask = torch.zeros(B, L, dtype=torch.uint8, device=DEV)
    step = torch.tensor([word], dtype=torch.int32, device=DEV)
    ref = D.attention_keep_scale(p, seed, word, B, h, L)
    for mode in (1, 0):
        o, lse = torch.empty(B * L, d, device=DEV), torch.empty(B, h, L, device=DEV)
        LB.check(lib.ltrx_mha_fwd(LB.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, LB.ptr(mask), B, L, h, dk, 3 * d, LB.ptr(o), d,
                                  LB.ptr(lse), p, seed, LB.ptr(step), None, None, mode, None), "mha_fwd")
        got = (o.view(B, L, h, dk).permute(0, 2, 1, 3) * L).cpu().numpy()       # [B, h, query, key]
        assert np.array_equal(got > 0, ref > 0), mode
        assert np.abs(got - ref).max() <= 1e-5, (mode, np.abs(got - ref).max())


@pytest.mark.parametrize("d,h,fc_act", [(64, 1, "ReLU"), (32, 2, None)])
def test_fused_step_with_dropout_matches_the_fp64_oracle_under_the_same_masks(d, h, fc_act):
    """training-mode parity was statistical only (the reference draws its masks from torch's generator).  The
    engine's masks are pure functions of (site seed, step word, element index), so the fp64 oracle is handed exactly the masks each
    step used (oracle/dropout_oracle.engine_masks) and the step -- FC dropout, attention-probability dropout, feed-forward dropout and
    both residual-branch dropouts active -- is compared with it at the engine's weights: loss within 1e-5, scores, every parameter
    gradient on the engine's ReLU branch; three steps (eager, eager, captured) each with its own masks."""
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    from oracle import dropout_oracle as D
    rng = np.random.default_rng(17)
    B, L, F, dff = 8, 40, 20, 64
    x = rng.standard_normal((B, L, F)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[3, 25:] = -1
    x[3, 25:] = 0
    mask = y == -1
    torch.manual_seed(33)
    model = make_model(dict(sizes=[48, d], input_norm=False, activation=fc_act, dropout=0.1),
                       dict(N=2, d_ff=dff, h=h, positional_encoding=None, dropout=0.2),
                       dict(d_output=1, output_activation=None), F).to(DEV)
    cfg = dict(n_features=F, fc_sizes=[48, d], fc_activation=fc_act, fc_input_norm=False, N=2, d_ff=dff, h=h, output_activation=None)
    ft = FusedTrainer(model, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=True, seed=77)
    assert ft._any_dropout and ft.p_fc == 0.1 and ft.layers[0]["p_att"] == 0.2
    named = dict(model.named_parameters())
    xt, yt = _t(x), _t(y)
    for step in range(3):
        w = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in named.items()}
        loss = float(ft.step(xt, yt).item())
        word = int(ft.drop_step.item())
        assert word == step + 1
        masks = D.engine_masks(ft, word)
        so, cache = M.forward(w, cfg, x.astype(np.float64), mask, masks)
        lo, gs = O.approxndcg(so, y, dtype=np.float64)[:2]
        assert abs(loss - lo) <= 1e-5 * (1 + abs(lo)), (step, loss, lo)
        sc = ft.scores.cpu().numpy().astype(np.float64)
        # (scores: 5e-5 of their scale -- measured 2.1e-5; the no-dropout bar of 2e-5 is exceeded by the 1 / (1 - p) multipliers of five
        #  dropout sites on top of the three-product round-off.  A mask that differed in ONE element would show as O(1e-2).)
        assert np.abs(sc - so)[~mask].max() <= 5e-5 * max(1.0, np.abs(so[~mask]).max()), (step, np.abs(sc - so)[~mask].max())
        pats = [(st["r"] > 0).view(B, L, -1).cpu().numpy() for st in ft.layers]
        fcp = [(t > 0).view(B, L, -1).cpu().numpy() for t in ft.fc_out] if fc_act == "ReLU" else None
        g_or = M.backward(w, cfg, cache, np.asarray(gs, dtype=np.float64), relu_masks=pats, fc_relu_masks=fcp)
        gmax = max(float(np.abs(v).max()) for v in g_or.values())
        for k, v in named.items():
            ge = v.grad.detach().cpu().numpy().astype(np.float64)
            own = float(np.abs(g_or[k]).max())
            err = float(np.abs(ge - g_or[k]).max())
            # (the second term is the floor for tensors whose TRUE gradient is 0 -- the key-projection bias: softmax is invariant to a
            #  per-query shift -- where the engine returns round-off of the attention backward: measured 1.1e-6 of the model's largest
            #  gradient with round 6's forward (one S accumulator, lazy softmax reference), 0.6e-6 before)
            assert err <= 1e-3 * own + 3e-6 * gmax, (step, k, err, own)


def _dropout_model(p, fc_act, fc_drop, N=2):
    from allrank_amd.model import make_model
    return make_model(dict(sizes=[48, 32], input_norm=False, activation=fc_act, dropout=fc_drop),
                      dict(N=N, d_ff=64, h=2, positional_encoding=None, dropout=p) if N else None,
                      dict(d_output=1, output_activation=None), 20).to(DEV)


@pytest.mark.parametrize("gemm,fc_act,compact", [("split_bf16_strict", None, False), ("hipblaslt", "ReLU", False),
                                                 ("split_bf16_strict", "ReLU", False), ("split_bf16_strict", "ReLU", True),
                                                 ("hipblaslt", None, True)])
def test_fused_step_dropout_gradients_match_finite_differences(gemm, fc_act, compact):
    """with the masks frozen (fixed seed and step word) the explicit backward must be the gradient of the explicit forward:
    for every parameter tensor, the central difference of the loss along that tensor's own gradient direction equals the
    gradient norm.  A forward/backward mask mismatch at any of the dropout sites breaks this by O(1)."""
    from allrank_amd.engine import FusedTrainer
    torch.manual_seed(5)
    model = _dropout_model(0.25, fc_act, 0.2)
    B, L = 6, 20
    rng = np.random.default_rng(2)
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[0, 15:] = -1
    if compact:                                # variable-length execution: packed rows, cu_seqlens attention (same mask algebra)
        y[3, 4:] = -1
        y[5, 19:] = -1
    ft = FusedTrainer(model, "listNet", {}, B, L, use_graph=False, gemm=gemm, seed=1234, compact=compact)
    ft._divisor = float(B)
    ft.y_in.copy_(_t(y))
    ft.mask.copy_(_t(y) == -1)
    if compact:
        ft._pack(_t(x).reshape(B * L, -1).contiguous(), [int(v) for v in (y != -1).sum(1)])
    else:
        ft.x_in.copy_(_t(x).reshape(B * L, -1))
    base = float(ft._body().item())
    again = float(ft._body().item())
    assert base == again                                     # same step word -> same masks -> same loss
    g = ft.flat_g.clone()
    p0 = ft.flat_p.clone()
    rows, bad = [], []
    for name, prm in model.named_parameters():
        o, shape = ft._pv[id(prm)]
        n = prm.numel()
        gn = float(g[o:o + n].norm().item())
        if gn < 2e-3:
            continue
        eps = 5e-3
        v = torch.zeros_like(p0)
        v[o:o + n] = g[o:o + n] / gn
        ft.flat_p.copy_(p0 + eps * v)
        lp = float(ft._body().item())
        ft.flat_p.copy_(p0 - eps * v)
        lm = float(ft._body().item())
        fd = (lp - lm) / (2 * eps)
        rows.append(dict(param=name, grad_norm=gn, fd=fd))
        if abs(fd - gn) > 0.04 * gn + 5e-4:       # ReLU kinks along a bias direction cost ~2%
            bad.append(rows[-1])
    ft.flat_p.copy_(p0)
    _log("dropout_fd_%s_%s_%s" % (gemm, fc_act, compact), rows)
    assert len(rows) >= 10 and not bad, bad
    # and the dropout is really on: a no-dropout trainer on the same weights gives a different loss
    ft2 = FusedTrainer(model, "listNet", {}, B, L, use_graph=False, gemm=gemm, dropout=False)
    ft2._divisor = float(B)
    ft2.x_in.copy_(_t(x).reshape(B * L, -1)); ft2.y_in.copy_(ft.y_in); ft2.mask.copy_(ft.mask)
    assert abs(float(ft2._body().item()) - base) > 1e-4


def test_fused_step_dropout_off_equals_eval_forward_and_graph_replays_fresh_masks():
    from allrank_amd.engine import FusedTrainer
    from allrank_amd import losses as E
    torch.manual_seed(6)
    model = _dropout_model(0.3, None, 0.0)
    B, L = 8, 24
    rng = np.random.default_rng(3)
    x = _t(rng.standard_normal((B, L, 20)).astype(np.float32))
    y = _t(rng.integers(0, 5, (B, L)).astype(np.float32))
    model.eval()
    with torch.no_grad():
        want = float(E.listNet(model(x, y == -1, None), y).item())
    off = FusedTrainer(model, "listNet", {}, B, L, lr=0.0, use_graph=True, dropout=False)
    on = FusedTrainer(model, "listNet", {}, B, L, lr=0.0, use_graph=True, seed=99)
    l_off = [float(off.step(x, y).item()) for _ in range(6)]
    l_on = [float(on.step(x, y).item()) for _ in range(8)]
    assert all(abs(v - want) <= 1e-5 * (1 + abs(want)) for v in l_off), (want, l_off)
    assert on.graph is not None and len(set(l_on[3:])) == len(l_on[3:]), l_on       # replays draw new masks
    assert abs(np.mean(l_on) - want) < 0.5 * abs(want)


def test_fused_trainer_trains_reference_dropout_config():
    """the reference's WEB30K ranker shape (fc 96, N=2, h=1 -> here h=2, d_ff=384, dropout 0.1; approxndcg.json) through the
    hipGraph-captured explicit step WITH dropout: the loss on a fixed batch goes down."""
    from allrank_amd.engine import FusedTrainer
    from allrank_amd.model import make_model
    torch.manual_seed(0)
    model = make_model(dict(sizes=[96], input_norm=False, activation=None, dropout=0.0),
                       dict(N=2, d_ff=384, h=2, positional_encoding=None, dropout=0.1),
                       dict(d_output=1, output_activation=None), 136).to(DEV)
    rng = np.random.default_rng(1)
    x = _t(rng.standard_normal((16, 60, 136)).astype(np.float32))
    y = _t(rng.integers(0, 5, (16, 60)).astype(np.float32))
    ft = FusedTrainer(model, "approxNDCGLoss", {}, 16, 60, lr=1e-3, use_graph=True)
    losses = [float(ft.step(x, y).item()) for _ in range(30)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < np.mean(losses[:5]) - 0.02, losses


# ------------------------------------------------------------------------------------------------------------------
# pointwise / pairwise losses, MRR, stochastic NeuralSort
# ------------------------------------------------------------------------------------------------------------------
def _extra_engine(kind, kw, yp, yt):
    from allrank_amd import losses as E
    fn = dict(ranknet=E.rankNet, bce=E.bce, ordinal=E.ordinal, pointwise_rmse=E.pointwise_rmse, binary_listnet=E.binary_listNet)[kind]
    p = _t(yp, True)
    l = fn(p, _t(yt), **kw)
    l.backward()
    return float(l.item()), p.grad.cpu().numpy()


_EXTRA_ORACLE = dict(ranknet=O.ranknet, bce=O.bce, ordinal=O.ordinal, pointwise_rmse=O.pointwise_rmse, binary_listnet=O.binary_listnet)


def test_extra_losses_match_reference_golden(extra_golden):
    from tests.cases import iter_extra_cases
    rows, bad = [], []
    for name, kind, kw, yp, yt, rl, rg in iter_extra_cases(extra_golden):
        l, g = _extra_engine(kind, kw, yp, yt)
        ok = close(l, rl) and grad_close(g, rg)
        rows.append(dict(case=name, loss=l, ref=float(rl), grad_err=float(np.abs(g - rg).max()), ok=ok))
        if not ok:
            bad.append(rows[-1])
    _log("extra_losses_golden", rows)
    assert len(rows) == 36 and not bad, bad


@pytest.mark.parametrize("B,L", [(9, 37), (64, 240), (3, 700)])
def test_extra_losses_match_oracle_on_random_inputs(B, L):
    rng = np.random.default_rng(B * 1000 + L)
    s = rng.standard_normal((B, L)).astype(np.float32) * 2
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    for b in range(B):
        y[b, L - (b * 7) % L:] = -1 if b % 3 else y[b, L - (b * 7) % L:]
    y[0] = 0                                                      # a slate without any pair / relevant item
    p = (1 / (1 + np.exp(-s))).astype(np.float32)
    p3 = (1 / (1 + np.exp(-rng.standard_normal((B, L, 3))))).astype(np.float32)
    yb = np.where(y == -1, -1, (y >= 3)).astype(np.float32)
    cases = [("ranknet", dict(), s, y), ("ranknet", dict(weight_by_diff=True), s, y), ("ranknet", dict(weight_by_diff_powed=True), s, y),
             ("bce", {}, p, yb), ("ordinal", dict(n=3), p3, y), ("pointwise_rmse", dict(no_of_levels=4), p, y),
             ("binary_listnet", {}, s, yb)]
    for kind, kw, yp, yt in cases:
        l, g = _extra_engine(kind, kw, yp, yt)
        lo, go = _EXTRA_ORACLE[kind](yp, yt, **kw)
        assert close(l, lo), (kind, kw, l, lo)
        assert grad_close(g, go), (kind, kw, float(np.abs(g - go).max()))
        assert not np.any(g[yt == -1] != 0), kind           # exactly 0 at padded slots


def test_ranknet_without_pairs_is_nan_with_zero_gradient():
    from allrank_amd import losses as E
    p = _t(np.asarray([[0.3, 0.1, 0.2]], np.float32), True)
    l = E.rankNet(p, _t(np.asarray([[1.0, 1.0, -1.0]], np.float32)))
    l.backward()
    assert np.isnan(l.item()) and not np.any(p.grad.cpu().numpy() != 0)


def test_mrr_matches_reference_golden_and_oracle(extra_golden):
    from allrank_amd import metrics as EM
    g = extra_golden
    for ci in range(int(g["n_cases"])):
        pre = "c%d." % ci
        s, y = g[pre + "s"], g[pre + "y"]
        ats = [int(a) for a in g[pre + "mrr.ats"]]
        assert np.array_equal(EM.mrr(_t(s), _t(y), ats=ats).cpu().numpy(), g[pre + "mrr.val"])
        assert np.array_equal(EM.mrr(_t(s), _t(y)).cpu().numpy(), g[pre + "mrr.none"])
        yz = np.where(y == -1, -1, 0).astype(np.float32)
        assert np.array_equal(EM.mrr(_t(s), _t(yz), ats=ats).cpu().numpy(), g[pre + "mrr.zero"])
    rng = np.random.default_rng(5)
    s = np.round(rng.standard_normal((200, 240)), 1).astype(np.float32)            # tie-heavy predictions
    y = rng.integers(0, 5, (200, 240)).astype(np.float32)
    y[::3, 100:] = -1
    assert np.array_equal(EM.mrr(_t(s), _t(y), ats=[1, 5, 10, 240]).cpu().numpy(), O.mrr(s, y, [1, 5, 10, 240]))


def test_stochastic_neuralndcg_matches_reference_golden(extra_golden):
    """n_samples Gumbel-perturbed copies per slate through the fused NeuralSort/Sinkhorn kernels, same draw as the reference
    run that made the fixture (including its sort-mask / read-out-mask mismatch on ragged batches)."""
    from allrank_amd import losses as E
    from tests.cases import iter_stochastic_cases
    rows, bad = [], []
    for name, c, s, y, gum, rl, rg, strict in iter_stochastic_cases(extra_golden):
        fn = E.neuralNDCG_transposed if c["tr"] else E.neuralNDCG
        p = _t(s, True)
        l = fn(p, _t(y), temperature=c["tau"], k=c["k"], powered_relevancies=c["pw"], stochastic=True, n_samples=gum.shape[0],
               beta=c["beta"], log_scores=c["log"], gumbel=_t(gum[..., None]))
        l.backward()
        g = p.grad.cpu().numpy()
        ok = close(l.item(), rl) and grad_close(np.where(strict, g, 0), np.where(strict, rg, 0)) and bool(np.isfinite(g).all())
        rows.append(dict(case=name, loss=float(l.item()), ref=float(rl), ok=ok,
                         grad_err=float(np.abs(np.where(strict, g - rg, 0)).max()), grad_scale=float(np.abs(rg).max())))
        if not ok:
            bad.append(rows[-1])
    _log("stochastic_neuralndcg_golden", rows)
    assert len(rows) == 16 and not bad, bad


def test_stochastic_neuralndcg_draws_its_own_noise():
    from allrank_amd import losses as E
    rng = np.random.default_rng(0)
    s, y = _t(rng.standard_normal((8, 60)).astype(np.float32), True), _t(rng.integers(0, 5, (8, 60)).astype(np.float32))
    torch.manual_seed(1)
    a = E.neuralNDCG(s, y, stochastic=True, n_samples=4)
    b = E.neuralNDCG(s, y, stochastic=True, n_samples=4)
    det = E.neuralNDCG(s, y)
    a.backward()
    assert a.item() != b.item() and abs(a.item() - det.item()) < 0.2 and np.isfinite(s.grad.cpu().numpy()).all()


@pytest.mark.parametrize("loss_name,loss_args", [("rankNet_weightByGTDiff", {}), ("binary_listNet", {}), ("pointwise_rmse", dict(no_of_levels=4))])
def test_fused_trainer_runs_the_row4_losses(loss_name, loss_args):
    """the explicit step with a pointwise / pairwise loss == the autograd path with the same loss function."""
    import copy
    from allrank_amd import losses as E
    from allrank_amd.engine import FusedTrainer, Trainer
    torch.manual_seed(3)
    m1 = _dropout_model(0.0, None, 0.0, N=1)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(8)
    B, L = 6, 30
    x = _t(rng.standard_normal((B, L, 20)).astype(np.float32))
    y = rng.integers(0, 2, (B, L)).astype(np.float32)
    y[1, 20:] = -1
    y = _t(y)
    ft = FusedTrainer(m1, loss_name, loss_args, B, L, lr=1e-3, use_graph=False, gemm="split_bf16_strict")
    from functools import partial
    tr = Trainer(m2, partial(getattr(E, loss_name), **loss_args), torch.optim.Adam(m2.parameters(), lr=1e-3))
    for step in range(3):
        lf, la = float(ft.step(x, y).item()), float(tr.step(x, y, None).item())
        assert abs(lf - la) <= (2e-5 if step == 0 else 1e-3) * (1 + abs(la)), (loss_name, step, lf, la)


def test_row4_kats_from_the_reference_tests():
    from allrank_amd import losses as E, metrics as EM
    from tests.cases import row4_kats, mrr_kats
    for kind, kw, yp, yt, expected in row4_kats():
        got = _extra_engine(kind, kw, np.asarray([yp], np.float32), np.asarray([yt], np.float32))[0]
        assert np.isfinite(got) and abs(got - expected) <= 1e-5 * (1 + abs(expected)), (kind, kw, yp, got, expected)
    for yp, yt, ats, expected in mrr_kats():
        got = EM.mrr(_t(np.asarray(yp, np.float32)), _t(np.asarray(yt, np.float32)), ats=ats).cpu().numpy()
        assert np.array_equal(got, np.asarray(expected, np.float32)), (yp, yt, ats, got)
    assert E.with_ordinals(_t(np.asarray([[2.0, 1.0, 0.0]], np.float32)), 2).tolist() == [[[1.0, 1.0], [1.0, 0.0], [0.0, 0.0]]]


def test_large_tile_gemm_matches_fp64_and_the_small_tile_kernel():
    """the 256 x 256 x 32 kernel (auto-selected for big exact-multiple shapes, forced here through the variant hook):
    same fp32-class error bound, all epilogues (bias, ReLU, ReLU(+dropout) mask, dropout) agree with the 128 x 128 kernel."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(11)
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    try:
        # forced: 6 = the 256-row tile, 7 = its 128-row form, 0 = whatever the dispatch picks (15360 x 512 -> 128-row tiles)
        for (Mm, N, K, forced) in [(512, 256, 32, 6), (256, 768, 544, 6), (1024, 512, 96, 6), (6144, 4096, 64, 0),
                                   (700, 512, 64, 6), (33, 256, 160, 6), (24000, 1536, 64, 0),       # ragged last row tile
                                   (384, 512, 96, 7), (200, 256, 64, 7), (15360, 512, 64, 0), (15300, 512, 32, 0),
                                   (40000, 512, 64, 0)]:      # one round of large tiles + the remaining rows (two launches)
            A = rng.standard_normal((Mm, K)).astype(np.float32)
            Bw = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32)
            aux = rng.standard_normal((Mm, N)).astype(np.float32)
            At, Bt, bt, auxt = _t(A), _t(Bw), _t(bias), _t(aux)
            outs = {}
            for variant in (1, forced):
                res = []
                for (b_, act, ax, p) in ((bt, 1, None, 0.0), (None, 0, None, 0.0), (None, 2, auxt, 0.25), (bt, 1, None, 0.3), (bt, 0, None, 0.3)):
                    C = torch.empty((Mm, N), device=DEV)
                    LB.check(lib.ltrx_gemm_nt(LB.ptr(At), K, LB.ptr(Bt), K, None, LB.ptr(C), N, Mm, N, K, LB.ptr(b_), act, LB.ptr(ax),
                                              N if ax is not None else 0, p, 77, LB.ptr(step), 0, variant, None), "gemm_nt")
                    res.append(C)
                outs[variant] = res
            v_new = forced
            ref = np.maximum(A.astype(np.float64) @ Bw.astype(np.float64).T + bias, 0)
            scale = (np.abs(A).astype(np.float64) @ np.abs(Bw).astype(np.float64).T).max()
            err = float(np.abs(outs[v_new][0].cpu().numpy() - ref).max() / scale)
            assert err < 6e-6, (Mm, N, K, err)           # K = 32: no averaging over the contraction
            for a, b in zip(outs[1], outs[v_new]):
                assert float((a - b).abs().max().item()) <= 2e-6 * scale + 1e-6, (Mm, N, K)
                assert torch.equal(a == 0, b == 0)                  # identical ReLU / dropout masks
    finally:
        pass


def test_gemm_nt_with_presplit_weight_image_is_bit_identical():
    """ltrx_split_image + the B_image argument of ltrx_gemm_nt (the weights of the explicit step are split once per optimizer step
    instead of in every tile): same bits as the fp32-operand call, every epilogue, both tile forms, ragged last row tile, one- and
    three-product arithmetic; shapes that run the small-tile kernel ignore the image."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(31)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for (Mm, N, K) in [(6144, 4096, 64), (700, 512, 96), (24000, 1536, 64), (15360, 512, 64), (40000, 512, 64), (300, 256, 136), (64, 96, 20)]:
        A = _t(rng.standard_normal((Mm, K)).astype(np.float32))
        Bw = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        bias = _t(rng.standard_normal(N).astype(np.float32))
        aux = _t(rng.standard_normal((Mm, N)).astype(np.float32))
        img = torch.empty_like(Bw)
        LB.check(lib.ltrx_split_image(LB.ptr(Bw), LB.ptr(img), Bw.numel(), None), "split_image")
        # the image really is {hi0..3, lo0..3} per 4 floats
        hl = img.view(torch.bfloat16).view(N, K // 4, 2, 4).float()
        ref_hi = Bw.view(N, K // 4, 4).to(torch.bfloat16).float()
        assert torch.equal(hl[:, :, 0], ref_hi) and torch.equal(hl[:, :, 1], (Bw.view(N, K // 4, 4) - ref_hi).to(torch.bfloat16).float())
        for prec in (0, 2):
            for (b_, act, ax, p) in ((bias, 1, None, 0.0), (None, 0, None, 0.0), (None, 2, aux, 0.25), (bias, 1, None, 0.3)):
                outs = []
                for image in (None, img):
                    C = torch.empty((Mm, N), device=DEV)
                    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(image), LB.ptr(C), N, Mm, N, K, LB.ptr(b_), act, LB.ptr(ax),
                                              N if ax is not None else 0, p, 77, LB.ptr(step), prec, 0, None), "gemm_nt")
                    outs.append(C)
                assert torch.equal(outs[0], outs[1]), (Mm, N, K, prec, act)


def test_gemm_nt_residual_epilogue_equals_gemm_then_add_bit_for_bit():
    """act 3 of ltrx_gemm_nt: C = aux + dropout_p(A B^T + bias), the SublayerConnection sum (transformer.py:98-106) written by
    the projection that closes the sublayer.  Same bits as the GEMM with the same dropout site followed by a separate fp32 add
    -- the composition the engine used before (the add lived in ltrx_layernorm_fwd) -- for both tile forms, a ragged last row
    tile, with and without dropout / weight image; an in-place residual (C == aux) is allowed."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(41)
    step = torch.full((1,), 3, dtype=torch.int32, device=DEV)
    for (Mm, N, K) in [(24000, 512, 512), (6144, 512, 2048), (700, 512, 96), (300, 256, 136), (64, 96, 20)]:
        A = _t(rng.standard_normal((Mm, K)).astype(np.float32))
        Bw = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        bias = _t(rng.standard_normal(N).astype(np.float32))
        res = _t(rng.standard_normal((Mm, N)).astype(np.float32))
        img = None
        if K % 4 == 0:
            img = torch.empty_like(Bw)
            LB.check(lib.ltrx_split_image(LB.ptr(Bw), LB.ptr(img), Bw.numel(), None), "split_image")
        for p in (0.0, 0.2):
            for image in ((None, img) if img is not None else (None,)):
                two = torch.empty((Mm, N), device=DEV)
                LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(image), LB.ptr(two), N, Mm, N, K, LB.ptr(bias), 0, None, 0, p, 91,
                                          LB.ptr(step), 0, 0, None), "gemm_nt")
                two = two + res
                one = torch.empty((Mm, N), device=DEV)
                LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(image), LB.ptr(one), N, Mm, N, K, LB.ptr(bias), 3, LB.ptr(res), N, p, 91,
                                          LB.ptr(step), 0, 0, None), "gemm_nt(act 3)")
                assert torch.equal(one, two), (Mm, N, K, p)
                inplace = res.clone()
                LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(image), LB.ptr(inplace), N, Mm, N, K, LB.ptr(bias), 3, LB.ptr(inplace), N,
                                          p, 91, LB.ptr(step), 0, 0, None), "gemm_nt(act 3, in place)")
                assert torch.equal(inplace, two), (Mm, N, K, p)
    C = torch.empty((64, 96), device=DEV)
    assert lib.ltrx_gemm_nt(LB.ptr(A), 20, LB.ptr(Bw), 20, None, LB.ptr(C), 96, 64, 96, 20, None, 3, None, 0, 0.0, 0, None, 0, 0, None) != 0   # act 3 needs aux
    assert lib.ltrx_gemm_nt(LB.ptr(A), 20, LB.ptr(Bw), 20, None, LB.ptr(C), 96, 64, 96, 20, None, 4, None, 0, 0.0, 0, None, 0, 0, None) != 0


def test_large_tile_wgrad_gemm_matches_fp64_and_the_small_tile_kernel():
    """the 256 x 256 split-K weight-gradient kernel (auto for NP, KP multiples of 256) vs fp64 and vs the 128 x 128 kernel."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(12)
    try:
        for (Mm, NP, KP) in [(2048, 256, 256), (4096, 512, 768), (15360, 2048, 512), (6176, 256, 512)]:
            A = rng.standard_normal((Mm, NP)).astype(np.float32)
            Bx = rng.standard_normal((Mm, KP)).astype(np.float32)
            At, Bt = _t(A), _t(Bx)
            ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(Mm, NP, KP), 64), dtype=torch.uint8, device=DEV)
            outs = {}
            for variant in (1, 0):
                C = torch.empty((NP, KP), device=DEV)
                gb = torch.empty(NP, device=DEV)
                LB.check(lib.ltrx_gemm_tn(LB.ptr(At), NP, LB.ptr(Bt), KP, LB.ptr(C), LB.ptr(gb), Mm, NP, KP, 0, variant, LB.ptr(ws), None), "gemm_tn")
                outs[variant] = (C, gb)
            ref = A.astype(np.float64).T @ Bx.astype(np.float64)
            scale = (np.abs(A).astype(np.float64).T @ np.abs(Bx).astype(np.float64)).max()
            assert float(np.abs(outs[0][0].cpu().numpy() - ref).max() / scale) < 4e-6, (Mm, NP, KP)
            assert float((outs[0][0] - outs[1][0]).abs().max().item()) <= 4e-6 * scale
            bref = A.astype(np.float64).sum(0)
            assert float(np.abs(outs[0][1].cpu().numpy() - bref).max()) < 1e-5 * max(1.0, np.abs(A).sum(0).max())
    finally:
        pass


def test_large_tile_wgrad_over_row_padded_features_matches_fp64():
    """ltrx_gemm_tn with KP = 136 columns of a B whose rows are padded to 256 floats (the engine's input buffer): the 256 x 256 kernel
    (opt-in: tile 9) computes the tile, only the 136 real columns reach the slabs and C (dense [NP, 136]); garbage in the padding does
    not matter; the same call on a dense B (ldb = 136) takes the small-tile kernel and agrees."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(13)
    for (Mm, NP, KP, ld) in [(15360, 512, 136, 256), (4096, 256, 300, 512), (6176, 512, 136, 256)]:
        A = rng.standard_normal((Mm, NP)).astype(np.float32)
        Bp = rng.standard_normal((Mm, ld)).astype(np.float32) * 100.0           # padding = garbage
        Bp[:, :KP] = rng.standard_normal((Mm, KP)).astype(np.float32)
        At, Bt = _t(A), _t(Bp)
        Bd = Bt[:, :KP].contiguous()
        ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(Mm, NP, KP), 64), dtype=torch.uint8, device=DEV)
        C1, g1 = torch.full((NP, KP), float("nan"), device=DEV), torch.empty(NP, device=DEV)
        C2, g2 = torch.full((NP, KP), float("nan"), device=DEV), torch.empty(NP, device=DEV)
        LB.check(lib.ltrx_gemm_tn(LB.ptr(At), NP, LB.ptr(Bt), ld, LB.ptr(C1), LB.ptr(g1), Mm, NP, KP, 0, 9, LB.ptr(ws), None), "gemm_tn(padded B)")
        LB.check(lib.ltrx_gemm_tn(LB.ptr(At), NP, LB.ptr(Bd), KP, LB.ptr(C2), LB.ptr(g2), Mm, NP, KP, 0, 0, LB.ptr(ws), None), "gemm_tn(dense B)")
        ref = A.astype(np.float64).T @ Bp[:, :KP].astype(np.float64)
        scale = (np.abs(A).astype(np.float64).T @ np.abs(Bp[:, :KP]).astype(np.float64)).max()
        assert float(np.abs(C1.cpu().numpy() - ref).max() / scale) < 4e-6, (Mm, NP, KP)
        assert float(np.abs(C2.cpu().numpy() - ref).max() / scale) < 4e-6, (Mm, NP, KP)
        bref = A.astype(np.float64).sum(0)
        assert float(np.abs(g1.cpu().numpy() - bref).max()) < 1e-5 * max(1.0, np.abs(A).sum(0).max())


def test_padded_input_rows_step_equals_the_dense_rows_step():
    """FusedTrainer(pad_input=True) (default: features in rows of 256 floats, first FC layer on the large-tile kernels) vs
    pad_input=False: same loss to GEMM round-off, gradients within the split-bf16 bound, eager and captured; also through
    variable-length execution."""
    import copy
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    rng = np.random.default_rng(8)
    B, L, F = 16, 240, 136
    x = _t(rng.standard_normal((B, L, F)).astype(np.float32))
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[3, 100:] = -1
    x[3, 100:] = 0
    yt = _t(y)
    torch.manual_seed(5)
    base = make_model(dict(sizes=[256], input_norm=False, activation=None, dropout=0.0),
                      dict(N=1, d_ff=512, h=4, positional_encoding=None, dropout=0.0),
                      dict(d_output=1, output_activation=None), F).to(DEV)
    for compact in (False, True):
        out = {}
        for pad in (True, False):
            m = copy.deepcopy(base)
            ft = FusedTrainer(m, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=True, pad_input=pad, compact=compact)
            assert ft._x_pad == pad and (ft.x_in.stride(0) == 256) == pad
            losses = [ft.step(x, yt).item()]
            g0 = {k: p.grad.detach().clone() for k, p in m.named_parameters()}          # gradients of the first step (equal weights)
            losses += [ft.step(x, yt).item() for _ in range(3)]                          # eager, capture, replay
            out[pad] = (losses, g0)
        assert abs(out[True][0][0] - out[False][0][0]) <= 2e-6 * (1 + abs(out[False][0][0])), (compact, out[True][0], out[False][0])
        assert abs(out[True][0][3] - out[False][0][3]) <= 2e-3 * (1 + abs(out[False][0][3]))
        floor = 1e-2 * max(float(r.abs().max()) for r in out[False][1].values())
        for k, g in out[True][1].items():
            ref = out[False][1][k]
            assert float((g - ref).abs().max()) <= 5e-5 * max(floor, float(ref.abs().max())), (compact, k)


def test_grouped_wgrad_launch_matches_fp64_is_deterministic_and_falls_back():
    """ltrx_gemm_tn_group: the four weight gradients of an encoder layer in one launch -- every result vs fp64 (same bound as the
    single-problem kernel), bit-identical run to run, bias sums present or absent per problem, strided operands (the fused QKV
    buffer), a ragged row count, and the per-problem fallback for shapes the large tile does not take."""
    import ctypes
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(21)
    cases = [(15360, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]),      # the bench layer at 64 slates of 240
             (6176, [(768, 256), (256, 256), (512, 256), (256, 512)]),          # ragged rows
             (4096, [(256, 256)]),
             (2048, [(96, 136), (256, 256)])]                                   # first shape not a large-tile shape: fallback
    for (Mm, probs) in cases:
        n = len(probs)
        As = [rng.standard_normal((Mm, a + 8)).astype(np.float32) for a, _ in probs]        # lda = NP + 8
        Bs = [rng.standard_normal((Mm, b)).astype(np.float32) for _, b in probs]
        At, Bt = [_t(a) for a in As], [_t(b) for b in Bs]
        NP = (ctypes.c_int * n)(*[a for a, _ in probs])
        KP = (ctypes.c_int * n)(*[b for _, b in probs])
        lda = (ctypes.c_int * n)(*[a + 8 for a, _ in probs])
        ldb = (ctypes.c_int * n)(*[b for _, b in probs])
        nb = lib.ltrx_gemm_tn_group_workspace_bytes(n, Mm, NP, KP)
        assert nb >= max(lib.ltrx_gemm_tn_workspace_bytes(Mm, a, b) for a, b in probs)
        ws = torch.empty(max(nb, 64), dtype=torch.uint8, device=DEV)
        runs = []
        for rep in range(2):
            Cs = [torch.full((a, b), float("nan"), device=DEV) for a, b in probs]
            gbs = [torch.full((a,), float("nan"), device=DEV) if (i % 2 == 0) else None for i, (a, _) in enumerate(probs)]
            vp = ctypes.c_void_p * n
            LB.check(lib.ltrx_gemm_tn_group(n, vp(*[t.data_ptr() for t in At]), lda, vp(*[t.data_ptr() for t in Bt]), ldb,
                                            vp(*[t.data_ptr() for t in Cs]), vp(*[(g.data_ptr() if g is not None else None) for g in gbs]),
                                            Mm, NP, KP, 0, LB.ptr(ws), ws.numel(), None, None, None, None), "gemm_tn_group")
            runs.append((Cs, gbs))
        # deferred form: the call leaves the slabs, ltrx_reduce_group sums them (all problems + an unrelated strided entry in one
        # launch) -- bit-identical to the call's own reduction
        Cs = [torch.full((a, b), float("nan"), device=DEV) for a, b in probs]
        gbs = [torch.full((a,), float("nan"), device=DEV) if (i % 2 == 0) else None for i, (a, _) in enumerate(probs)]
        so, bso, sp = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), ctypes.c_int(-1)
        LB.check(lib.ltrx_gemm_tn_group(n, vp(*[t.data_ptr() for t in At]), lda, vp(*[t.data_ptr() for t in Bt]), ldb,
                                        vp(*[t.data_ptr() for t in Cs]), vp(*[(g.data_ptr() if g is not None else None) for g in gbs]),
                                        Mm, NP, KP, 0, LB.ptr(ws), ws.numel(), so, bso, ctypes.byref(sp), None), "gemm_tn_group(deferred)")
        part = _t(rng.standard_normal((37, 2 * 200)).astype(np.float32))                 # [rows][da(200) | db(200)] partials
        da, db = torch.empty(200, device=DEV), torch.empty(200, device=DEV)
        ent = [(part.data_ptr(), 37, 400, 200, da.data_ptr()), (part.data_ptr() + 800, 37, 400, 200, db.data_ptr())]
        if sp.value > 0:
            for i, (a, b) in enumerate(probs):
                ent.append((so[i], sp.value, a * b, a * b, Cs[i].data_ptr()))
                if gbs[i] is not None:
                    assert bso[i]
                    ent.append((bso[i], sp.value, a, a, gbs[i].data_ptr()))
                else:
                    assert not bso[i]
        else:
            assert any(a % 256 or b % 256 for a, b in probs)                              # only the fallback case leaves nothing to sum
        ne = len(ent)
        LB.check(lib.ltrx_reduce_group(ne, (ctypes.c_void_p * ne)(*[e[0] for e in ent]), (ctypes.c_int * ne)(*[e[1] for e in ent]),
                                       (ctypes.c_size_t * ne)(*[e[2] for e in ent]), (ctypes.c_size_t * ne)(*[e[3] for e in ent]),
                                       (ctypes.c_void_p * ne)(*[e[4] for e in ent]), None), "reduce_group")
        for i in range(n):
            assert torch.equal(Cs[i], runs[0][0][i]), ("deferred", Mm, probs[i])
            if gbs[i] is not None:
                assert torch.equal(gbs[i], runs[0][1][i])
        p64 = part.cpu().numpy().astype(np.float64)
        assert np.abs(da.cpu().numpy() - p64[:, :200].sum(0)).max() < 1e-5 and np.abs(db.cpu().numpy() - p64[:, 200:].sum(0)).max() < 1e-5
        for i, (a, b) in enumerate(probs):
            A64 = As[i][:, :a].astype(np.float64)
            ref = A64.T @ Bs[i].astype(np.float64)
            scale = (np.abs(A64).T @ np.abs(Bs[i]).astype(np.float64)).max()
            got = runs[0][0][i]
            assert float(np.abs(got.cpu().numpy() - ref).max() / scale) < 4e-6, (Mm, a, b)
            assert torch.equal(got, runs[1][0][i]), ("run-to-run", Mm, a, b)
            if runs[0][1][i] is not None:
                bref = A64.sum(0)
                assert float(np.abs(runs[0][1][i].cpu().numpy() - bref).max()) < 1e-5 * max(1.0, np.abs(A64).sum(0).max())
                assert torch.equal(runs[0][1][i], runs[1][1][i])
    # argument validation: more problems than the table holds, a workspace too small for even one problem
    NP5, KP5 = (ctypes.c_int * 5)(*[256] * 5), (ctypes.c_int * 5)(*[256] * 5)
    assert lib.ltrx_gemm_tn_group_workspace_bytes(5, 1024, NP5, KP5) == 0
    vp1 = ctypes.c_void_p * 1
    a1, b1, c1 = torch.zeros((1024, 256), device=DEV), torch.zeros((1024, 256), device=DEV), torch.zeros((256, 256), device=DEV)
    one = (ctypes.c_int * 1)(256)
    assert lib.ltrx_gemm_tn_group(1, vp1(a1.data_ptr()), one, vp1(b1.data_ptr()), one, vp1(c1.data_ptr()), vp1(None), 1024, one, one, 0,
                                  LB.ptr(ws), 16, None, None, None, None) != 0
    assert lib.ltrx_reduce_group(17, None, None, None, None, None, None) != 0
    assert lib.ltrx_reduce_group(0, None, None, None, None, None, None) == 0


def test_grouped_wgrad_step_equals_the_per_projection_step():
    """FusedTrainer(group_wgrad=True) (default) vs group_wgrad=False: same loss bit for bit (the forward is untouched), every
    gradient equal to the per-projection launch's to split-order round-off, with and without sublayer dropout (the flush points
    differ: the dropout buffer is reused inside a layer)."""
    import copy
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    rng = np.random.default_rng(5)
    B, L, F = 16, 64, 136
    x = _t(rng.standard_normal((B, L, F)).astype(np.float32))
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[2, 40:] = -1
    yt = _t(y)
    for pdrop in (0.0, 0.1):
        torch.manual_seed(3)
        base = make_model(dict(sizes=[256], input_norm=False, activation=None, dropout=0.0),
                          dict(N=2, d_ff=512, h=2, positional_encoding=None, dropout=pdrop),
                          dict(d_output=1, output_activation=None), F).to(DEV)
        grads = {}
        for grouped in (True, False):
            m = copy.deepcopy(base)
            ft = FusedTrainer(m, "listNet", {}, B, L, lr=1e-3, use_graph=False, seed=11, group_wgrad=grouped)
            assert ft.group_wgrad == grouped
            loss = ft.step(x, yt).item()
            grads[grouped] = (loss, {k: p.grad.detach().clone() for k, p in m.named_parameters()})
            assert not ft._wg_pending
        assert grads[True][0] == grads[False][0]
        # (a tensor whose gradient is pure cancellation noise -- the final norm's b_2 under listNet, whose d loss / d scores sum to 0
        #  per slate -- is compared on the scale of the step's gradients, not on its own)
        floor = 1e-2 * max(float(r.abs().max()) for r in grads[False][1].values())
        for k, g in grads[True][1].items():
            ref = grads[False][1][k]
            tol = 2e-5 * max(floor, float(ref.abs().max()))
            assert float((g - ref).abs().max()) <= tol, (pdrop, k, float((g - ref).abs().max()), tol)


def test_one_launch_weight_refresh_and_batch_ingest_are_bit_identical_to_the_separate_launches():
    """ltrx_weight_images == ltrx_transpose_batch + ltrx_split_image x 2 (bits); ltrx_ingest_batch == copy + (y == pad) (bits),
    with and without x, vector and scalar x paths."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(31)
    mats = [(512, 136), (1536, 512), (96, 40), (2048, 512)]                      # [rows, cols] inside the flat buffer
    offs, o = [], 0
    for r, c in mats:
        offs.append(o)
        o += (r * c + 3) // 4 * 4
    nflat = o + 8
    flat = _t(rng.standard_normal(nflat).astype(np.float32))
    desc, tstart, od = [], [0], 0
    for (r, c), so in zip(mats, offs):
        desc += [so, od, r, c]
        tstart.append(tstart[-1] + ((r + 31) // 32) * ((c + 31) // 32))
        od += (r * c + 3) // 4 * 4
    tdesc = torch.tensor(desc, dtype=torch.int64, device=DEV)
    tst = torch.tensor(tstart, dtype=torch.int32, device=DEV)
    ft_a, ft_b = torch.zeros(od, device=DEV), torch.zeros(od, device=DEV)
    ia_p, ib_p = torch.zeros(nflat, device=DEV), torch.zeros(nflat, device=DEV)
    ia_t, ib_t = torch.zeros(od, device=DEV), torch.zeros(od, device=DEV)
    LB.check(lib.ltrx_transpose_batch(LB.ptr(flat), LB.ptr(ft_a), LB.ptr(tdesc), LB.ptr(tst), len(mats), tstart[-1], None), "transpose_batch")
    LB.check(lib.ltrx_split_image(LB.ptr(ft_a), LB.ptr(ia_t), od, None), "split_image")
    LB.check(lib.ltrx_split_image(LB.ptr(flat), LB.ptr(ia_p), nflat, None), "split_image")
    w0 = flat[offs[0]:offs[0] + 512 * 136].view(512, 136)                       # + the row-padded copy of one matrix: [512][136] -> [512][160]
    wp, wpi = torch.zeros((512, 160), device=DEV), torch.zeros((512, 160), device=DEV)
    LB.check(lib.ltrx_weight_images(LB.ptr(flat), nflat, LB.ptr(ib_p), LB.ptr(ft_b), LB.ptr(ib_t), LB.ptr(tdesc), LB.ptr(tst), len(mats),
                                    tstart[-1], LB.ptr(w0), 512, 136, 160, LB.ptr(wp), LB.ptr(wpi), None), "weight_images")
    ref_p = torch.zeros((512, 160), device=DEV)
    ref_p[:, :136] = w0
    ref_pi = torch.zeros((512, 160), device=DEV)
    LB.check(lib.ltrx_split_image(LB.ptr(ref_p), LB.ptr(ref_pi), ref_p.numel(), None), "split_image")
    assert torch.equal(wp, ref_p) and torch.equal(wpi.view(torch.int32), ref_pi.view(torch.int32))
    assert torch.equal(ft_a, ft_b)
    assert torch.equal(ia_p.view(torch.int32), ib_p.view(torch.int32))
    assert torch.equal(ia_t.view(torch.int32), ib_t.view(torch.int32))
    for (r, c), so, k in zip(mats, offs, range(len(mats))):
        assert torch.equal(ft_b[desc[4 * k + 1]:desc[4 * k + 1] + r * c].view(c, r), flat[so:so + r * c].view(r, c).t())
    ib_p.zero_()
    LB.check(lib.ltrx_weight_images(LB.ptr(flat), nflat, LB.ptr(ib_p), None, None, None, None, 0, 0, None, 0, 0, 0, None, None, None),
             "weight_images(no transposes)")
    assert torch.equal(ia_p.view(torch.int32), ib_p.view(torch.int32))
    assert lib.ltrx_weight_images(LB.ptr(flat), nflat + 1, LB.ptr(ib_p), None, None, None, None, 0, 0, None, 0, 0, 0, None, None, None) != 0
    assert lib.ltrx_weight_images(LB.ptr(flat), nflat, LB.ptr(ib_p), None, None, None, None, 0, 0, LB.ptr(w0), 512, 136, 130, LB.ptr(wp),
                                  LB.ptr(wpi), None) != 0                                      # ld < cols
    for (B, L, F) in [(8, 240, 136), (3, 7, 5)]:
        x = _t(rng.standard_normal((B, L, F)).astype(np.float32))
        yv = rng.integers(0, 5, (B, L)).astype(np.float32)
        yv[1, L // 2:] = -1
        y = _t(yv)
        xd, yd = torch.full((B * L, F), 7.0, device=DEV), torch.full((B, L), 7.0, device=DEV)
        md = torch.full((B, L), 9, dtype=torch.uint8, device=DEV)
        LB.check(lib.ltrx_ingest_batch(LB.ptr(x), LB.ptr(y), x.numel(), y.numel(), F, F, -1.0, LB.ptr(xd), LB.ptr(yd), LB.ptr(md), None), "ingest")
        assert torch.equal(xd.view(-1), x.view(-1)) and torch.equal(yd, y) and torch.equal(md.bool(), y == -1)
        ld = (F + 255) // 256 * 256 if F % 4 == 0 else F + 3                    # rows into padded rows: the padding is not touched
        xp = torch.full((B * L, ld), 7.0, device=DEV)
        LB.check(lib.ltrx_ingest_batch(LB.ptr(x), LB.ptr(y), x.numel(), y.numel(), F, ld, -1.0, LB.ptr(xp), LB.ptr(yd), LB.ptr(md), None),
                 "ingest(padded rows)")
        assert torch.equal(xp[:, :F], x.view(-1, F)) and bool((xp[:, F:] == 7.0).all())
        md.fill_(9)
        yd.fill_(7.0)
        LB.check(lib.ltrx_ingest_batch(None, LB.ptr(y), 0, y.numel(), 0, 0, -1.0, None, LB.ptr(yd), LB.ptr(md), None), "ingest(y only)")
        assert torch.equal(yd, y) and torch.equal(md.bool(), y == -1)
    assert lib.ltrx_ingest_batch(None, LB.ptr(y), 5, y.numel(), 5, 5, -1.0, None, LB.ptr(yd), LB.ptr(md), None) != 0
    assert lib.ltrx_ingest_batch(LB.ptr(x), LB.ptr(y), x.numel(), y.numel(), F, F - 1, -1.0, LB.ptr(xd), LB.ptr(yd), LB.ptr(md), None) != 0


def test_one_bit_relu_mask_gemm_epilogues_equal_the_fp32_activation_forms():
    """ltrx_gemm_nt act 4 (ReLU + mask bits out) == act 1 and act 5 (mask bits in) == act 2, bit for bit, with and without dropout,
    exact and ragged row counts; shapes the large-tile kernel does not take unconditionally report 0 bytes / LTRX_EUNSUPPORTED."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(41)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for (Mm, N, K1, p) in [(15360, 2048, 512, 0.0), (12300, 2048, 512, 0.1), (16384, 1024, 256, 0.3)]:
        nbytes = lib.ltrx_gemm_nt_relu_bits_bytes(Mm, N, K1)
        assert nbytes == ((Mm + 255) // 256) * (N // 256) * 8192
        x = _t(rng.standard_normal((Mm, K1)).astype(np.float32))
        w1 = _t((rng.standard_normal((N, K1)) / np.sqrt(K1)).astype(np.float32))
        b1 = _t(rng.standard_normal(N).astype(np.float32) * 0.1)
        dy = _t(rng.standard_normal((Mm, K1)).astype(np.float32))             # gradient w.r.t. the NEXT layer's output [M, K1]
        w2t = _t((rng.standard_normal((N, K1)) / np.sqrt(K1)).astype(np.float32))   # W2^T: [N, K1] (W2 is [K1, N])
        bits = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        r_ref, r_bit = torch.empty((Mm, N), device=DEV), torch.empty((Mm, N), device=DEV)
        LB.check(lib.ltrx_gemm_nt(LB.ptr(x), K1, LB.ptr(w1), K1, None, LB.ptr(r_ref), N, Mm, N, K1, LB.ptr(b1), 1, None, 0, p, 77, LB.ptr(step),
                                  0, 0, None), "act 1")
        LB.check(lib.ltrx_gemm_nt(LB.ptr(x), K1, LB.ptr(w1), K1, None, LB.ptr(r_bit), N, Mm, N, K1, LB.ptr(b1), 4, LB.ptr(bits), 0, p, 77,
                                  LB.ptr(step), 0, 0, None), "act 4")
        assert torch.equal(r_ref, r_bit), (Mm, N, p)
        assert 0.2 < float((r_ref > 0).float().mean()) < 0.6
        g_ref, g_bit = torch.empty((Mm, N), device=DEV), torch.empty((Mm, N), device=DEV)
        LB.check(lib.ltrx_gemm_nt(LB.ptr(dy), K1, LB.ptr(w2t), K1, None, LB.ptr(g_ref), N, Mm, N, K1, None, 2, LB.ptr(r_ref), N, p, 0, LB.ptr(step),
                                  0, 0, None), "act 2")
        LB.check(lib.ltrx_gemm_nt(LB.ptr(dy), K1, LB.ptr(w2t), K1, None, LB.ptr(g_bit), N, Mm, N, K1, None, 5, LB.ptr(bits), 0, p, 0, LB.ptr(step),
                                  0, 0, None), "act 5")
        assert torch.equal(g_ref, g_bit), (Mm, N, p)
    for (Mm, N) in [(12000, 2048), (1000, 2048), (15360, 2000), (15360, 512)]:      # 376 tiles (split dispatch), 32 tiles, N % 256, 120 tiles
        assert lib.ltrx_gemm_nt_relu_bits_bytes(Mm, N, 512) == 0
    assert lib.ltrx_gemm_nt_relu_bits_bytes(15360, 1024, 144) == 0                 # K % 32 != 0 (d_model 144)
    a, w = torch.zeros((1000, 512), device=DEV), torch.zeros((2048, 512), device=DEV)
    c, bits = torch.zeros((1000, 2048), device=DEV), torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    assert lib.ltrx_gemm_nt(LB.ptr(a), 512, LB.ptr(w), 512, None, LB.ptr(c), 2048, 1000, 2048, 512, None, 4, LB.ptr(bits), 0, 0.0, 0, None, 0, 0, None) != 0
    assert lib.ltrx_gemm_nt(LB.ptr(a), 512, LB.ptr(w), 512, None, LB.ptr(c), 2048, 1000, 2048, 512, None, 6, LB.ptr(bits), 0, 0.0, 0, None, 0, 0, None) != 0


def test_relu_bits_step_is_bit_identical_to_the_activation_reading_step():
    """FusedTrainer(relu_bits=True) (default) vs relu_bits=False at a shape where the mask form applies (64 slates x 240, d_ff 2048),
    with feed-forward dropout: same loss and same gradients, bit for bit, eager and captured."""
    import copy
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    rng = np.random.default_rng(6)
    B, L, F = 64, 240, 40
    x = _t(rng.standard_normal((B, L, F)).astype(np.float32))
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[5, 200:] = -1
    yt = _t(y)
    torch.manual_seed(4)
    base = make_model(dict(sizes=[256], input_norm=False, activation=None, dropout=0.0),
                      dict(N=1, d_ff=2048, h=4, positional_encoding=None, dropout=0.2),
                      dict(d_output=1, output_activation=None), F).to(DEV)
    out = {}
    for bits in (True, False):
        m = copy.deepcopy(base)
        ft = FusedTrainer(m, "listNet", {}, B, L, lr=1e-3, use_graph=True, seed=9, relu_bits=bits)
        assert (ft._relu_bits(ft.layers[0]) is not None) == bits
        losses = [ft.step(x, yt).item() for _ in range(4)]                   # 2 eager + capture + replay
        out[bits] = (losses, {k: p.detach().clone() for k, p in m.named_parameters()})
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    for k, w in out[True][1].items():
        assert torch.equal(w, out[False][1][k]), k


def test_relu_bits_default_falls_back_when_d_model_is_no_multiple_of_32():
    """d_model = 144, d_ff = 1024, 64 x 240 rows -> 240 large tiles, inside the mask form's tile range, but the
    GEMMs contract over K = 144 (K % 32 != 0): the default relu_bits=True must select acts 1 / 2 (no mask buffer) and train, with the
    same losses as relu_bits=False."""
    import copy
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    rng = np.random.default_rng(16)
    B, L, F = 64, 240, 24
    x = _t(rng.standard_normal((B, L, F)).astype(np.float32))
    yt = _t(rng.integers(0, 5, (B, L)).astype(np.float32))
    torch.manual_seed(5)
    base = make_model(dict(sizes=[144], input_norm=False, activation=None, dropout=0.0),
                      dict(N=1, d_ff=1024, h=4, positional_encoding=None, dropout=0.0),
                      dict(d_output=1, output_activation=None), F).to(DEV)
    res = {}
    for bits in (True, False):
        ft = FusedTrainer(copy.deepcopy(base), "listNet", {}, B, L, lr=1e-3, use_graph=False, seed=3, relu_bits=bits)
        assert ft._relu_bits(ft.layers[0]) is None and "rbits" not in ft.layers[0]
        res[bits] = [ft.step(x, yt).item() for _ in range(3)]
    assert res[True] == res[False] and all(np.isfinite(res[True]))


def _image_of(t):
    from allrank_amd import _lib as LB
    img = torch.empty_like(t)
    LB.check(LB.lib().ltrx_split_image(LB.ptr(t), LB.ptr(img), t.numel(), None), "split_image")
    return img


def test_activation_operand_images_equal_the_fp32_hand_over_bit_for_bit():
    """round 5: an activation written as a pre-split bf16 hi / lo image by its producer and staged by its consumers with plain copies.
    Kernel level: ltrx_layernorm_fwd_image == split_image(ltrx_layernorm_fwd); ltrx_gemm_nt_img with LTRX_GEMM_A_IS_IMAGE == the fp32
    operand, with LTRX_GEMM_C_AS_IMAGE == split_image(C) (through bias / ReLU-mask / residual / dropout epilogues, the 256-, 128- and
    64-row tile forms, ragged row counts); ltrx_gemm_tn_group_img with image B operands == ltrx_gemm_tn_group; the predicate refuses
    shapes of the small-tile kernels and the call returns LTRX_EUNSUPPORTED there; the image decodes back to the value (hi + lo, to 2^-17 relative)."""
    import ctypes
    from allrank_amd import _lib as LB
    from allrank_amd.engine import FusedTrainer
    lib = LB.lib()
    rng = np.random.default_rng(81)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    # LayerNorm forward
    for (rows, D) in [(3000, 512), (777, 256), (640, 1024)]:
        x, a, b = _t(rng.standard_normal((rows, D)).astype(np.float32)), _t(rng.standard_normal(D).astype(np.float32)), _t(rng.standard_normal(D).astype(np.float32))
        y, yi = torch.empty_like(x), torch.empty_like(x)
        m1, r1, m2, r2 = (torch.empty(rows, device=DEV) for _ in range(4))
        LB.check(lib.ltrx_layernorm_fwd(LB.ptr(x), None, LB.ptr(a), LB.ptr(b), rows, D, 1e-6, None, LB.ptr(y), LB.ptr(m1), LB.ptr(r1), 0.0, 0, None, None), "ln")
        LB.check(lib.ltrx_layernorm_fwd_image(LB.ptr(x), None, LB.ptr(a), LB.ptr(b), rows, D, 1e-6, None, LB.ptr(yi), LB.ptr(m2), LB.ptr(r2), 0.0, 0, None, None), "ln image")
        assert torch.equal(yi.view(torch.int32), _image_of(y).view(torch.int32)) and torch.equal(m1, m2) and torch.equal(r1, r2)
        g = yi.contiguous().view(torch.int32).view(-1, 4)                          # hi + lo of every element, the image decoded
        lo16, hi16 = (lambda x_: (x_ << 16).view(torch.float32)), (lambda x_: (x_ & -65536).view(torch.float32))
        dec = torch.stack([lo16(g[:, 0]) + lo16(g[:, 2]), hi16(g[:, 0]) + hi16(g[:, 2]),
                           lo16(g[:, 1]) + lo16(g[:, 3]), hi16(g[:, 1]) + hi16(g[:, 3])], 1).view(yi.shape)
        assert float((dec - y).abs().max()) <= 2.0 ** -16 * float(y.abs().max()) and torch.equal(dec > 0, y > 0)
    assert lib.ltrx_layernorm_fwd_image(LB.ptr(x[:, :300].contiguous()), None, LB.ptr(a), LB.ptr(b), 10, 300, 1e-6, None, LB.ptr(yi), LB.ptr(m2), LB.ptr(r2), 0.0, 0, None, None) == -2
    # NT GEMM: (M, N, K) -> 256-row tiles (exact / ragged), 128-row, 64-row, the split dispatch (one round + the rest)
    for (Mm, N, K) in [(15360, 2048, 512), (12300, 2048, 256), (15360, 512, 2048), (7680, 512, 512), (15360, 1536, 512), (61440, 512, 64)]:
        assert lib.ltrx_gemm_nt_image_ok(Mm, N, K) == 1, (Mm, N, K)
        A = _t(rng.standard_normal((Mm, K)).astype(np.float32))
        Bw = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        bias = _t(rng.standard_normal(N).astype(np.float32))
        aux = _t(rng.standard_normal((Mm, N)).astype(np.float32))
        Ai, Bi = _image_of(A), _image_of(Bw)
        nb = lib.ltrx_gemm_nt_relu_bits_bytes(Mm, N, K)
        for act, p in [(0, 0.0), (1, 0.2), (3, 0.1)] + ([(4, 0.0), (4, 0.3)] if nb else []):
            outs = []
            for fl in (0, 1, 2, 3):
                C = torch.zeros((Mm, N), device=DEV)
                bits = torch.zeros(max(nb, 16), dtype=torch.uint8, device=DEV)
                ax = bits if act == 4 else (aux if act == 3 else None)
                rc = lib.ltrx_gemm_nt_img(LB.ptr(Ai if fl & 1 else A), K, LB.ptr(Bw), K, LB.ptr(Bi), LB.ptr(C), N, Mm, N, K, LB.ptr(bias), act,
                                          LB.ptr(ax), N if act == 3 else 0, p, 99, LB.ptr(step), 0, 0, fl, None)
                assert rc == 0, (Mm, N, K, act, fl, rc)
                outs.append((C, bits))
            ref = outs[0][0]
            assert torch.equal(outs[1][0], ref), (Mm, N, K, act, p, "A image")
            assert torch.equal(outs[2][0].view(torch.int32), _image_of(ref).view(torch.int32)), (Mm, N, K, act, p, "C image")
            assert torch.equal(outs[3][0].view(torch.int32), _image_of(ref).view(torch.int32)), (Mm, N, K, act, p, "A and C image")
            assert all(torch.equal(o[1], outs[0][1]) for o in outs), (Mm, N, K, act, p, "mask bits")
    for (Mm, N, K) in [(1000, 2048, 512), (15360, 2000, 512), (15360, 512, 144), (70000, 1024, 512)]:      # small-tile kernel territory
        if lib.ltrx_gemm_nt_image_ok(Mm, N, K):
            continue
        A, Bw, C = torch.zeros((Mm, K), device=DEV), torch.zeros((N, K), device=DEV), torch.zeros((Mm, N), device=DEV)
        assert lib.ltrx_gemm_nt_img(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(Bw), LB.ptr(C), N, Mm, N, K, None, 0, None, 0, 0.0, 0, None, 0, 0, 1, None) != 0
    assert lib.ltrx_gemm_nt_image_ok(1000, 2048, 512) == 0 and lib.ltrx_gemm_nt_image_ok(15360, 512, 144) == 0
    # grouped weight gradient with image B operands (the four projections of an encoder layer; 3 of the 4 inputs as images)
    Mm, d, dff = 15360, 512, 2048
    probs = [(d, dff, True), (dff, d, True), (d, d, False), (3 * d, d, True)]       # (NP, KP, B is an image)
    dys = [_t(rng.standard_normal((Mm, n_)).astype(np.float32)) for n_, _, _ in probs]
    xs = [_t(rng.standard_normal((Mm, k_)).astype(np.float32)) for _, k_, _ in probs]
    xi = [_image_of(x_) if im else x_ for x_, (_, _, im) in zip(xs, probs)]
    n = len(probs)
    vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
    NP, KP = ci(*[p_[0] for p_ in probs]), ci(*[p_[1] for p_ in probs])
    wsb = lib.ltrx_gemm_tn_group_workspace_bytes(n, Mm, NP, KP)
    res = []
    for use_img in (False, True):
        ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
        Cs = [torch.zeros((n_, k_), device=DEV) for n_, k_, _ in probs]
        bs = [torch.zeros(n_, device=DEV) for n_, _, _ in probs]
        Bops = xi if use_img else xs
        rc = lib.ltrx_gemm_tn_group_img(n, vp(*[t.data_ptr() for t in dys]), ci(*[t.stride(0) for t in dys]), vp(*[t.data_ptr() for t in Bops]),
                                        ci(*[t.stride(0) for t in Bops]), vp(*[t.data_ptr() for t in Cs]), vp(*[t.data_ptr() for t in bs]), Mm, NP, KP, 0,
                                        LB.ptr(ws), wsb, None, None, None, ci(*[1 if (use_img and p_[2]) else 0 for p_ in probs]), None)
        assert rc == 0, rc
        res.append((Cs, bs))
    for a_, b_ in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert torch.equal(a_, b_)
    ref = dys[0].double().t() @ xs[0].double()
    assert float((res[1][0][0].double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_64_row_tile_gemm_equals_the_other_large_tile_forms_bit_for_bit():
    """ltrx_gemm_nt tile 8 (64 x 256 tiles, two workgroups per CU: the automatic choice for small batches) == tiles 7 and 6, bits,
    through every epilogue (bias, ReLU, ReLU mask, residual, dropout), exact and ragged row counts, with and without the image."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(51)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    for (Mm, N, K) in [(7680, 512, 2048), (3000, 512, 160), (6176, 256, 512)]:
        A = _t(rng.standard_normal((Mm, K)).astype(np.float32))
        Bw = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
        bias = _t(rng.standard_normal(N).astype(np.float32))
        aux = _t(rng.standard_normal((Mm, N)).astype(np.float32))
        img = torch.empty_like(Bw)
        LB.check(lib.ltrx_split_image(LB.ptr(Bw), LB.ptr(img), Bw.numel(), None), "split_image")
        for (act, p, use_img) in [(0, 0.0, True), (1, 0.2, True), (2, 0.1, False), (3, 0.0, True), (3, 0.3, False)]:
            outs = {}
            for v in (8, 7, 6):
                C = torch.full((Mm, N), float("nan"), device=DEV)
                LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(Bw), K, LB.ptr(img) if use_img else None, LB.ptr(C), N, Mm, N, K, LB.ptr(bias), act,
                                          LB.ptr(aux) if act >= 2 else None, N if act >= 2 else 0, p, 5, LB.ptr(step), 0, v, None), "gemm_nt tile %d" % v)
                outs[v] = C
            assert torch.equal(outs[8], outs[7]) and torch.equal(outs[8], outs[6]), (Mm, N, K, act, p)
    C = torch.empty((7680, 512), device=DEV)                                  # the automatic choice at 32 slates x 240 is this tile
    auto = torch.empty((7680, 512), device=DEV)
    A = _t(rng.standard_normal((7680, 2048)).astype(np.float32))
    Bw = _t((rng.standard_normal((512, 2048)) / 45.0).astype(np.float32))
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), 2048, LB.ptr(Bw), 2048, None, LB.ptr(C), 512, 7680, 512, 2048, None, 0, None, 0, 0.0, 0, None, 0, 8, None), "t8")
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), 2048, LB.ptr(Bw), 2048, None, LB.ptr(auto), 512, 7680, 512, 2048, None, 0, None, 0, 0.0, 0, None, 0, 0, None), "t0")
    assert torch.equal(C, auto)
    ref = A.double() @ Bw.double().t()
    assert float((auto.double() - ref).abs().max() / (A.double().abs() @ Bw.double().abs().t()).max()) < 4e-6


def test_row4_losses_edge_shapes():
    """single-item slates, a fully padded slate, the maximum slate length and a single slate: engine == oracle (NaN where
    the reference's own arithmetic is 0/0)."""
    rng = np.random.default_rng(77)
    for (B, L) in [(4, 1), (1, 2048), (3, 5), (1, 1)]:
        s = rng.standard_normal((B, L)).astype(np.float32)
        y = rng.integers(0, 3, (B, L)).astype(np.float32)
        if B >= 3:
            y[1, :] = -1                                       # a fully padded slate
            y[2, L // 2:] = -1
        p = (1 / (1 + np.exp(-s))).astype(np.float32)
        p3 = (1 / (1 + np.exp(-rng.standard_normal((B, L, 2))))).astype(np.float32)
        yb = np.where(y == -1, -1, (y >= 1)).astype(np.float32)
        for kind, kw, yp, yt in [("ranknet", {}, s, y), ("ranknet", dict(weight_by_diff=True), s, y), ("bce", {}, p, yb),
                                 ("ordinal", dict(n=2), p3, y), ("pointwise_rmse", dict(no_of_levels=2), p, y),
                                 ("binary_listnet", {}, s, yb)]:
            if kind == "binary_listnet" and B >= 3:
                # documented deviation: a fully padded slate contributes 0 where the reference's softmax
                # over an empty set is NaN -- compare without that slate (mean over 3 slates -> rescale)
                l, g = _extra_engine(kind, kw, yp, yt)
                keep = [b for b in range(B) if np.any(yt[b] != -1)]
                with np.errstate(all="ignore"):
                    lo, go2 = _EXTRA_ORACLE[kind](yp[keep], yt[keep], **kw)
                assert close(l, lo * len(keep) / B), (B, L, kind, l, lo)
                assert np.allclose(g[keep], go2 * len(keep) / B, atol=1e-6) and not np.any(g[1] != 0)
                continue
            l, g = _extra_engine(kind, kw, yp, yt)
            with np.errstate(all="ignore"):
                lo, go = _EXTRA_ORACLE[kind](yp, yt, **kw)
            assert close(l, lo), (B, L, kind, l, lo)
            both_nan = np.isnan(g) & np.isnan(go)
            assert np.all(both_nan | (np.abs(g - go) <= 2e-4 * max(float(np.nanmax(np.abs(go))) if np.isfinite(go).any() else 1.0, 1e-6) + 1e-7)), (B, L, kind)
    from allrank_amd import metrics as EM
    s = rng.standard_normal((2, 2048)).astype(np.float32)
    y = rng.integers(0, 5, (2, 2048)).astype(np.float32)
    assert np.array_equal(EM.mrr(_t(s), _t(y), ats=[1, 10, 2048]).cpu().numpy(), O.mrr(s, y, [1, 10, 2048]))


@pytest.mark.parametrize("D", [512, 256, 96])
def test_score_head_kernels_match_torch(D):
    """OutputLayer with d_output == 1 (model.py:111-117): forward scores and backward (dx, dw, db) of the explicit step's head
    kernels (vectorised for D % 256 == 0, generic otherwise) against torch."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    rng = np.random.default_rng(D)
    Mm = 3000
    x = _t(rng.standard_normal((Mm, D)).astype(np.float32))
    w = _t(rng.standard_normal((1, D)).astype(np.float32))
    b = _t(rng.standard_normal(1).astype(np.float32))
    ds = _t(rng.standard_normal(Mm).astype(np.float32))
    sc = torch.empty(Mm, device=DEV)
    LB.check(lib.ltrx_score_head_fwd(LB.ptr(x), LB.ptr(w), LB.ptr(b), Mm, D, LB.ptr(sc), None), "head_fwd")
    ref = (x.double() @ w.double().t()).squeeze(1) + b.double()
    assert float((sc.double() - ref).abs().max()) < 1e-4
    dx, dw, db = torch.empty_like(x), torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(max(lib.ltrx_score_head_bwd_workspace_bytes(Mm, D), 64), dtype=torch.uint8, device=DEV)
    LB.check(lib.ltrx_score_head_bwd(LB.ptr(ds), LB.ptr(x), LB.ptr(w), Mm, D, LB.ptr(dx), LB.ptr(dw), LB.ptr(db), LB.ptr(ws), None), "head_bwd")
    assert torch.equal(dx, ds[:, None] * w)
    assert float((dw.double() - ds.double()[None, :] @ x.double()).abs().max()) < 2e-3
    assert abs(float(db.item()) - float(ds.double().sum())) < 1e-3


def test_fused_trainer_gradient_clipping_matches_clip_grad_norm():
    """train_utils.py:24-25: clip_grad_norm_ before the optimizer step -- the explicit step keeps the coefficient on the
    device (graph-capturable) and must follow the autograd path step for step."""
    import copy
    from allrank_amd import losses as E
    from allrank_amd.engine import FusedTrainer, Trainer
    torch.manual_seed(11)
    m1 = _dropout_model(0.0, None, 0.0, N=1)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(12)
    B, L = 6, 30
    x = _t(rng.standard_normal((B, L, 20)).astype(np.float32))
    y = _t(rng.integers(0, 5, (B, L)).astype(np.float32))
    clip = 0.05
    ft = FusedTrainer(m1, "listNet", {}, B, L, lr=1e-3, use_graph=True, gemm="split_bf16_strict", gradient_clipping_norm=clip)
    tr = Trainer(m2, E.listNet, torch.optim.Adam(m2.parameters(), lr=1e-3), gradient_clipping_norm=clip)
    for step in range(5):
        lf, la = float(ft.step(x, y).item()), float(tr.step(x, y, None).item())
        assert abs(lf - la) <= (2e-5 if step == 0 else 1e-3) * (1 + abs(la)), (step, lf, la)
    assert float(ft.grad_norm.item()) > clip and float(ft.clip_scale.item()) < 1.0       # the clip was active
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert float((sd1[k] - sd2[k]).abs().max().item()) <= 1.01e-2, k


@pytest.mark.parametrize("L", [2048, 3000, 4096, 7000])
def test_listwise_losses_at_and_beyond_the_lds_slate_length(L):
    """The reference's losses take any slate length (validation sets are padded to their longest slate, dataset_loading.py:185-194;
    approxNDCG.py:7-53, listMLE.py:7-38, lambdaLoss.py:7-81, listNet.py:8-30).  L = 2048 is LTRX_MAX_SLATE_LEN: the per-slate LDS
    working sets (106 KB for lambdaLoss) and the partner-range split.  Beyond it the four hot losses keep their work arrays in LDS
    while they fit the CU's 160 KB and in the call's workspace after that: 3000 = lambdaLoss's last LDS length,
    4096 = lambdaLoss in the workspace, 7000 = approxNDCG and listMLE in the workspace too.  All against the oracle."""
    rng = np.random.default_rng(L)
    B = 2
    s = rng.standard_normal((B, L)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[1, (3 * L) // 4:] = -1
    perm = rng.permutation(L)
    jobs = [("lambdaloss", dict(weighing_scheme="lambdaRank_scheme", k=None)), ("lambdaloss", dict(weighing_scheme="ndcgLoss2PP_scheme", k=100)),
            ("approxndcg", dict(alpha=1.0)), ("listnet", {}), ("listmle", dict(perm=perm))]
    for kind, kw in jobs:
        l, g = _engine_loss(kind, kw, s, y)
        lo, go = _oracle_loss(kind, kw, s, y)
        assert close(l, lo, rtol=3e-5), (L, kind, l, lo)
        assert grad_close(g, go, rtol=5e-4), (L, kind, float(np.abs(g - go).max()))


def test_long_slate_workspace_form_equals_lds_form_inside_the_fused_loss():
    """FusedLoss (the explicit training step's loss launcher) sizes its workspace through *_workspace_bytes(B, L): at L = 4096 the
    lambdaLoss arrays are in that workspace; value and gradient equal the plugin call's"""
    from allrank_amd import losses as E
    rng = np.random.default_rng(11)
    B, L = 3, 4096
    s = rng.standard_normal((B, L)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    y[2, 100:] = -1
    for name, kw in (("lambdaLoss", dict(weighing_scheme="lambdaRank_scheme")), ("approxNDCGLoss", {}), ("listNet", {})):
        fl = E.FusedLoss(name, B, L, "cuda:0", **kw)
        loss, grad = fl.run(_t(s), _t(y), float(B))
        sp = _t(s, True)
        l2 = getattr(E, name)(sp, _t(y), **kw)
        l2.backward()
        assert torch.equal(loss.reshape(()), l2.detach().reshape(())) and torch.equal(grad, sp.grad), name


# ---------------------------------------------------------------------------------------------------------------------
# compacted (variable-length) execution: padded slots skipped, same results
# ---------------------------------------------------------------------------------------------------------------------
def test_gather_scatter_rows():
    from allrank_amd import _lib as LB
    lib = LB.lib()
    g = torch.Generator().manual_seed(3)
    src = torch.randn((50, 13), generator=g).cuda()
    idx = torch.tensor([4, 0, 49, 7, 7, 31], dtype=torch.int32).cuda()
    dst = torch.full((8, 16), 9.0).cuda()
    st = LB.stream_of(src)
    LB.check(lib.ltrx_gather_rows(LB.ptr(src), 13, LB.ptr(idx), 6, 8, 13, LB.ptr(dst), 16, st), "gather_rows")
    assert torch.equal(dst[:6, :13], src[idx.long()]) and (dst[6:8, :13] == 0).all() and (dst[:, 13:] == 9.0).all()
    back = torch.zeros((50, 13)).cuda()
    uniq = torch.tensor([4, 0, 49, 7, 31], dtype=torch.int32).cuda()
    LB.check(lib.ltrx_scatter_rows(LB.ptr(dst), 16, LB.ptr(uniq), 5, 13, LB.ptr(back), 13, st), "scatter_rows")
    assert torch.equal(back[uniq.long()], dst[:5, :13]) and int((back != 0).any(1).sum().item()) == 5


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
@pytest.mark.parametrize("B,L,h,dk,lens", [(3, 70, 4, 8, [70, 1, 33]), (4, 240, 8, 64, [240, 100, 129, 17]),
                                           (2, 300, 2, 32, [257, 300]), (3, 600, 2, 64, [600, 257, 31]), (2, 1024, 1, 64, [1024, 700])])
def test_attention_varlen_matches_padded(B, L, h, dk, lens, p_drop):
    """cu_seqlens layout (packed valid rows, no mask) == padded layout + key mask on the valid rows: forward output and
    all three input gradients, bit for bit (same tiles, same order), dropout included (the mask hash is keyed by the
    position inside the slate)."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    d = h * dk
    g = torch.Generator().manual_seed(B * 1000 + L)
    qkv = torch.randn((B, L, 3 * d), generator=g).cuda()
    do = torch.randn((B, L, d), generator=g).cuda()
    lens_t = torch.tensor(lens)
    mask = (torch.arange(L)[None, :] >= lens_t[:, None]).to(torch.uint8).cuda()
    do = do * (mask == 0)[:, :, None]          # padded rows carry no gradient in the step (the loss masks them)
    valid = (mask == 0).reshape(-1)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens_t, 0)
    n = int(cu[-1])
    cu = cu.cuda()
    st = LB.stream_of(qkv)
    ws = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, L, h, dk, 1), 64), dtype=torch.uint8, device="cuda")

    def run(qkv_, do_, kpm, cu_, rows, order_=None):
        o = torch.zeros((rows, d), device="cuda")
        lse = torch.zeros((B, h, L), device="cuda")
        dqkv = torch.zeros((rows, 3 * d), device="cuda")
        LB.check(lib.ltrx_mha_fwd(LB.ptr(qkv_), qkv_.data_ptr() + 4 * d, qkv_.data_ptr() + 8 * d, LB.ptr(kpm), B, L, h, dk, 3 * d,
                                  LB.ptr(o), d, LB.ptr(lse), p_drop, 77, None, LB.ptr(cu_), LB.ptr(order_), 1, st), "mha_fwd")
        LB.check(lib.ltrx_mha_bwd(LB.ptr(qkv_), qkv_.data_ptr() + 4 * d, qkv_.data_ptr() + 8 * d, LB.ptr(kpm), LB.ptr(o), LB.ptr(do_),
                                  LB.ptr(lse), B, L, h, dk, 3 * d, d, LB.ptr(dqkv), dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d,
                                  3 * d, p_drop, 77, None, LB.ptr(cu_), LB.ptr(order_), 1, LB.ptr(ws), st), "mha_bwd")
        return o, dqkv

    o_p, dqkv_p = run(qkv.reshape(B * L, 3 * d), do.reshape(B * L, d), mask, None, B * L)
    qkv_c = qkv.reshape(B * L, 3 * d)[valid].contiguous()
    do_c = do.reshape(B * L, d)[valid].contiguous()
    o_c, dqkv_c = run(qkv_c, do_c, None, cu, n)
    assert torch.equal(o_c, o_p[valid])
    assert torch.equal(dqkv_c, dqkv_p[valid])
    assert torch.isfinite(o_c).all() and torch.isfinite(dqkv_c).all()
    order = torch.argsort(lens_t, descending=True).to(torch.int32).cuda()        # launch order: results unchanged
    o_s, dqkv_s = run(qkv_c, do_c, None, cu, n, order)
    assert torch.equal(o_s, o_c) and torch.equal(dqkv_s, dqkv_c)
    o_s, dqkv_s = run(qkv.reshape(B * L, 3 * d), do.reshape(B * L, d), mask, None, B * L, order)
    assert torch.equal(o_s, o_p) and torch.equal(dqkv_s, dqkv_p)


@pytest.mark.parametrize("gemm", ["split_bf16", "hipblaslt"])
@pytest.mark.parametrize("loss_name,loss_args,host_lengths", [("approxNDCGLoss", {}, True), ("listNet", {}, False),
                                                               ("lambdaLoss", dict(weighing_scheme="ndcgLoss2_scheme"), True),
                                                               ("rankNet", {}, False)])
def test_fused_trainer_compact_matches_padded_and_oracle(loss_name, loss_args, host_lengths, gemm):
    """FusedTrainer(compact=True) (valid items packed, padded slots never computed) == the padded step == the oracle:
    loss of every step, all gradients of the first step, weights after 4 steps."""
    import copy
    from allrank_amd.engine import FusedTrainer
    cfg = dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None)
    params = M.init_params(cfg, seed=21)
    m1 = _make_engine_model(cfg, params)
    m2 = copy.deepcopy(m1)
    rng = np.random.default_rng(22)
    B, L = 5, 70
    lens = [70, 3, 41, 1, 64]                   # 179 valid items -> 192 packed rows (13 alignment rows)
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    y = rng.integers(0, 5, (B, L)).astype(np.float32)
    for b, n in enumerate(lens):
        y[b, n:] = -1
        x[b, n:] = 0
    if not host_lengths:                        # padding need not be at the end when the trainer counts on the device
        y[2, 5] = -1
        x[2, 5] = 0
    xt, yt = _t(x), _t(y)
    fc = FusedTrainer(m1, loss_name, loss_args, B, L, lr=1e-3, gemm=gemm, compact=True)
    fp = FusedTrainer(m2, loss_name, loss_args, B, L, lr=1e-3, gemm=gemm, use_graph=False)
    ofn = {"approxNDCGLoss": lambda s, t: O.approxndcg(s, t), "listNet": lambda s, t: O.listnet(s, t),
           "lambdaLoss": lambda s, t: O.lambdaloss(s, t, **loss_args), "rankNet": lambda s, t: O.ranknet(s, t)}[loss_name]
    oopt = M.Adam(params, lr=1e-3)
    rows = []
    for step in range(4):
        lc = float(fc.step(xt, yt, lengths=lens if host_lengths else None).item())
        lp = float(fp.step(xt, yt).item())
        lo = float(M.train_step(params, cfg, oopt, x, y, ofn)[0])
        rows.append((lc, lp, lo))
        tol = 1e-5 if step == 0 else 2e-3
        assert abs(lc - lp) <= tol * (1 + abs(lp)) and abs(lc - lo) <= tol * (1 + abs(lo)), rows
        if step == 0:
            assert fc.n_valid == int((y != -1).sum()) and fc.rows % 32 == 0 and fc.rows < B * L
            gscale = float(fp.flat_g.abs().max().item())
            gerr = float((fc.flat_g - fp.flat_g).abs().max().item())
            assert gerr <= 2e-5 * gscale + 1e-9, (gerr, gscale)
            v = (yt != -1)
            assert (fc.scores[v] - fp.scores[v]).abs().max().item() < 2e-5
    _log("fused_trainer_compact_%s_%s" % (loss_name, gemm), rows)
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    for k in sd1:
        assert (sd1[k] - sd2[k]).abs().max().item() <= 8.1e-3, k


def test_fused_trainer_compact_dropout_trains():
    """compact execution with the reference's dropout rates: finite, loss decreases on a fixed batch."""
    from allrank_amd.engine import FusedTrainer
    from allrank_amd.model import make_model
    torch.manual_seed(5)
    tr = dict(N=2, d_ff=64, h=4, positional_encoding=None, dropout=0.1)
    fc = dict(sizes=[32], input_norm=False, activation=None, dropout=0.0)
    model = make_model(fc, tr, dict(d_output=1, output_activation=None), 20).cuda()
    rng = np.random.default_rng(6)
    B, L = 8, 60
    x = rng.standard_normal((B, L, 20)).astype(np.float32)
    w = rng.standard_normal(20).astype(np.float32)
    y = np.clip(np.round((x @ w) * 0.6 + 2), 0, 4).astype(np.float32)
    lens = rng.integers(5, L + 1, B)
    for b, n in enumerate(lens):
        y[b, n:] = -1
        x[b, n:] = 0
    ft = FusedTrainer(model, "approxNDCGLoss", {}, B, L, lr=3e-3, compact=True, seed=9)
    ls = [float(ft.step(_t(x), _t(y), lengths=lens.tolist()).item()) for _ in range(60)]
    assert np.isfinite(ls).all() and np.mean(ls[-10:]) < np.mean(ls[:10]) - 0.02, (ls[:3], ls[-3:])


# ---------------------------------------------------------------------------------------------------------------------
# the whole training step at BASELINE.json's full size (config 3, 256 slates x 240 items): size-independent properties
# ---------------------------------------------------------------------------------------------------------------------
def test_full_size_step_properties():
    """The numpy oracle needs minutes at this size, so the full-size step is checked through properties the domain offers:
    (1) slates are independent -- the batch loss is the mean of the losses of its four 64-slate quarters and the gradient
    the mean of theirs; (2) shuffling the slates of the batch changes neither; (3) with no positional encoding the model is
    equivariant and ApproxNDCG invariant under a permutation of the items of each slate; (4) padded slots contribute
    nothing -- the variable-length step on a ragged version of the batch equals the padded step."""
    import copy
    import bench
    from allrank_amd.engine import FusedTrainer
    w = bench.WORKLOADS["attn_approxndcg"]
    B, L, F = 256, 240, w["n_features"]
    dev = torch.device(DEV)
    GT = 5e-4          # gradients: sums over 61440 rows of fp32-class products in a different order, relative to the largest entry
    x, y, _ = bench.synth_batch(B, L, F, 123, dev)
    model0 = bench.build_model(w, dev, 0.0)

    def first_step(xb, yb, nb, **kw):
        """loss, flat gradient and scores of ONE step from the common initial weights"""
        m = copy.deepcopy(model0)
        tr = FusedTrainer(m, w["loss"], {}, nb, L, lr=1e-3, use_graph=False, **kw)
        lengths = kw.pop("lengths", None)
        loss = tr.step(xb, yb) if not tr.compact else tr.step(xb, yb, lengths=(yb != -1).sum(1).cpu())
        out = (float(loss.item()), tr.flat_g.clone(), tr.scores.clone())
        del tr, m
        return out

    l_all, g_all, s_all = first_step(x, y, B)
    gscale = float(g_all.abs().max().item())
    assert np.isfinite(l_all) and gscale > 0
    # (1) quarters
    lq, gq = [], []
    for i in range(4):
        l_i, g_i, s_i = first_step(x[64 * i:64 * i + 64], y[64 * i:64 * i + 64], 64)
        lq.append(l_i)
        gq.append(g_i)
        assert (s_i - s_all[64 * i:64 * i + 64]).abs().max().item() < 2e-5
    assert abs(np.mean(lq) - l_all) <= 1e-5 * (1 + abs(l_all)), (lq, l_all)
    gmean = torch.stack(gq).mean(0)
    assert (gmean - g_all).abs().max().item() <= GT * gscale, ((gmean - g_all).abs().max().item(), gscale)
    # (2) slate shuffle
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(5)).to(dev)
    l_p, g_p, s_p = first_step(x[perm], y[perm], B)
    assert abs(l_p - l_all) <= 1e-5 * (1 + abs(l_all))
    assert (s_p - s_all[perm]).abs().max().item() < 2e-5
    assert (g_p - g_all).abs().max().item() <= GT * gscale
    # (3) item permutation inside every slate
    ip = torch.argsort(torch.rand((B, L), generator=torch.Generator().manual_seed(6)), dim=1).to(dev)
    xi = torch.gather(x, 1, ip[:, :, None].expand(B, L, F))
    yi = torch.gather(y, 1, ip)
    l_i, g_i, s_i = first_step(xi, yi, B)
    assert abs(l_i - l_all) <= 1e-5 * (1 + abs(l_all)), (l_i, l_all)
    # (within-slate order changes which keys share a tile and the summation order of the attention contractions; with the
    #  three-product bf16 arithmetic of the attention the two runs carry independent ~2e-5 errors on scores of magnitude ~6)
    assert (s_i - torch.gather(s_all, 1, ip)).abs().max().item() < 1e-4
    # (gradients: the two runs also take different ReLU masks on the handful of hidden units whose pre-activation is within the
    #  forward round-off of 0 -- see tests/test_gpu_benchdims.py -- each worth one row's contribution to the FFN gradients)
    assert (g_i - g_all).abs().max().item() <= 4 * GT * gscale
    # (4) ragged batch: variable-length execution == padded execution
    xr, yr, _ = bench.synth_batch(B, L, F, 124, dev, ragged=True)
    l_pad, g_pad, s_pad = first_step(xr, yr, B)
    l_c, g_c, s_c = first_step(xr, yr, B, compact=True)
    v = yr != -1
    assert abs(l_c - l_pad) <= 1e-5 * (1 + abs(l_pad)), (l_c, l_pad)
    assert (s_c[v] - s_pad[v]).abs().max().item() < 2e-5
    gs = float(g_pad.abs().max().item())
    assert (g_c - g_pad).abs().max().item() <= GT * gs
    _log("full_size_step_properties", dict(loss=l_all, quarter_losses=lq, shuffled=l_p, item_permuted=l_i, padded=l_pad, compact=l_c))
