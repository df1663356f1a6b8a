"""-m gpu: the drop-in epoch loop (allrank_amd.fit.fit, reference signature of train_utils.py:78-147) and the epoch metrics
pass (allrank_amd.data.evaluate vs compute_metrics, train_utils.py:47-56 as restated by the oracle)."""
import os
import types
from functools import partial

import numpy as np
import pytest
import torch

from oracle import ltr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _data(n, L, F, seed, ragged=True):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, L, F)).astype(np.float32)
    w = rng.standard_normal(F).astype(np.float32)
    y = np.clip(np.round((x @ w) / np.sqrt(F) + 1.5), 0, 4).astype(np.float32)      # learnable labels
    idx = np.tile(np.arange(L, dtype=np.int64), (n, 1))
    if ragged:
        for b in range(n):
            k = int(rng.integers(L // 3, L + 1))
            y[b, k:] = -1
            x[b, k:] = 0
            idx[b, k:] = -1
    return torch.tensor(x), torch.tensor(y), torch.tensor(idx)


def _model(F, dropout=0.0, pe=None):
    from allrank_amd.model import make_model
    torch.manual_seed(7)
    return make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                      dict(N=2, d_ff=64, h=4, positional_encoding=pe, dropout=dropout), dict(d_output=1, output_activation=None), F).to(DEV)


def _loaders(n_train=40, n_val=24, L=30, F=20, bs=16):
    from torch.utils.data import DataLoader, TensorDataset
    tr = TensorDataset(*_data(n_train, L, F, 1))
    va = TensorDataset(*_data(n_val, L, F, 2))
    return DataLoader(tr, batch_size=bs, shuffle=False), DataLoader(va, batch_size=bs, shuffle=False)


@pytest.mark.parametrize("compact", [False, True])
def test_fit_has_the_reference_contract_and_runs_the_fused_step(tmp_path, compact):
    from allrank_amd import losses as E, fit as EF
    cfg = types.SimpleNamespace(metrics={"ndcg": [5, 10]}, val_metric="ndcg_5")
    train_dl, val_dl = _loaders()
    model = _model(20, dropout=0.1, pe=dict(strategy="fixed", max_indices=24))
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    res = EF.fit(epochs=4, model=model, loss_func=partial(E.approxNDCGLoss, alpha=1.0), optimizer=opt, scheduler=sched, train_dl=train_dl,
                 valid_dl=val_dl, config=cfg, gradient_clipping_norm=1.0, early_stopping_patience=10, device=torch.device(DEV),
                 output_dir=str(tmp_path), tensorboard_output_path=None, compact=compact)
    assert EF.last_run["engine"] == "fused" and EF.last_run["compact"] == compact, EF.last_run
    assert set(res) == {"epochs", "train_metrics", "val_metrics", "num_params"} and res["epochs"] == 3
    assert set(res["val_metrics"]) == {"ndcg_5", "ndcg_10"} and set(res["train_metrics"]) == {"ndcg_5", "ndcg_10"}
    assert all(0.0 < float(v) <= 1.0 for v in list(res["val_metrics"].values()) + list(res["train_metrics"].values()))
    assert res["num_params"] == sum(p.numel() for p in model.parameters())
    assert abs(opt.param_groups[0]["lr"] - 2e-3 * 0.5 ** 4) < 1e-12          # the scheduler ran once per epoch
    sd = torch.load(os.path.join(str(tmp_path), "model.pkl"))
    fresh = _model(20, dropout=0.1, pe=dict(strategy="fixed", max_indices=24))
    assert not fresh.load_state_dict(sd, strict=True).missing_keys
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), sd[k].cpu()), k


def test_fit_fused_and_autograd_paths_agree_and_learn(tmp_path):
    from allrank_amd import losses as E, fit as EF
    cfg = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric="ndcg_5")
    out = {}
    for fused in (True, False):
        train_dl, val_dl = _loaders(n_train=48, bs=16)
        model = _model(20)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        hist = []
        # one epoch at a time so that the per-epoch numbers can be compared
        res = EF.fit(epochs=1, model=model, loss_func=partial(E.listNet), optimizer=opt, scheduler=None, train_dl=train_dl, valid_dl=val_dl,
                     config=cfg, gradient_clipping_norm=None, early_stopping_patience=5, device=torch.device(DEV),
                     output_dir=str(tmp_path), tensorboard_output_path=None, use_fused=fused, compact=False)
        assert EF.last_run["engine"] == ("fused" if fused else "autograd")
        out[fused] = (res["train_metrics"]["ndcg_5"], res["val_metrics"]["ndcg_5"])
    # same data order, same initial weights, no dropout: the two engines run the same three training steps
    assert abs(out[True][0] - out[False][0]) < 2e-3 and abs(out[True][1] - out[False][1]) < 5e-3, out
    # early stopping: patience 0 stops as soon as the validation metric fails to improve
    train_dl, val_dl = _loaders()
    model = _model(20)
    res = EF.fit(epochs=30, model=model, loss_func=partial(E.listNet), optimizer=torch.optim.Adam(model.parameters(), lr=0.05), scheduler=None,
                 train_dl=train_dl, valid_dl=val_dl, config=cfg, gradient_clipping_norm=None, early_stopping_patience=0,
                 device=torch.device(DEV), output_dir=str(tmp_path), tensorboard_output_path=None)
    assert res["epochs"] < 29


def test_fused_step_short_last_batch_is_normalised_by_its_real_slate_count():
    """an epoch of 16 / 16 / 8 slates (DataLoader drop_last=False, dataset_loading.py:245).  The short batch
    arrives topped up with fully padded slates and global_batch = 8; its loss, gradients and the weights after it must equal
    the autograd Trainer's on the real 8 slates -- also when it falls on the step at which the hipGraph would be captured
    (step 3) and when a graph captured on full batches is already live (step 6).  Both mean-normalised losses whose divisor
    is a launch scalar are covered."""
    import copy
    from allrank_amd import losses as E
    from allrank_amd.engine import FusedTrainer
    L, F, bs = 30, 20, 16
    x, y, idx = (t.to(DEV) for t in _data(40, L, F, 5))
    for loss_name in ("listNet", "approxNDCGLoss"):
        m_f = _model(F)
        m_a = copy.deepcopy(m_f)                     # autograd reference: the reference's own loss_batch arithmetic on the REAL slates
        ft = FusedTrainer(m_f, loss_name, {}, bs, L, lr=1e-3, use_graph=True)
        for epoch in range(2):
            for j in (0, 16, 32):
                xb, yb = x[j:j + bs], y[j:j + bs]
                real = xb.shape[0]
                m_a.load_state_dict(m_f.state_dict())               # same weights on both sides at every step
                m_a.zero_grad(set_to_none=True)
                la = getattr(E, loss_name)(m_a(xb, yb == -1, None), yb)
                la.backward()
                if real < bs:
                    padn = bs - real
                    xb = torch.cat([xb, xb.new_zeros((padn, L, F))])
                    yb = torch.cat([yb, yb.new_full((padn, L), -1.0)])
                lf = ft.step(xb, yb, None, global_batch=real)
                assert abs(float(lf.item()) - float(la.item())) <= 2e-6 * (1 + abs(float(la.item()))), (loss_name, epoch, j, float(lf), float(la))
                gmax = max(float(pa.grad.abs().max()) for pa in m_a.parameters())
                for (n, pf), (_, pa) in zip(m_f.named_parameters(), m_a.named_parameters()):
                    assert float((pf.grad - pa.grad).abs().max()) <= 2e-4 * gmax, (loss_name, epoch, j, n)
        assert ft.graph is not None          # the full batches did get their graph


def test_captured_steps_form_an_lru_over_batch_divisors():
    """captured steps are keyed by the batch divisor; a fifth distinct divisor evicts the least recently used
    capture (one warning) instead of silently running eagerly, and every divisor -- captured, evicted, re-captured -- gives the
    loss of the eager step on the same weights."""
    import copy
    import warnings
    from allrank_amd.engine import FusedTrainer
    L, F, bs = 30, 20, 16
    x, y, _ = (t.to(DEV) for t in _data(16, L, F, 9))
    m_g = _model(F)
    m_e = copy.deepcopy(m_g)
    tg = FusedTrainer(m_g, "approxNDCGLoss", {}, bs, L, lr=1e-3, use_graph=True)
    te = FusedTrainer(m_e, "approxNDCGLoss", {}, bs, L, lr=1e-3, use_graph=False)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for div in (16, 16, 16, 15, 14, 13, 12, 11, 16, 15, 11, 12):
            lg, le = tg.step(x, y, None, global_batch=div), te.step(x, y, None, global_batch=div)
            assert torch.equal(lg, le) and torch.equal(tg.flat_p, te.flat_p), div
            assert len(tg._graphs) <= tg.max_graphs
    assert tg.use_graph and len(tg._graphs) == tg.max_graphs
    assert sum("evicting the least recently used" in str(w.message) for w in rec) == 1
    assert (11.0, True) in tg._graphs and (12.0, True) in tg._graphs     # the most recent divisors are the live captures


@pytest.mark.parametrize("name,kw", [("Adam", dict(lr=2e-3, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.01)),
                                     ("AdamW", dict(lr=2e-3, weight_decay=0.05)),
                                     ("SGD", dict(lr=0.05, momentum=0.9, nesterov=True, weight_decay=1e-3)),
                                     ("SGD", dict(lr=0.05))])
def test_fused_optimizers_equal_torch_optim(name, kw):
    """`getattr(torch.optim, config.optimizer.name)(**args)` (main.py:82): Adam with non-default betas / eps / L2 weight decay, AdamW and
    SGD (momentum, Nesterov, weight decay) run on the explicit step -- the flat-buffer update kernels against torch's own optimizer
    driven by the same gradients, five steps; and fit() picks the fused engine for them."""
    import copy
    from functools import partial
    from allrank_amd import losses as E, fit as EF
    from allrank_amd.engine import FusedTrainer
    L, F, bs = 30, 20, 16
    x, y, _ = (t.to(DEV) for t in _data(16, L, F, 9))
    m_f = _model(F)
    m_t = copy.deepcopy(m_f)
    opt_t = getattr(torch.optim, name)(m_t.parameters(), **kw)
    spec, reason = EF._fused_spec(m_f, partial(E.listNet), getattr(torch.optim, name)(m_f.parameters(), **kw))
    assert spec is not None and reason == "", reason
    ft = FusedTrainer(m_f, spec[0], spec[1], bs, L, lr=spec[2], use_graph=True, **spec[3])
    for step in range(5):
        ft.step(x, y, None)
        # torch's optimizer on the torch copy, fed the ENGINE's gradients of this step (the update rule is what is under test)
        for pt, pf in zip(m_t.parameters(), m_f.parameters()):
            pt.grad = pf.grad.detach().clone()
        opt_t.step()
        for (n, pt), (_, pf) in zip(m_t.named_parameters(), m_f.named_parameters()):
            assert float((pt - pf).detach().abs().max()) <= 2e-6 * max(1.0, float(pt.detach().abs().max())), (name, step, n)
        m_t.load_state_dict(m_f.state_dict())             # (keep the two weight sets bit-identical: only the optimizer state runs free)


def test_fit_falls_back_to_the_autograd_trainer(tmp_path):
    from allrank_amd import losses as E, fit as EF
    cfg = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric="ndcg_5")
    train_dl, val_dl = _loaders()
    model = _model(20)
    opt = torch.optim.RMSprop(model.parameters(), lr=0.001)              # (Adam / AdamW / SGD are fused; anything else keeps torch's optimizer)
    res = EF.fit(2, model, partial(E.listNet), opt, None, train_dl, val_dl, cfg, None, 5, torch.device(DEV), str(tmp_path), None)
    assert EF.last_run["engine"] == "autograd" and "RMSprop" in EF.last_run["reason"] and res["epochs"] == 1


def test_evaluate_equals_compute_metrics_of_the_oracle():
    """epoch NDCG@{5,10,30,60} of allrank_amd.data.evaluate == mean over all slates of the reference's
    metrics.ndcg (train_utils.py:32-56) as restated by oracle/ltr_oracle.py, on the SAME scores, for slates of very different
    lengths batched to the longest one (the validation transform, dataset_loading.py:185-194)."""
    from allrank_amd.data import DeviceSlates, evaluate
    rng = np.random.default_rng(3)
    F = 20
    lens = rng.integers(1, 90, 70)
    lens[5] = 1
    X = rng.standard_normal((int(lens.sum()), F)).astype(np.float32)
    y = rng.choice(5, size=int(lens.sum()), p=[0.5, 0.3, 0.15, 0.03, 0.02]).astype(np.float32)
    qid = np.repeat(np.arange(len(lens)), lens)
    y[qid == 7] = 0.0                                             # a slate without any relevant item -> NDCG 1.0 (metrics.py:24)
    ds = DeviceSlates(X, y, qid, device=DEV)
    model = _model(F)
    ats = [5, 10, 30, 60]
    got = evaluate(model, ds, {"ndcg": ats, "mrr": [10]}, batch_size=16)
    # the same scores through the oracle, slate by slate padded to the longest
    Lmax = int(lens.max())
    vals = []
    model.eval()
    with torch.no_grad():
        for xb, yb, idx in ds.batches(16, None):
            sc = model.score(xb, yb == -1, idx).cpu().numpy()
            vals.append(O.ndcg(sc, yb.cpu().numpy(), ats=ats)[0])
            assert xb.shape[1] == Lmax
    ref = np.concatenate(vals).mean(0)
    for at, r in zip(ats, ref):
        assert abs(got["ndcg_%d" % at] - float(r)) <= 1e-5, (at, got["ndcg_%d" % at], r)
    assert 0.0 <= got["mrr_10"] <= 1.0


@pytest.mark.parametrize("shape,K,N,act", [((7, 33), 136, 512, 0), ((96, 240), 512, 2048, 1), ((5, 9), 20, 1, 0), ((3, 11), 45, 64, 0)])
def test_ops_linear_matches_fp64(shape, K, N, act):
    """ops.linear (nn.Linear of the drop-in nn.Module path on the split-bf16 GEMMs): forward and all three gradients against
    fp64 torch; shapes the kernels do not take (K % 4 != 0) fall back to F.linear and must agree just the same."""
    from allrank_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(*shape, K, device=DEV, generator=g, requires_grad=True)
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device=DEV, generator=g, requires_grad=True)
    go = torch.randn(*shape, N, device=DEV, generator=g)
    y = ops.linear(x, w, b, act)
    y.backward(go)
    x64, w64, b64 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    if act:                                             # the engine's own ReLU mask: a pre-activation within round-off of 0 may
        y64 = y64 * (y.detach() > 0)                    # land on either side in ANY finite-precision forward
    y64.backward(go.double())
    M = int(np.prod(shape))
    for name, a, r, scale in (("y", y, y64, None), ("dx", x.grad, x64.grad, None), ("dw", w.grad, w64.grad, None), ("db", b.grad, b64.grad, None)):
        ref = r.detach()
        err = float((a.detach().double() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (name, err, float(ref.abs().max()))


def test_fused_trainer_score_matches_module_eval_forward():
    """FusedTrainer.score = model.eval(); model.score(...) (dropout off) through the forward half of the explicit step, also
    right after training steps with dropout and from its hipGraph replay"""
    from allrank_amd.engine import FusedTrainer
    from allrank_amd.model import make_model
    torch.manual_seed(5)
    B, L, F = 6, 40, 24
    model = make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                       dict(N=2, d_ff=64, h=4, positional_encoding=None, dropout=0.3), dict(d_output=1, output_activation=None), F).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(B, L, F, device=DEV, generator=g)
    y = torch.randint(0, 5, (B, L), device=DEV, generator=g).float()
    y[1, 25:] = -1
    x[1, 25:] = 0
    ft = FusedTrainer(model, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=True)
    for i in range(5):
        ft.step(x, y)                                   # training steps (dropout on) move the weights
        sc = ft.score(x, y).clone()                     # i = 0, 1 eager warm-up, then capture, then replays
        model.eval()
        with torch.no_grad():
            ref = model.score(x, y == -1, None)
        model.train()
        valid = y != -1
        assert float((sc - ref)[valid].abs().max()) <= 2e-5 * max(1.0, float(ref[valid].abs().max())), i


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_ops_feed_forward_matches_fp64(p):
    """ops.feed_forward (PositionwiseFeedForward as one autograd node): forward and all five gradients against fp64 torch.
    The dropout mask depends only on (seed, element index): it is read off a first call whose second projection is the identity
    (y = the dropped, rectified hidden activation itself) and then used in the fp64 restatement of a call with general weights."""
    from allrank_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    shape, K, Fh, seed = (6, 50), 32, 64, 12345
    x = torch.randn(*shape, K, device=DEV, generator=g, requires_grad=True)
    w1 = (torch.randn(Fh, K, device=DEV, generator=g) / K ** 0.5).requires_grad_(True)
    b1 = torch.randn(Fh, device=DEV, generator=g, requires_grad=True)
    eye, zero = torch.eye(Fh, device=DEV), torch.zeros(Fh, device=DEV)
    with torch.no_grad():
        r = ops.feed_forward(x, w1, b1, eye, zero, p, seed)
        assert torch.equal(r, ops.feed_forward(x, w1, b1, eye, zero, p, seed))          # same seed, same mask
    keep = (r != 0).double() / (1.0 - p)                 # ReLU and dropout combined (pre-activations at exactly 0 have measure 0)
    if p:
        pre = torch.nn.functional.linear(x.detach(), w1.detach(), b1.detach())
        frac = float(((r == 0) & (pre > 1e-3)).double().sum() / (pre > 1e-3).double().sum())
        assert abs(frac - p) < 0.03, frac                # the rate of the counter-based generator
    N = 40
    w2 = (torch.randn(N, Fh, device=DEV, generator=g) / Fh ** 0.5).requires_grad_(True)
    b2 = torch.randn(N, device=DEV, generator=g, requires_grad=True)
    go = torch.randn(*shape, N, device=DEV, generator=g)
    y = ops.feed_forward(x, w1, b1, w2, b2, p, seed)
    y.backward(go)
    t64 = [t.detach().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    h64 = torch.nn.functional.linear(t64[0], t64[1], t64[2]) * keep
    y64 = torch.nn.functional.linear(h64, t64[3], t64[4])
    y64.backward(go.double())
    for name, a, ref in [("y", y, y64)] + [(n, t.grad, t6.grad) for n, t, t6 in zip(("dx", "dw1", "db1", "dw2", "db2"), (x, w1, b1, w2, b2), t64)]:
        err = float((a.detach().double() - ref.detach()).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.detach().abs().max())), (name, err)


@pytest.mark.parametrize("p", [0.0, 0.2])
def test_attention_packed_equals_sliced_attention(p):
    """ops.attention_packed(qkv) == ops.attention(q, k, v) on the column blocks of the same tensor, bit for bit, output and gradient"""
    from allrank_amd import ops
    g = torch.Generator(device=DEV).manual_seed(21)
    B, L, h, dk = 3, 70, 4, 64
    d = h * dk
    qkv0 = torch.randn(B, L, 3 * d, device=DEV, generator=g)
    mask = torch.zeros(B, L, dtype=torch.bool, device=DEV)
    mask[1, 50:] = True
    go = torch.randn(B, L, d, device=DEV, generator=g)
    a = qkv0.clone().requires_grad_(True)
    oa = ops.attention_packed(a, mask, h, p, seed=77)
    oa.backward(go)
    b = qkv0.clone().requires_grad_(True)
    ob = ops.attention(b[:, :, :d], b[:, :, d:2 * d], b[:, :, 2 * d:], mask, h, p, seed=77)
    ob.backward(go)
    assert torch.equal(oa, ob) and torch.equal(a.grad, b.grad)


def test_fit_validation_with_longer_slates_than_training(tmp_path):
    """the reference validates on slates padded to the longest query (dataset_loading.py:185-194), i.e. usually LONGER than the
    training slate length: the static-shape scorer of the fused step does not apply and the validation pass goes through the
    nn.Module forward (same kernels through ops.linear / feed_forward / attention_packed) -- values must equal a direct evaluation"""
    from torch.utils.data import DataLoader, TensorDataset
    from allrank_amd import losses as E, fit as EF, metrics as EM
    cfg = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric="ndcg_5")
    train_dl = DataLoader(TensorDataset(*_data(32, 30, 20, 1)), batch_size=16, shuffle=False)
    xv, yv, iv = _data(20, 47, 20, 2)
    val_dl = DataLoader(TensorDataset(xv, yv, iv), batch_size=16, shuffle=False)
    model = _model(20)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    res = EF.fit(epochs=2, model=model, loss_func=partial(E.listNet), optimizer=opt, scheduler=None, train_dl=train_dl, valid_dl=val_dl,
                 config=cfg, gradient_clipping_norm=None, early_stopping_patience=10, device=torch.device(DEV), output_dir=str(tmp_path),
                 tensorboard_output_path=None)
    assert EF.last_run["engine"] == "fused"
    model.eval()
    with torch.no_grad():
        sc = model.score(xv.to(DEV), (yv == -1).to(DEV), iv.to(DEV))
        ref = float(EM.ndcg(sc, yv.to(DEV), ats=[5]).mean())
    assert abs(float(res["val_metrics"]["ndcg_5"]) - ref) < 1e-6


def test_prefetcher_delivers_the_loader_batches_unchanged():
    """allrank_amd.fit._Prefetcher: pinned double buffering + copy stream must hand over exactly the loader's batches, including a
    short last batch, device-resident batches and a consumer that keeps launching work on the compute stream"""
    from allrank_amd.fit import _Prefetcher
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(n, 17, 5, generator=g), torch.randint(0, 5, (n, 17), generator=g).float(), torch.randint(0, 17, (n, 17), generator=g))
               for n in (8, 8, 8, 8, 8, 3)]
    sink = torch.zeros(2048, 2048, device=DEV)
    got = []
    for xb, yb, ib in _Prefetcher(batches, DEV):
        assert xb.is_cuda and yb.is_cuda and ib.is_cuda
        got.append((xb.clone(), yb.clone(), ib.clone()))
        for _ in range(3):
            sink = sink @ sink * 1e-3                    # keep the compute stream busy behind the consumer's reads
    assert len(got) == len(batches)
    for (a, b, c), (x, y, i) in zip(got, batches):
        assert torch.equal(a.cpu(), x) and torch.equal(b.cpu(), y) and torch.equal(c.cpu(), i)
    dev_batches = [tuple(t.to(DEV) for t in b) for b in batches[:2]]
    for (a, b, c), (x, y, i) in zip(_Prefetcher(dev_batches, DEV), dev_batches):
        assert a is x and b is y and c is i
