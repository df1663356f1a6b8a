"""-m gpu: the device-resident loader BEHIND the reference's loader names:
``allrank_amd.data.load_libsvm_dataset`` / ``create_data_loaders`` -- what ``install()`` binds to allrank/data/dataset_loading.py:197-248
and main.py:8 -- against the restated host loader of the reference (oracle/loader_oracle.py, pinned to the reference's own loaders bit
for bit by tests/test_loader_cpu.py; the reference tree itself cannot travel to the GPU box):

  * libsvm files -> device parse -> DeviceLoader batches == torch DataLoader + FixLength + ToTensor batches, same seeds, two epochs of
    the reference's loader traffic: bit for bit on the padding branch, as sets on the rows FixLength permutes;
  * the sampling branch through the loader (subset, gather, relevance rules), and rank blocks == the one-rank batches bit for bit
    INCLUDING sampled slates (the draw is keyed by the slate id, not by the batch row);
  * ``fit()`` fed by the device loader == ``fit()`` fed by the host loader: identical weights after every epoch;
  * ``train_metrics="reference"`` and the finiteness switch (config.detect_anomaly, main.py:89).
"""
import json
import os
import types
from functools import partial

import numpy as np
import pytest
import torch

from oracle import loader_oracle as LO
from tests.test_loader_cpu import _write, _seed, _epochs, _same, _canon

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _device_loaders(path, slate_length, batch_size, rank=0, world=1):
    from allrank_amd import data as ED
    tr, va = ED.load_libsvm_dataset(path, slate_length, "vali", device=DEV)
    return (ED.DeviceLoader(tr, world * batch_size, shuffle=True, rank=rank, world=world),
            ED.DeviceLoader(va, world * batch_size, shuffle=False, rank=rank, world=world))


def test_rebound_loaders_equal_the_reference_loaders_padded_only(tmp_path):
    from allrank_amd import data as ED
    path = _write(tmp_path)
    _seed()
    ref = _epochs(*LO.create_data_loaders(*LO.load_libsvm_dataset(path, 40, "vali"), num_workers=0, batch_size=8))
    state_ref = torch.get_rng_state()
    _seed()
    tr_ds, va_ds = ED.load_libsvm_dataset(path, 40, "vali", device=DEV)                    # main.py:57-61
    assert tr_ds.shape[-1] == va_ds.shape[-1] == 9                                         # main.py:63-64
    tr, va = ED.create_data_loaders(tr_ds, va_ds, num_workers=1, batch_size=8)             # main.py:67-68
    assert isinstance(tr, ED.DeviceLoader) and tr.batch_size == 8 * max(1, torch.cuda.device_count())
    tr, va = ED.DeviceLoader(tr_ds, 8, shuffle=True), ED.DeviceLoader(va_ds, 8, shuffle=False)
    mine = _epochs(tr, va)
    assert torch.equal(torch.get_rng_state(), state_ref)            # the same draws from torch's global generator
    assert all(t.is_cuda for b in mine for t in b)
    _same(ref, mine, canon=True)
    # the training batches (slate_length 40 > longest slate 29: padding branch only) are bit-identical without any canonical order
    n_tr = len(tr)
    _same(ref[:n_tr], mine[:n_tr])


def test_sampling_branch_through_the_loader_and_rank_blocks(tmp_path):
    from sklearn.datasets import load_svmlight_file
    from allrank_amd.parallel import shard_slates
    path = _write(tmp_path, n_q=33, long={3: 33, 8: 12, 20: 60})
    L = 12
    X, y, q = load_svmlight_file(os.path.join(path, "train.txt"), query_id=True)
    X = np.asarray(X.todense(), dtype=np.float32)
    host = LO.HostSlates(X, y, q)
    _seed()
    tr, _ = _device_loaders(path, L, 8)
    ids = [c.clone() for c in tr._ids]                               # the slate ids of every batch (consumes one iteration's draws)
    _seed()
    one = list(tr)
    assert [int(b[0].shape[0]) for b in one] == [int(c.numel()) for c in ids]
    sampled = 0
    for c, (xb, yb, ib) in zip(ids, one):
        xb, yb, ib = xb.cpu().numpy(), yb.cpu().numpy(), ib.cpu().numpy()
        for r, s in enumerate(c.tolist()):
            xs, ys = host.xs[s].astype(np.float32), host.ys[s].astype(np.float32)
            n = len(ys)
            if n < L:
                assert ib[r, :n].tolist() == list(range(n)) and (ib[r, n:] == -1).all() and (yb[r, n:] == -1).all() and not xb[r, n:].any()
                assert np.array_equal(xb[r, :n], xs) and np.array_equal(yb[r, :n], ys)
            else:
                sampled += 1
                ii = ib[r]
                assert len(set(ii.tolist())) == L and ii.min() >= 0 and ii.max() < n
                assert np.array_equal(xb[r], xs[ii]) and np.array_equal(yb[r], ys[ii])
                if ys.sum() > 0:
                    assert yb[r].sum() > 0                                                # dataset_loading.py:71-76
    assert sampled >= 4
    # a new iteration draws a new sample (numpy's global generator moved on), the same seeds reproduce the epoch
    again = list(tr)
    assert any(not torch.equal(a[2], b[2]) for a, b in zip(one, again))
    # rank blocks == the one-rank batches, sampled slates included
    for world in (2, 3):
        _seed()
        whole = list(_device_loaders(path, L, 4 * world)[0])
        blocks = []
        for r in range(world):
            _seed()
            blocks.append(list(_device_loaders(path, L, 4, rank=r, world=world)[0]))
        for k, w in enumerate(whole):
            n = int(w[0].shape[0])
            for r in range(world):
                lo, hi = shard_slates(n, r, world)
                b = blocks[r][k]
                assert b.global_slates == n and b.offset == lo and b.order_tag == blocks[0][k].order_tag
                assert torch.equal(b.lengths.to(DEV), (b[1] != -1).sum(1).to(torch.int32))
                for j in range(3):
                    assert torch.equal(b[j], w[j][lo:hi])


CONFIG = {
    "model": {"fc_model": {"sizes": [64], "input_norm": False, "activation": None, "dropout": 0.0},
              "transformer": {"N": 1, "d_ff": 64, "h": 2, "positional_encoding": None, "dropout": 0.0},
              "post_model": {"output_activation": None, "d_output": 1}},
    "optimizer": {"name": "Adam", "args": {"lr": 0.001}},
    "training": {"epochs": 3, "early_stopping_patience": 100, "gradient_clipping_norm": None},
}


def _fit_job(path, kind, tmp_path, slate_length=40, loss_name="approxNDCGLoss", detect_anomaly=False, poison=False, **fit_kw):
    from torch import optim
    from allrank_amd import data as ED, fit as EF, losses
    from allrank_amd.model import make_model
    _seed()
    torch.cuda.manual_seed_all(42)
    if kind == "device":
        tr_ds, va_ds = ED.load_libsvm_dataset(path, slate_length, "vali", device=DEV)
        if poison:
            tr_ds.slates.x_items[7, 2] = float("nan")
        tr, va = ED.DeviceLoader(tr_ds, 8, shuffle=True), ED.DeviceLoader(va_ds, 8, shuffle=False)
    else:
        tr_ds, va_ds = LO.load_libsvm_dataset(path, slate_length, "vali")
        tr, va = LO.create_data_loaders(tr_ds, va_ds, num_workers=0, batch_size=8)
    model = make_model(n_features=tr_ds.shape[-1], **json.loads(json.dumps(CONFIG["model"])))
    model.to(DEV)
    optimizer = getattr(optim, CONFIG["optimizer"]["name"])(params=model.parameters(), **CONFIG["optimizer"]["args"])
    loss_func = partial(getattr(losses, loss_name))
    config = types.SimpleNamespace(metrics={"ndcg": [5, 10]}, val_metric="ndcg_5", detect_anomaly=detect_anomaly)
    out = tmp_path / ("out_%s" % kind)
    out.mkdir(exist_ok=True)
    log, orig = [], EF.log.info

    def spy(msg, *a):
        if isinstance(msg, str) and msg.startswith("Epoch :"):
            log.append((float(a[1]), float(a[2]), {k: v.detach().clone() for k, v in model.state_dict().items()}))
        return orig(msg, *a)
    EF.log.info = spy
    try:
        result = EF.fit(model=model, loss_func=loss_func, optimizer=optimizer, scheduler=None, train_dl=tr, valid_dl=va, config=config,
                        device=torch.device(DEV), output_dir=str(out), tensorboard_output_path=None, **CONFIG["training"], **fit_kw)
    finally:
        EF.log.info = orig
    assert EF.last_run["engine"] == "fused", EF.last_run
    return result, log, model, (tr_ds, va_ds), dict(EF.last_run)


def test_fit_from_the_device_loader_equals_fit_from_the_host_loader(tmp_path):
    """same seeds: the device loader hands fit() the batches the reference's loader would -- every epoch's training loss and the
    weights after every epoch are IDENTICAL (deterministic kernels on identical inputs); the validation numbers agree to round-off
    (FixLength permutes the longest validation slates at random; loss and NDCG do not depend on item order beyond that)"""
    path = _write(tmp_path, n_q=50, F=12)
    r_h, log_h, _, _, _ = _fit_job(path, "host", tmp_path)
    r_d, log_d, _, _, run = _fit_job(path, "device", tmp_path)
    assert len(log_h) == len(log_d) == 3
    for (t1, v1, w1), (t2, v2, w2) in zip(log_h, log_d):
        assert t1 == t2, (t1, t2)
        assert abs(v1 - v2) <= 1e-5 * (1 + abs(v1))
        assert all(torch.equal(w1[k], w2[k]) for k in w1)
    for k in r_h["train_metrics"]:
        assert r_h["train_metrics"][k] == r_d["train_metrics"][k]
        assert abs(r_h["val_metrics"][k] - r_d["val_metrics"][k]) <= 1e-6
    # what fit() recorded about the run: one entry per epoch, every slate of the set trained on once per epoch
    assert [e["slates"] for e in run["epoch_log"]] == [50, 50, 50] and all(e["slots"] == 50 * 40 for e in run["epoch_log"])
    assert all(e["train_s"] > 0 and e["val_s"] > 0 for e in run["epoch_log"])


def test_fit_variable_length_from_the_device_loader_uses_host_lengths(tmp_path):
    """ragged data (< 80 % valid slots) -> variable-length execution; the DeviceLoader supplies the slate lengths from the host, so
    the step never counts valid items on the device (no per-step sync) -- and trains exactly what the padded step trains"""
    from allrank_amd.engine import FusedTrainer
    path = _write(tmp_path, n_q=50, F=12)
    seen = []
    orig = FusedTrainer._pack

    def spy(self, xb, lengths):
        seen.append(lengths is not None)
        return orig(self, xb, lengths)
    FusedTrainer._pack = spy
    try:
        r_c, log_c, _, _, run = _fit_job(path, "device", tmp_path)            # compact=None -> auto (35 % valid)
    finally:
        FusedTrainer._pack = orig
    assert run["compact"] is True and seen and all(seen)
    r_p, log_p, _, _, run_p = _fit_job(path, "device", tmp_path, compact=False)
    assert run_p["compact"] is False
    for (t1, v1, w1), (t2, v2, w2) in zip(log_c, log_p):
        assert abs(t1 - t2) <= 1e-5 * (1 + abs(t1)) and abs(v1 - v2) <= 1e-5 * (1 + abs(v1))
    werr = max(float((log_c[-1][2][k] - log_p[-1][2][k]).abs().max()) for k in log_c[-1][2])
    assert werr <= 21 * 1.1e-3                                                 # (21 Adam steps of lr 1e-3; see test_gpu_main_sequence)


def test_train_metrics_reference_mode_is_the_second_pass(tmp_path):
    """train_utils.py:99: the reference computes train metrics in a second pass with the END-of-epoch weights.  The last epoch's
    numbers of ``train_metrics="reference"`` must be what an independent evaluation of the final model gives on the training set;
    the default (metrics of the training forward, weights moving) differs from it; both modes train identical weights."""
    from allrank_amd import data as ED
    path = _write(tmp_path, n_q=50, F=12)
    r_ref, log_ref, model, (tr_ds, _), _ = _fit_job(path, "device", tmp_path, train_metrics="reference")
    r_def, log_def, _, _, _ = _fit_job(path, "device", tmp_path)
    assert all(torch.equal(log_ref[-1][2][k], log_def[-1][2][k]) for k in log_ref[-1][2])
    ev = ED.evaluate(model, tr_ds.slates, {"ndcg": [5, 10]}, batch_size=8, slate_length=40)
    for k, v in ev.items():
        assert abs(float(r_ref["train_metrics"][k]) - v) <= 2e-6, (k, r_ref["train_metrics"], ev)
    assert any(abs(float(r_ref["train_metrics"][k]) - float(r_def["train_metrics"][k])) > 1e-4 for k in ev)
    assert all(abs(float(r_ref["val_metrics"][k]) - float(r_def["val_metrics"][k])) <= 1e-6 for k in ev)
    with pytest.raises(ValueError, match="train_metrics"):
        _fit_job(path, "device", tmp_path, train_metrics="twice")


def test_finiteness_switch_names_the_first_offending_tensor(tmp_path):
    """config.detect_anomaly (main.py:89) on the explicit step: a NaN feature in one training item -> FloatingPointError at the
    first step that sees it, naming the first parameter tensor (flat-buffer order) whose gradient is not finite; the kernel itself:
    a NaN planted in one parameter's gradient slice is found, by name, with its count"""
    from allrank_amd.engine import FusedTrainer
    from allrank_amd.model import make_model
    path = _write(tmp_path, n_q=50, F=12)
    with pytest.raises(FloatingPointError) as e:
        _fit_job(path, "device", tmp_path, detect_anomaly=True, poison=True)
    msg = str(e.value)
    assert "input_layer.layers.0.weight" in msg and "non-finite" in msg and "epoch 0" in msg
    # clean data: the switch costs a sync per step and changes nothing
    r_on, log_on, _, _, _ = _fit_job(path, "device", tmp_path, detect_anomaly=True)
    r_off, log_off, _, _, _ = _fit_job(path, "device", tmp_path)
    assert all(torch.equal(log_on[-1][2][k], log_off[-1][2][k]) for k in log_on[-1][2])
    # the kernel
    model = make_model(n_features=12, **json.loads(json.dumps(CONFIG["model"]))).to(DEV)
    ft = FusedTrainer(model, "listNet", {}, 4, 16, use_graph=False)
    assert ft.first_nonfinite() == (None, 0)
    p = model.encoder.layers[0].feed_forward.w_2.bias
    p.grad[3] = float("inf")
    model.output_layer.w_1.weight.grad[0, 5] = float("nan")
    model.output_layer.w_1.weight.grad[0, 6] = float("-inf")
    assert ft.first_nonfinite() == ("encoder.layers.0.feed_forward.w_2.bias", 3)
