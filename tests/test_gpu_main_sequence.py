"""-m gpu: the call sequence of allrank/main.py:34-110, reproduced line for line with this repository's modules only (the reference
tree cannot travel to the GPU box and must not be copied into the repository, so ``test_reference_main.py``'s run of the UNMODIFIED
main.py stays skipped there; this is its in-tree twin -- VERDICT r3 item 5):

    seeds (main.py:36-38) -> libsvm files shaped like allrank/data/generate_dummy_data.py:31-42 writes them (100 queries x 20 docs x
    20 features, labels 0..4) -> a dataset padded to ``slate_length`` the way FixLength pads (dataset_loading.py:81-93) and
    DataLoaders built the way create_data_loaders builds them (:232-248: shuffle for train, none for validation, drop_last=False)
    -> make_model(n_features=..., **config.model) (main.py:75) -> getattr(torch.optim, name)(params=..., **args) (:82) ->
    partial(getattr(losses, name), **args) (:83) -> getattr(lr_scheduler, name)(optimizer, **args) (:85) ->
    fit(model=, loss_func=, optimizer=, scheduler=, train_dl=, valid_dl=, config=, device=, output_dir=, tensorboard_output_path=,
    **config.training) (:90-102) -> the result through dump_experiment_result's ``.item()`` calls and assert_expected_metrics'
    comparison (utils/experiments.py:16-43).

``fit`` is ``allrank_amd.fit.fit`` -- what ``install(fit=True)`` binds to ``allrank.main.fit`` -- and must land on the explicit step
(``last_run["engine"] == "fused"``): the run_example job (FC + one transformer layer + ListNet: the hipGraph-captured step) and
BASELINE configs[0] as SURVEY 8(d) prescribes it (transformer null: the slate-resident FC + ListNet step of csrc/ltrx_fcstep.hip).
"""
import json
import os
import types
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIG = {
    "model": {"fc_model": {"sizes": [64], "input_norm": False, "activation": None, "dropout": 0.0},
              "transformer": {"N": 1, "d_ff": 64, "h": 1, "positional_encoding": None, "dropout": 0.0},
              "post_model": {"output_activation": None, "d_output": 1}},
    "data": {"path": None, "validation_ds_role": "vali", "num_workers": 0, "batch_size": 32, "slate_length": 24},
    "optimizer": {"name": "Adam", "args": {"lr": 0.001}},
    "lr_scheduler": {"name": "StepLR", "args": {"step_size": 3, "gamma": 0.5}},
    "training": {"epochs": 4, "early_stopping_patience": 100, "gradient_clipping_norm": None},
    "val_metric": "ndcg_5", "metrics": ["ndcg_5"],
    "loss": {"name": "listNet", "args": {}},
    "expected_metrics": {"val": {"ndcg_5": 0.3}},
}


def _dummy_libsvm(path, rng, num_queries=100, results_len=20, num_labels=5, num_features=20):
    from sklearn.datasets import dump_svmlight_file
    X = rng.standard_normal((num_queries * results_len, num_features))
    y = rng.integers(0, num_labels, num_queries * results_len)
    qid = np.repeat(np.arange(num_queries), results_len)
    dump_svmlight_file(X, y, path, query_id=qid)


def _load(path, slate_length):
    """libsvm -> padded slates (features 0, label -1, index -1 on padding), as LibSVMDataset + FixLength + ToTensor deliver them"""
    from sklearn.datasets import load_svmlight_file
    X, y, q = load_svmlight_file(path, query_id=True)
    X = np.asarray(X.todense(), dtype=np.float32)
    xs, ys, ids = [], [], []
    for qq in np.unique(q):
        sel = np.nonzero(q == qq)[0][:slate_length]
        n = len(sel)
        xb = np.zeros((slate_length, X.shape[1]), np.float32)
        yb = np.full(slate_length, -1.0, np.float32)
        ib = np.full(slate_length, -1, np.int64)
        xb[:n], yb[:n], ib[:n] = X[sel], y[sel], np.arange(n)
        xs.append(xb); ys.append(yb); ids.append(ib)
    return torch.utils.data.TensorDataset(torch.tensor(np.stack(xs)), torch.tensor(np.stack(ys)), torch.tensor(np.stack(ids)))


def _main_sequence(cfg, tmp_path):
    from torch import optim
    import allrank_amd
    from allrank_amd import fit as EF, losses
    from allrank_amd.model import make_model
    torch.manual_seed(42)                                              # main.py:36-38
    torch.cuda.manual_seed_all(42)
    np.random.seed(42)
    rng = np.random.default_rng(42)
    data = tmp_path / "dummy_data"
    data.mkdir()
    for role in ("train", "vali"):
        _dummy_libsvm(str(data / ("%s.txt" % role)), rng)
    out_dir = tmp_path / "job" / "results" / "r4"
    os.makedirs(out_dir)
    train_ds = _load(str(data / "train.txt"), cfg["data"]["slate_length"])
    val_ds = _load(str(data / "vali.txt"), cfg["data"]["slate_length"])
    n_features = train_ds.tensors[0].shape[-1]
    assert n_features == val_ds.tensors[0].shape[-1]
    gen = torch.Generator().manual_seed(42)
    train_dl = torch.utils.data.DataLoader(train_ds, batch_size=cfg["data"]["batch_size"], shuffle=True, generator=gen, num_workers=0)
    val_dl = torch.utils.data.DataLoader(val_ds, batch_size=cfg["data"]["batch_size"] * 2, shuffle=False, num_workers=0)
    dev = torch.device("cuda:0")
    model = make_model(n_features=n_features, **json.loads(json.dumps(cfg["model"])))
    model.to(dev)
    optimizer = getattr(optim, cfg["optimizer"]["name"])(params=model.parameters(), **cfg["optimizer"]["args"])
    loss_func = partial(getattr(losses, cfg["loss"]["name"]), **cfg["loss"]["args"])
    scheduler = (getattr(optim.lr_scheduler, cfg["lr_scheduler"]["name"])(optimizer, **cfg["lr_scheduler"]["args"])
                 if cfg["lr_scheduler"]["name"] else None)
    config = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric=cfg["val_metric"], expected_metrics=cfg["expected_metrics"])
    w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    result = EF.fit(model=model, loss_func=loss_func, optimizer=optimizer, scheduler=scheduler, train_dl=train_dl, valid_dl=val_dl,
                    config=config, device=dev, output_dir=str(out_dir), tensorboard_output_path=None, **cfg["training"])
    # dump_experiment_result (utils/experiments.py:16-31): every metric and num_params go through .item()
    flat = {"train_metrics/%s" % k: v.item() for k, v in result["train_metrics"].items()}
    flat.update({"val_metrics/%s" % k: v.item() for k, v in result["val_metrics"].items()})
    flat["num_params"] = result["num_params"].item()
    flat["epochs"] = result["epochs"]
    with open(os.path.join(str(out_dir), "experiment_result.json"), "w") as fh:
        json.dump(flat, fh)
    # assert_expected_metrics (utils/experiments.py:34-43)
    for role, metrics in config.expected_metrics.items():
        for name, expected in metrics.items():
            assert result["%s_metrics" % role][name] >= expected, (role, name, result)
    return EF, model, w0, flat, out_dir


@pytest.mark.parametrize("job", ["run_example", "fc_listnet", "fc_listnet_padded_240"])
def test_main_call_sequence_trains_on_the_fused_step(tmp_path, job):
    cfg = json.loads(json.dumps(CONFIG))
    if job.startswith("fc_listnet"):                                   # BASELINE configs[0] / SURVEY 8(d) config (1): transformer null
        cfg["model"]["transformer"] = None
    if job == "fc_listnet_padded_240":                                 # run_example pads its 20-document slates to slate_length 240
        cfg["data"]["slate_length"] = 240                              # (92 % padding)
    EF, model, w0, flat, out_dir = _main_sequence(cfg, tmp_path)
    assert EF.last_run["engine"] == "fused", EF.last_run
    if job.startswith("fc_listnet"):
        # the slate-resident step reads the padded batch in place: it is taken even where fit() would ask for variable-length execution
        assert EF.last_run["fcstep"] is True and EF.last_run["compact"] is False, EF.last_run
    assert flat["epochs"] == cfg["training"]["epochs"] - 1
    assert 0.3 <= flat["val_metrics/ndcg_5"] <= 1.0 and np.isfinite(flat["train_metrics/ndcg_5"])
    assert flat["num_params"] == sum(p.numel() for p in model.parameters())
    assert os.path.exists(os.path.join(str(out_dir), "model.pkl"))       # train_utils.py:139
    saved = torch.load(os.path.join(str(out_dir), "model.pkl"), map_location="cpu")
    moved = max(float((saved[k] - w0[k].cpu()).abs().max()) for k in w0)
    assert 1e-3 < moved < 0.5, moved                                     # 16 Adam steps of lr 1e-3 / 5e-4 actually happened
    for k, v in model.state_dict().items():                              # the module's parameters ARE the trained flat buffer
        assert torch.equal(saved[k], v.detach().cpu())


@pytest.mark.parametrize("loss_name,model", [("listNet", "attn"), ("neuralNDCG", "attn"), ("listNet", "fc_only"), ("neuralNDCG", "attn+devloader")])
def test_main_call_sequence_under_the_launcher_equals_the_one_rank_run(tmp_path, loss_name, model):
    """VERDICT r4 missing #1: N > 1 GPUs through main.py's own call sequence.  ``allrank_amd.launch.spawn`` starts two ranks (gloo,
    both on this box's one GPU); each runs tests/dist_main_worker.py -- the sequence above with the objects install() binds under a
    process group (rank device, identity wrapper, global batch = world x batch_size) and ``fit``.  Per-epoch training / validation
    loss, the trained weights and the metrics must equal the 1-rank run at the same global batch (2 x 16 vs 1 x 32 slates; the last
    batch of an epoch has 4 slates: 2 + 2).  tests/test_launch_cpu.py runs the UNMODIFIED main.run() under the same launcher."""
    import sys
    from allrank_amd import launch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    worker = os.path.join(root, "tests", "dist_main_worker.py")
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    extra = ["fc_only"] if model == "fc_only" else []        # (BASELINE configs[0]: the slate-resident FC + ListNet step, sharded)
    # "+devloader" (round 6): the two ranks read their blocks from the HBM-resident dataset through the DeviceLoader install() binds to
    # main.py:57-68; the one-rank run keeps the HOST loader -- same seeds, so the same global batches in the same order
    extra2 = extra + (["devloader"] if model.endswith("+devloader") else [])
    rc = launch.spawn(1, [sys.executable, worker, one, "32", loss_name, "0"] + extra, timeout=600, log_dir=str(tmp_path / "log1"))
    assert rc == 0, open(tmp_path / "log1" / "rank0.log").read()[-4000:]
    rc = launch.spawn(2, [sys.executable, worker, two, "16", loss_name, "1"] + extra2, backend="gloo", devices=[0, 0], timeout=600,
                      log_dir=str(tmp_path / "log2"))
    logs = "".join(open(tmp_path / "log2" / f).read()[-4000:] for f in sorted(os.listdir(tmp_path / "log2")))
    assert rc == 0, logs
    a, b = torch.load(one), torch.load(two)
    assert (a["world"], b["world"]) == (1, 2) and a["train_batch"] == b["train_batch"] == 32 and b["device"] == "cuda:0"
    assert len(a["losses"]) == len(b["losses"]) == 3
    for (t1, v1), (t2, v2) in zip(a["losses"], b["losses"]):
        assert abs(t1 - t2) <= 1e-5 * (1 + abs(t1)), ("train loss per epoch", a["losses"], b["losses"])
        assert abs(v1 - v2) <= 2e-4 * (1 + abs(v1)), ("validation loss per epoch", a["losses"], b["losses"])
    wa, wb = a["weights"], b["weights"]
    assert set(wa) == set(wb) and not any(k.startswith("module.") for k in wb)        # state_dict keys of the bare model
    werr = max(float((wa[k] - wb[k]).abs().max()) for k in wa)
    # 12 Adam steps of lr 1e-3 / 5e-4: an entry whose gradient is below its round-off may move by lr per step in either direction
    assert werr <= 12 * 1.1e-3, werr
    n_ok = sum(int(((wa[k] - wb[k]).abs() <= 5e-5).sum()) for k in wa)
    n_all = sum(v.numel() for v in wa.values())
    assert n_ok >= 0.9 * n_all, n_ok / n_all
    for k in a["val"]:
        assert abs(a["val"][k] - b["val"][k]) <= 2e-3 and abs(a["train"][k] - b["train"][k]) <= 2e-3, (a["val"], b["val"], a["train"], b["train"])
