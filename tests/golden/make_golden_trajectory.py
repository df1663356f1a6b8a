"""Generate tests/golden/trajectory_golden.npz from the REAL reference (allegro/allRank at /root/reference) on CPU: whole TRAINING
TRAJECTORIES of the reference's own entry point -- ``allrank.main.run()`` (main.py:34-110) with its own ``fit`` (train_utils.py:78-147),
its own loaders, losses, metrics and torch.optim.Adam -- on three small jobs, so that the engine's end-to-end path (install(fit=True):
device-resident loader -> explicit step -> validation) can be held against the reference's numbers epoch by epoch on the GPU box,
where the reference cannot travel ("NDCG@5 parity" end to end, not only per step).

Jobs (dropout 0, seeds 42 as main.py:36-38 sets them; every job shuffles its training set through the reference's DataLoader):
    dummy_fc_listnet    BASELINE configs[0]: run_example's dummy data (generate_dummy_data.py:31-42), FCModel[64] + ListNet, StepLR
    dummy_attn_listnet  the run_example job itself: + one transformer layer (h 1, d_ff 64)
    ragged_attn_approx  slates of 5..30 items padded to 32, FC[64] + 2 layers (h 2, d_ff 128) + ApproxNDCG, ndcg@5/@10

Recorded per job: the data (float32 features exactly as the libsvm text encodes them, labels, query ids), the config JSON, the
initial weights, and per epoch: training loss, validation loss, train metrics (the reference's second pass, train_utils.py:99),
validation metrics, the label sum and size of every training batch in order (= which slates were in which batch), the weights after
the epoch.  Recording hooks wrap ``loss_batch`` / ``compute_metrics`` of the imported package in memory; nothing of the reference
is modified or copied.

    python tests/golden/make_golden_trajectory.py        # build container only
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402

BASE = {
    "model": {"fc_model": {"sizes": [64], "input_norm": False, "activation": None, "dropout": 0.0},
              "transformer": {"N": 1, "d_ff": 64, "h": 1, "positional_encoding": None, "dropout": 0.0},
              "post_model": {"output_activation": None, "d_output": 1}},
    "data": {"path": None, "validation_ds_role": "vali", "num_workers": 0, "batch_size": 32, "slate_length": 24},
    "optimizer": {"name": "Adam", "args": {"lr": 0.001}},
    "lr_scheduler": {"name": "StepLR", "args": {"step_size": 3, "gamma": 0.5}},
    "training": {"epochs": 4, "early_stopping_patience": 100, "gradient_clipping_norm": None},
    "val_metric": "ndcg_5", "metrics": ["ndcg_5"],
    "loss": {"name": "listNet", "args": {}},
    "expected_metrics": {"val": {"ndcg_5": 0.0}},
}


def _jobs():
    a = json.loads(json.dumps(BASE))
    a["model"]["transformer"] = None
    b = json.loads(json.dumps(BASE))
    c = json.loads(json.dumps(BASE))
    c["model"]["transformer"] = {"N": 2, "d_ff": 128, "h": 2, "positional_encoding": None, "dropout": 0.0}
    c["data"].update(batch_size=16, slate_length=32)
    c["lr_scheduler"] = {"name": None, "args": {}}
    c["loss"] = {"name": "approxNDCGLoss", "args": {}}
    c["metrics"] = ["ndcg_5", "ndcg_10"]
    return [("dummy_fc_listnet", a, "dummy"), ("dummy_attn_listnet", b, "dummy"), ("ragged_attn_approx", c, "ragged")]


def _data(kind):
    """{role: (X f32, y f32, qid i64)}"""
    out = {}
    if kind == "dummy":
        from allrank.data.generate_dummy_data import generate_dummy_data
        np.random.seed(42)                                       # generate_dummy_data.py:31
        for role in ("train", "vali"):
            X, y, qid = generate_dummy_data(num_queries=100, results_len=20, num_labels=5, num_features=20)
            out[role] = (np.asarray(X, dtype=np.float32), np.asarray(y, dtype=np.float32), np.asarray(qid, dtype=np.int64))
    else:
        rng = np.random.default_rng(7)
        for role, n_q in (("train", 120), ("vali", 60)):
            lens = rng.integers(5, 31, n_q)
            X = rng.standard_normal((lens.sum(), 24)).astype(np.float32)
            y = rng.choice(5, size=lens.sum(), p=[0.5, 0.25, 0.15, 0.06, 0.04]).astype(np.float32)
            out[role] = (X, y, np.repeat(np.arange(1000, 1000 + n_q), lens).astype(np.int64))
    return out


def write_job_files(data, folder):
    """the libsvm text both sides train from: float32 feature values written as the doubles they are (exact round trip)"""
    from sklearn.datasets import dump_svmlight_file
    os.makedirs(folder, exist_ok=True)
    for role, (X, y, qid) in data.items():
        dump_svmlight_file(X.astype(np.float64), y.astype(np.float64), os.path.join(folder, "%s.txt" % role), query_id=qid)


def _run_reference(cfg, data, tmp):
    import allrank.main as M
    import allrank.training.train_utils as TU
    folder = os.path.join(tmp, "data")
    write_job_files(data, folder)
    cfg = json.loads(json.dumps(cfg))
    cfg["data"]["path"] = folder
    cfg_path = os.path.join(tmp, "cfg.json")
    with open(cfg_path, "w") as fh:
        json.dump(cfg, fh)
    rec = {"batches": [], "epochs": [], "weights": [], "init": None}
    cur = {"batches": [], "n_metric_calls": 0}
    orig_lb, orig_cm = TU.loss_batch, TU.compute_metrics

    def loss_batch(model, loss_func, xb, yb, indices, gradient_clipping_norm, opt=None):
        if opt is not None:
            if rec["init"] is None:
                rec["init"] = {k: v.detach().clone().numpy() for k, v in model.state_dict().items()}
            cur["batches"].append((float(yb[yb != -1].double().sum()), int(len(xb))))
        return orig_lb(model, loss_func, xb, yb, indices, gradient_clipping_norm, opt)

    def compute_metrics(metrics, model, dl, dev):
        out = orig_cm(metrics, model, dl, dev)
        cur["n_metric_calls"] += 1
        if cur["n_metric_calls"] % 2 == 1:                       # train_utils.py:99 (train), :107 (validation)
            cur["train_metrics"] = dict(out)
        else:
            cur["val_metrics"] = dict(out)
            rec["weights"].append({k: v.detach().clone().numpy() for k, v in model.state_dict().items()})
        return out

    orig_es = TU.epoch_summary

    def epoch_summary(epoch, train_loss, val_loss, train_metrics, val_metrics):
        rec["epochs"].append((float(train_loss), float(val_loss), dict(cur["train_metrics"]), dict(cur["val_metrics"])))
        rec["batches"].append(list(cur["batches"]))
        cur["batches"] = []
        return orig_es(epoch, train_loss, val_loss, train_metrics, val_metrics)

    TU.loss_batch, TU.compute_metrics, TU.epoch_summary = loss_batch, compute_metrics, epoch_summary
    old_argv = sys.argv
    sys.argv = ["allrank", "--job-dir", os.path.join(tmp, "job"), "--run-id", "traj", "--config-file-name", cfg_path]
    try:
        M.run()                                                  # main.py:34-110, the reference's own fit
    finally:
        sys.argv = old_argv
        TU.loss_batch, TU.compute_metrics, TU.epoch_summary = orig_lb, orig_cm, orig_es
    return rec


def build():
    import logging
    load_reference(stable_sort=True)
    out = {}
    jobs = _jobs()
    out["jobs"] = np.array([n for n, _, _ in jobs])
    for name, cfg, kind in jobs:
        data = _data(kind)
        with tempfile.TemporaryDirectory() as tmp:
            rec = _run_reference(cfg, data, tmp)
        for h in list(logging.getLogger("allrank").handlers):    # (init_logger adds a file handler per run)
            logging.getLogger("allrank").removeHandler(h)
        names = [m for m in cfg["metrics"]]
        out[name + "/config"] = np.array(json.dumps(cfg))
        for role, (X, y, qid) in data.items():
            out["%s/data/%s/X" % (name, role)], out["%s/data/%s/y" % (name, role)], out["%s/data/%s/qid" % (name, role)] = X, y, qid
        out[name + "/metric_names"] = np.array(names)
        out[name + "/train_loss"] = np.array([e[0] for e in rec["epochs"]], dtype=np.float64)
        out[name + "/val_loss"] = np.array([e[1] for e in rec["epochs"]], dtype=np.float64)
        out[name + "/train_metrics"] = np.array([[float(e[2][m]) for m in names] for e in rec["epochs"]], dtype=np.float64)
        out[name + "/val_metrics"] = np.array([[float(e[3][m]) for m in names] for e in rec["epochs"]], dtype=np.float64)
        out[name + "/batch_label_sums"] = np.array([[b[0] for b in ep] for ep in rec["batches"]], dtype=np.float64)
        out[name + "/batch_sizes"] = np.array([[b[1] for b in ep] for ep in rec["batches"]], dtype=np.int64)
        for k, v in rec["init"].items():
            out["%s/init/%s" % (name, k)] = v
        for e, wts in enumerate(rec["weights"]):
            for k, v in wts.items():
                out["%s/weights_epoch%d/%s" % (name, e, k)] = v
    return {"trajectory_golden.npz": out}


if __name__ == "__main__":
    for fname, arrays in build().items():
        np.savez_compressed(os.path.join(HERE, fname), **arrays)
        print("wrote", fname, len(arrays), "arrays", os.path.getsize(os.path.join(HERE, fname)), "bytes")
