"""An UNMODIFIED allrank/main.py driven through ``allrank_amd.install(fit=True)`` (VERDICT r2 missing #4).

main.py:34-110 parses --job-dir / --run-id / --config-file-name, loads the libsvm data, builds the model through ``make_model``, the
loss through ``getattr(losses, name)`` + functools.partial, torch.optim.Adam and the scheduler, calls ``fit(...)`` and finally
``dump_experiment_result`` / ``assert_expected_metrics`` on what fit returned.  The reference tree cannot travel to the GPU box
(it must not be copied into this repository) and this container has no GPU, so the run is covered in two halves:

  * here (no GPU, reference present): the whole of main.run() with the REAL install(fit=True) rebinding, the real data pipeline of
    the reference on generate_dummy_data output, our make_model / losses / metrics -- and our fit() replaced, at the last moment,
    by a probe that checks what main.py hands over is exactly what the explicit step takes (``_fused_spec`` accepts it: an
    allrank_amd LTRModel, a partial of an allrank_amd loss, a default Adam) and returns ``make_result`` values, which then go through
    the reference's own dump_experiment_result / assert_expected_metrics;
  * on a machine that has BOTH a GPU and a checkout (ALLRANK_REFERENCE=/path/to/allRank): the same with the real fit -- the
    ``-m gpu`` test below (skipped on the GPU box of this build, where the reference is absent).

Several GPUs (round 5): the same unmodified main.run() as TWO ranks under ``python -m allrank_amd.launch`` -- process group, rank devices,
no DataParallel wrapper, global batch = world x batch_size, identical batches on both ranks, rank 0 owning the job directory -- is
tests/test_launch_cpu.py (CPU half, gloo); its in-tree GPU twin (training under the launcher == the one-rank run) is
tests/test_gpu_main_sequence.py::test_main_call_sequence_under_the_launcher_equals_the_one_rank_run.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

from oracle.ref_loader import reference_available, REFERENCE_ROOT

pytestmark = pytest.mark.skipif(not reference_available(), reason="needs a checkout of allegro/allRank (ALLRANK_REFERENCE)")

# scripts/local_config.json of the reference (the run_example.sh job), edited as SURVEY 8(d) config (1) prescribes: slate length of
# the dummy data, ListNet on a d_output = 1 head; the 1-layer transformer of the example is kept
CONFIG = {
    "model": {"fc_model": {"sizes": [64], "input_norm": False, "activation": None, "dropout": 0.0},
              "transformer": {"N": 1, "d_ff": 64, "h": 1, "positional_encoding": None, "dropout": 0.0},
              "post_model": {"output_activation": None, "d_output": 1}},
    "data": {"path": None, "validation_ds_role": "vali", "num_workers": 0, "batch_size": 32, "slate_length": 24},
    "optimizer": {"name": "Adam", "args": {"lr": 0.001}},
    "lr_scheduler": {"name": "StepLR", "args": {"step_size": 3, "gamma": 0.5}},
    "training": {"epochs": 3, "early_stopping_patience": 100, "gradient_clipping_norm": None},
    "val_metric": "ndcg_5", "metrics": ["ndcg_5"],
    "loss": {"name": "listNet", "args": {}},
    "expected_metrics": {"val": {"ndcg_5": 0.3}},
}


def _prepare(tmp_path):
    """dummy libsvm data exactly as allrank/data/generate_dummy_data.py:31-42 writes it (100 queries x 20 docs x 20 features,
    seed 42), a config file, and the argv main.parse_args() reads"""
    from oracle.ref_loader import load_reference
    load_reference(stable_sort=False)
    from allrank.data.generate_dummy_data import generate_dummy_data
    from sklearn.datasets import dump_svmlight_file
    np.random.seed(42)
    data = tmp_path / "dummy_data"
    data.mkdir()
    for role in ("train", "vali"):
        X, y, qid = generate_dummy_data(num_queries=100, results_len=20, num_labels=5, num_features=20)
        dump_svmlight_file(X, y, str(data / ("%s.txt" % role)), query_id=qid)
    cfg = json.loads(json.dumps(CONFIG))
    cfg["data"]["path"] = str(data)
    (tmp_path / "cfg.json").write_text(json.dumps(cfg))
    return ["allrank", "--job-dir", str(tmp_path / "job"), "--run-id", "r3", "--config-file-name", str(tmp_path / "cfg.json")]


def test_unmodified_main_reaches_the_explicit_step_with_a_fusable_job(tmp_path, monkeypatch):
    argv = _prepare(tmp_path)
    import allrank_amd
    from allrank_amd import fit as EF, losses as E
    from allrank_amd.model import LTRModel
    seen = {}

    def probe(epochs, model, loss_func, optimizer, scheduler, train_dl, valid_dl, config, gradient_clipping_norm, early_stopping_patience,
              device, output_dir, tensorboard_output_path, **ext):
        spec, reason = EF._fused_spec(model, loss_func, optimizer)
        seen.update(spec=spec, reason=reason, model=model, epochs=epochs, n_batches=len(train_dl), ext=ext)
        xb, yb, idx = next(iter(train_dl))
        seen.update(xb=tuple(xb.shape), yb=tuple(yb.shape), idx=tuple(idx.shape), idx_dtype=idx.dtype,
                    batch_shape=EF._batch_shape(train_dl)[:2])
        return EF.make_result(epochs - 1, {"ndcg_5": 0.5}, {"ndcg_5": 0.5}, sum(p.numel() for p in model.parameters()))

    monkeypatch.setattr(EF, "fit", probe)
    monkeypatch.setattr(sys, "argv", argv)
    done = allrank_amd.install(fit=True)
    try:
        import importlib
        main = importlib.import_module("allrank.main")
        allrank_amd.install(fit=True)                     # (main imported after the first install: rebind its imported names too)
        assert main.fit is probe and main.make_model is allrank_amd.model.make_model
        main.run()                                        # main.py:34-110, untouched
    finally:
        allrank_amd.uninstall()
    assert "allrank.training.train_utils.fit" in done
    assert isinstance(seen["model"], LTRModel) and seen["reason"] == ""
    # what FusedTrainer(model, loss_name, loss_args, ..., lr=, **optimizer kwargs) is built from
    assert seen["spec"] == ("listNet", {}, 0.001, dict(optimizer="Adam", betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0))
    assert seen["epochs"] == 3 and seen["n_batches"] == 4 and seen["ext"] == {}
    assert seen["xb"] == (32, 24, 20) and seen["yb"] == (32, 24) and seen["idx"] == (32, 24) and seen["idx_dtype"] == torch.int64
    assert seen["batch_shape"] == (32, 24)                # the shapes the static step is built for, without consuming a batch
    out = json.load(open(os.path.join(str(tmp_path / "job"), "results", "r3", "experiment_result.json")))
    assert out["val_metrics/ndcg_5"] == 0.5 and out["num_params"] == sum(p.numel() for p in seen["model"].parameters())
    assert E.listNet.__name__ == "listNet"


@pytest.mark.gpu
def test_unmodified_main_trains_on_the_fused_step(tmp_path, monkeypatch):
    """needs a GPU AND a reference checkout (ALLRANK_REFERENCE): `allrank_amd.install(fit=True); allrank.main.run()`"""
    argv = _prepare(tmp_path)
    import allrank_amd
    from allrank_amd import fit as EF
    monkeypatch.setattr(sys, "argv", argv)
    allrank_amd.install(fit=True)
    try:
        import importlib
        main = importlib.import_module("allrank.main")
        allrank_amd.install(fit=True)
        main.run()
    finally:
        allrank_amd.uninstall()
    assert EF.last_run["engine"] == "fused", EF.last_run
    out = json.load(open(os.path.join(str(tmp_path / "job"), "results", "r3", "experiment_result.json")))
    assert 0.3 <= out["val_metrics/ndcg_5"] <= 1.0 and np.isfinite(out["train_metrics/ndcg_5"])
    assert os.path.exists(os.path.join(str(tmp_path / "job"), "results", "r3", "model.pkl"))
