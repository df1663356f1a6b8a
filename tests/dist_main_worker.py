"""one rank of test_gpu_main_sequence.py::test_main_call_sequence_under_the_launcher_equals_the_one_rank_run.

The call sequence of allrank/main.py:34-110 (see test_gpu_main_sequence.py: the reference tree does not travel to the GPU box) with
exactly the objects ``allrank_amd.install(fit=True)`` binds into an unmodified main.py under ``python -m allrank_amd.launch``:
``launch.setup()`` (GPU of the rank + process group), ``launch.get_torch_device`` (main.py:71), ``launch.create_data_loaders``
(main.py:66-67: global batch = world x batch_size, sampler on torch's global generator as in dataset_loading.py:245),
``launch.CustomDataParallel`` (main.py:76-78, taken when FORCE_WRAP) and ``allrank_amd.fit.fit`` (main.py:90).

    dist_main_worker.py OUT.pt BATCH_SIZE LOSS FORCE_WRAP [fc_only] [devloader]
``devloader``: the datasets / loaders are the ones install() binds to main.py:57-68 since round 6 (``allrank_amd.data.load_libsvm_dataset``
-> HBM-resident slates, ``launch.create_data_loaders`` -> DeviceLoader: every rank assembles only its block of each global batch).
Under the launcher's environment (2 ranks, gloo, both on GPU 0): the sharded run; rank 0 writes OUT.pt.  Without: the 1-rank run.
"""
import json
import os
import sys
import tempfile
import types
from functools import partial

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    out_path, batch_size, loss_name, force_wrap = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    fc_only = "fc_only" in sys.argv[5:]                              # BASELINE configs[0]: transformer null -> the slate-resident FC step
    devloader = "devloader" in sys.argv[5:]
    from torch import optim
    from allrank_amd import fit as EF, launch, losses
    from allrank_amd.model import make_model
    from tests.test_gpu_main_sequence import CONFIG, _dummy_libsvm, _load
    cfg = json.loads(json.dumps(CONFIG))
    cfg["training"]["epochs"] = 3
    if fc_only:
        cfg["model"]["transformer"] = None
    launch.setup()                                                     # install(fit=True) does this first
    try:
        torch.manual_seed(42)                                          # main.py:36-38
        torch.cuda.manual_seed_all(42)
        np.random.seed(42)
        rng = np.random.default_rng(42)
        with tempfile.TemporaryDirectory() as tmp:
            for role in ("train", "vali"):
                _dummy_libsvm(os.path.join(tmp, "%s.txt" % role), rng)
            if devloader:
                from allrank_amd import data as ED
                train_ds, val_ds = ED.load_libsvm_dataset(tmp, cfg["data"]["slate_length"], "vali")                     # main.py:57-61
                val_ds.slate_length = cfg["data"]["slate_length"]      # (the host twin `_load` pads the validation role to slate_length too)
                n_features = train_ds.shape[-1]
            else:
                train_ds = _load(os.path.join(tmp, "train.txt"), cfg["data"]["slate_length"])
                val_ds = _load(os.path.join(tmp, "vali.txt"), cfg["data"]["slate_length"])
                n_features = train_ds.tensors[0].shape[-1]
            train_dl, val_dl = launch.create_data_loaders(train_ds, val_ds, num_workers=0, batch_size=batch_size)      # main.py:66-67
            if devloader:
                assert isinstance(train_dl, ED.DeviceLoader) and (train_dl.rank, train_dl.world) == (launch.rank(), launch.world_size())
            dev = launch.get_torch_device()                            # main.py:71
            model = make_model(n_features=n_features, **json.loads(json.dumps(cfg["model"])))
            if force_wrap:                                             # main.py:76-78 on a node with several visible GPUs
                model = launch.CustomDataParallel(model)
            model.to(dev)
            optimizer = getattr(optim, cfg["optimizer"]["name"])(params=model.parameters(), **cfg["optimizer"]["args"])
            loss_func = partial(getattr(losses, loss_name))
            scheduler = getattr(optim.lr_scheduler, cfg["lr_scheduler"]["name"])(optimizer, **cfg["lr_scheduler"]["args"])
            config = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric=cfg["val_metric"])
            epoch_losses, orig = [], EF.log.info

            def spy(msg, *a):
                if isinstance(msg, str) and msg.startswith("Epoch :"):
                    epoch_losses.append((float(a[1]), float(a[2])))
                return orig(msg, *a)
            EF.log.info = spy
            try:
                result = EF.fit(model=model, loss_func=loss_func, optimizer=optimizer, scheduler=scheduler, train_dl=train_dl,
                                valid_dl=val_dl, config=config, device=dev, output_dir=tmp, tensorboard_output_path=None,
                                **cfg["training"])
            finally:
                EF.log.info = orig
            assert EF.last_run["engine"] == "fused" and bool(EF.last_run["fcstep"]) == fc_only, EF.last_run
            saved = torch.load(os.path.join(tmp, "model.pkl"), map_location="cpu") if launch.rank() == 0 else None
        if launch.rank() == 0:
            torch.save(dict(losses=epoch_losses, weights=saved, world=launch.world_size(), device=str(dev),
                            train_batch=int(train_dl.batch_size), val={k: float(v) for k, v in result["val_metrics"].items()},
                            train={k: float(v) for k, v in result["train_metrics"].items()}), out_path)
        print("MAIN_WORKER_OK rank %d world %d" % (launch.rank(), launch.world_size()))
    finally:
        launch.shutdown()


if __name__ == "__main__":
    main()
