"""one rank of tests/test_launch_cpu.py::test_unmodified_main_under_the_launcher_*: the body of ``python -m allrank_amd.launch``
(``launch.run_main``: setup -> install(fit=True) -> allrank.main.run(), the reference's main.py untouched) with ``fit`` replaced by
a probe that records what main.py handed over.  No GPU here, so the probe does not train; it checks the plumbing a multi-GPU run
stands on: this rank's device, the process group, no DataParallel wrapper although ``device_count() > 1`` (main.py:76-78), the
global batch = world x batch_size (dataset_loading.py:240-241), identical global batches on every rank for two epochs.

    launch_main_worker.py OUT_DIR FORCE_DEVICE_COUNT -- <main.py args>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    out_dir, force_count = sys.argv[1], int(sys.argv[2])
    main_args = sys.argv[sys.argv.index("--") + 1:]
    from oracle.ref_loader import load_reference
    load_reference(stable_sort=False)
    if force_count:
        torch.cuda.device_count = lambda: force_count          # main.py:76 takes the multi-GPU branch (no GPU in this container)
    import torch.distributed as dist
    from allrank_amd import fit as EF, launch, parallel
    seen = {}

    def probe(epochs, model, loss_func, optimizer, scheduler, train_dl, valid_dl, config, gradient_clipping_norm, early_stopping_patience,
              device, output_dir, tensorboard_output_path, **ext):
        import allrank.main as M
        RN = sys.modules["allrank.models.losses.neuralNDCG"]    # (the package re-exports a function of the same name)
        import allrank.models.model_utils as MU
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        spec, reason = EF._fused_spec(model, loss_func, optimizer)
        sums = []
        for _ in range(2):
            for i, (xb, yb, idx) in enumerate(train_dl):
                if world > 1 and i == 0:
                    EF._check_same_batch(xb, yb, world)          # the collective fit() itself runs on the first batch
                sums.append([int(xb.shape[0]), float(xb.double().sum()), float(yb.double().sum())])
        seen.update(rank=rank, world=world, backend=dist.get_backend() if world > 1 else None, device=str(device),
                    main_device=str(M.get_torch_device()), loss_module_device=str(RN.get_torch_device()),
                    utils_device=str(MU.get_torch_device()), wrapped=isinstance(model, torch.nn.DataParallel),
                    model_type=type(model).__name__, param_device=str(next(model.parameters()).device),
                    train_batch=int(train_dl.batch_size), val_batch=int(valid_dl.batch_size), sums=sums, reason=reason,
                    fusable=spec is not None, shard=list(parallel.shard_slates(int(train_dl.batch_size), rank, world)),
                    output_dir=output_dir, wrapper_is_identity=M.CustomDataParallel is launch.CustomDataParallel,
                    loaders_rebound=M.create_data_loaders is launch.create_data_loaders)
        return EF.make_result(epochs - 1, {"ndcg_5": 0.5}, {"ndcg_5": 0.5}, sum(p.numel() for p in model.parameters()))

    EF.fit = probe
    try:
        launch.run_main(main_args)
        with open(os.path.join(out_dir, "rank%d.json" % seen["rank"]), "w") as fh:
            json.dump(seen, fh)
    finally:
        launch.shutdown()


if __name__ == "__main__":
    main()
