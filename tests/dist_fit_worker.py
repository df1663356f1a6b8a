"""worker of test_sharded_fit_with_uneven_last_batch_equals_one_rank (2 ranks on one GPU, gloo): the drop-in epoch loop
``allrank_amd.fit.fit`` (reference signature, allrank/training/train_utils.py:78-147) on 33 slates with a global batch of 16 --
batches of 16 / 16 / 1 slates, i.e. the last batch gives rank 0 one slate and rank 1 none (DataLoader drop_last=False,
allrank/data/dataset_loading.py:245); the validation pass is sharded the same way (17 slates: 16 + 1, round 5) -- against the same loop on one rank (``--ref``: a plain single process, run first by the test):
same per-epoch training loss (the reference's loss on the gathered batch) and the same trained weights."""
import os
import sys
import tempfile
import types
from functools import partial

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allrank_amd import losses as E  # noqa: E402
from allrank_amd import fit as FIT  # noqa: E402
from allrank_amd.model import make_model  # noqa: E402


def data(n, L, F, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, L, F)).astype(np.float32)
    w = rng.standard_normal(F).astype(np.float32)
    y = np.clip(np.round((x @ w) / np.sqrt(F) + 1.5), 0, 4).astype(np.float32)
    idx = np.tile(np.arange(L, dtype=np.int64), (n, 1))
    for b in range(n):
        k = int(rng.integers(L // 3, L + 1))
        y[b, k:] = -1
        x[b, k:] = 0
        idx[b, k:] = -1
    return torch.tensor(x), torch.tensor(y), torch.tensor(idx)


def build(F, variant="plain"):
    torch.manual_seed(7)
    pe = None
    if variant == "options":           # the options of the shipped configs around the encoder: learned positional encoding fed with `indices`
        pe = dict(strategy="learned", max_indices=40)            # (positional.py:40-77), input_norm, ReLU FC stack, Sigmoid output
        return make_model(dict(sizes=[48, 32], input_norm=True, activation="ReLU", dropout=0.0),
                          dict(N=2, d_ff=64, h=2, positional_encoding=pe, dropout=0.0),
                          dict(d_output=1, output_activation="Sigmoid"), F).to("cuda:0")
    return make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                      dict(N=1, d_ff=64, h=4, positional_encoding=None, dropout=0.0),
                      dict(d_output=1, output_activation=None), F).to("cuda:0")


def run_fit(model, loss_name, tr, va, epochs, tmp, use_fused=True, variant="plain"):
    from torch.utils.data import DataLoader, TensorDataset
    train_dl = DataLoader(TensorDataset(*tr), batch_size=16, shuffle=False)
    valid_dl = DataLoader(TensorDataset(*va), batch_size=16, shuffle=False)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched, clip, loss_kw = None, None, {}
    if variant == "options":           # AdamW + gradient clipping (train_utils.py:24-25) + ReduceLROnPlateau (train_utils.py:117-122) + a loss
        opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.01)      # with a batch-global normaliser (lambdaLoss mean)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="max", factor=0.5, patience=0)
        clip, loss_kw = 0.05, dict(weighing_scheme="lambdaRank_scheme", reduction="mean")
    cfg = types.SimpleNamespace(metrics={"ndcg": [5], "mrr": [3]} if variant == "options" else {"ndcg": [5]}, val_metric="ndcg_5")
    losses = []
    orig = FIT.log.info

    def spy(msg, *a):
        if isinstance(msg, str) and msg.startswith("Epoch :"):
            losses.append(float(a[1]))
        return orig(msg, *a)
    FIT.log.info = spy
    try:
        res = FIT.fit(epochs, model, partial(getattr(E, loss_name), **loss_kw), opt, sched, train_dl, valid_dl, cfg, clip, 100, "cuda:0", tmp, None,
                      use_fused=use_fused)
    finally:
        FIT.log.info = orig
    return losses, res


def main():
    """``--ref FILE``: one process, no process group: fit() on one rank, results saved to FILE.
    ``--cmp FILE`` (under torchrun, 2 ranks): the sharded fit(), compared on rank 0 with FILE."""
    import datetime
    mode, path = sys.argv[1], sys.argv[2]
    torch.cuda.set_device(0)
    L, F = 30, 20
    tr, va = data(33, L, F, 1), data(17, L, F, 2)     # validation: batches of 16 / 1 -> the sharded pass gives rank 1 an EMPTY block of the last one
    # (loss, fused step?): the last job runs the nn.Module + autograd Trainer (what fit() falls back to for a job the explicit step does not
    # cover) -- sharded training through parallel.FlatGradients and the sharded validation pass through the module forward
    jobs = (("approxNDCGLoss", True, "plain"), ("neuralNDCG", True, "plain"), ("listNet", False, "plain"), ("lambdaLoss", True, "options"))
    if mode == "--ref":
        out = {}
        for loss_name, fused, variant in jobs:
            m1 = build(F, variant)
            with tempfile.TemporaryDirectory() as tmp:
                l1, r1 = run_fit(m1, loss_name, tr, va, 3 if variant == "options" else 2, tmp, fused, variant)
            assert FIT.last_run["engine"] == ("fused" if fused else "autograd"), FIT.last_run
            out[loss_name] = (l1, {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()},
                              {k: float(v) for k, v in r1["val_metrics"].items()})
        torch.save(out, path)
        print("FIT_REF_OK")
        return
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        ref = torch.load(path) if rank == 0 else None
        for loss_name, fused, variant in jobs:
            m2 = build(F, variant)
            with tempfile.TemporaryDirectory() as tmp:
                l2, r2 = run_fit(m2, loss_name, tr, va, 3 if variant == "options" else 2, tmp, fused, variant)
            assert FIT.last_run["engine"] == ("fused" if fused else "autograd"), FIT.last_run
            if rank == 0:
                l1, w1, v1 = ref[loss_name]
                for a, b in zip(l1, l2):
                    assert abs(a - b) <= 1e-5 * (1 + abs(a)), (loss_name, "train loss per epoch", l1, l2)
                sd = {k: v.detach().cpu() for k, v in m2.state_dict().items()}
                werr = max(float((w1[k] - v).abs().max()) for k, v in sd.items())
                # two epochs of lr = 1e-3 Adam steps; entries whose gradient is below its round-off may take opposite signs
                assert werr <= (9 * 2.1e-3 * 2 if variant == "options" else 6 * 2.1e-3), (loss_name, "weights", werr)
                # (entries whose gradient is below its round-off -- e.g. the key biases, which softmax cancels -- move by lr * sign(noise)
                #  per step in ANY arithmetic: the bulk of the weights must agree closely, not every entry)
                n_ok = sum(int(((w1[k] - v).abs() <= 5e-5).sum()) for k, v in sd.items())
                n_all = sum(v.numel() for v in sd.values())
                assert n_ok >= 0.9 * n_all, (loss_name, "fraction of weights that agree to 5e-5", n_ok / n_all)
                for k in v1:
                    assert abs(v1[k] - float(r2["val_metrics"][k])) <= 2e-3, (loss_name, k, v1, r2["val_metrics"])
        dist.barrier()
        if rank == 0:
            print("FIT_EQUIV_OK")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
