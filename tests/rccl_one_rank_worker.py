"""RCCL on the ONE GPU a build box has: ``init_process_group("nccl", world_size=1, device_id=cuda:0)`` and the
SHARDED form of the explicit step on that one-rank group -- ``FusedTrainer(force_dist=True)``: global divisor through shard_context,
bucketed ``all_reduce`` of the flat gradient behind the backward, the one-float normaliser all-reduce of neuralNDCG / lambdaLoss(mean)
between the loss kernel's phases, hipGraph segments cut at every collective -- against the NON-distributed step: a one-rank sum is the
identity, so losses, gradients and weights must be IDENTICAL bit for bit, step after step, captured and eager.  What the 8-GPU
node adds to this is peers, not code paths (replaces allrank/main.py:76-78, models/model_utils.py:40-53).

    rccl_one_rank_worker.py OUT.json
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allrank_amd.engine import FusedTrainer  # noqa: E402
from allrank_amd.model import make_model  # noqa: E402


def build(seed=7):
    torch.manual_seed(seed)
    return make_model(dict(sizes=[64], input_norm=False, activation=None, dropout=0.0),
                      dict(N=2, d_ff=128, h=2, positional_encoding=None, dropout=0.0),
                      dict(d_output=1, output_activation=None), 24).to("cuda:0")


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.time()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    rec = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": round(time.time() - t0, 2)}
    # pre-flight: a checked all-reduce, an async one, an all-gather (what fit() / bench.py issue besides the step's collectives)
    t = torch.arange(1024, device=dev, dtype=torch.float32)
    dist.all_reduce(t)
    w = dist.all_reduce(t, async_op=True)
    w.wait()
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(1024, dtype=torch.float32)), "one-rank all_reduce(SUM) must be the identity"
    parts = [torch.empty(3, device=dev, dtype=torch.float64)]
    dist.all_gather(parts, torch.tensor([1.0, 2.0, 3.0], device=dev, dtype=torch.float64))
    assert parts[0].tolist() == [1.0, 2.0, 3.0]
    B, L = 8, 40
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.standard_normal((B, L, 24)).astype(np.float32), device=dev)
    y = torch.tensor(rng.integers(0, 5, (B, L)).astype(np.float32), device=dev)
    y[5, 30:] = -1
    y[2, :] = 0                                                   # idcg == 0: outside neuralNDCG's normaliser count
    rec["jobs"] = []
    for loss_name, args in (("approxNDCGLoss", {}), ("neuralNDCG", {}),
                            ("lambdaLoss", dict(weighing_scheme="lambdaRank_scheme", reduction="mean")), ("listNet", {})):
        plain = FusedTrainer(build(), loss_name, args, B, L, lr=1e-3, use_graph=True)
        cap = FusedTrainer(build(), loss_name, args, B, L, lr=1e-3, world_size=1, use_graph=True, force_dist=True)
        eag = FusedTrainer(build(), loss_name, args, B, L, lr=1e-3, world_size=1, use_graph=False, force_dist=True)
        assert cap.sharded and eag.sharded and not plain.sharded
        for step in range(6):                                     # two eager warm-ups, the capture step, three replays
            lp, lc, le = (tr.step(x, y).clone() for tr in (plain, cap, eag))
            assert torch.equal(lp, lc) and torch.equal(lp, le), (loss_name, step, "loss", lp.item(), lc.item(), le.item())
            assert torch.equal(plain.flat_g, cap.flat_g) and torch.equal(plain.flat_g, eag.flat_g), (loss_name, step, "gradients")
            assert torch.equal(plain.flat_p, cap.flat_p) and torch.equal(plain.flat_p, eag.flat_p), (loss_name, step, "weights")
        assert cap.capture_fallback is None, (loss_name, cap.capture_fallback)
        segs = cap.graph
        assert segs is not None and len(segs) >= 1 + len(cap._buckets), (loss_name, None if segs is None else len(segs))
        assert plain.graph is not None and len(plain.graph) == 1
        rec["jobs"].append({"loss": loss_name, "graph_segments": len(segs), "buckets": len(cap._buckets), "steps": 6,
                            "loss_value": float(lc.item()), "bit_identical_to_the_non_distributed_step": True})
    # a short last batch: its own divisor -> its own capture, collectives included
    cap = FusedTrainer(build(), "neuralNDCG", {}, B, L, lr=1e-3, world_size=1, use_graph=True, force_dist=True)
    plain = FusedTrainer(build(), "neuralNDCG", {}, B, L, lr=1e-3, use_graph=True)
    ys = y.clone()
    ys[5:] = -1
    xs = x.clone()
    xs[5:] = 0
    for step in range(8):
        full = step % 2 == 0
        a = (x, y, B) if full else (xs, ys, 5)
        lc, lp = cap.step(a[0], a[1], global_batch=a[2]).clone(), plain.step(a[0], a[1], global_batch=a[2]).clone()
        assert torch.equal(lc, lp) and torch.equal(cap.flat_p, plain.flat_p), ("short batch", step)
    assert len(cap._graphs) == 2 and cap.capture_fallback is None
    rec["short_batch_captures"] = len(cap._graphs)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    with open(out_path, "w") as fh:
        json.dump(rec, fh)
    print("RCCL_ONE_RANK_OK", json.dumps(rec))


if __name__ == "__main__":
    main()
