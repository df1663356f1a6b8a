"""host-batch training loop at config 3: the reference's per-step pageable `.to(device)` (train_utils.py:95) vs allrank_amd.fit._Prefetcher
(pinned staging + copy stream, one batch ahead), both feeding FusedTrainer.step.  usage (GPU box): python tools/fit_h2d_timing.py"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as BN
from allrank_amd.engine import FusedTrainer
from allrank_amd.fit import _Prefetcher
dev = torch.device("cuda:0")
w = BN.WORKLOADS["attn_approxndcg"]
B, L, nb = 256, 240, 12
model = BN.build_model(w, dev)
tr = FusedTrainer(model, w["loss"], {}, B, L, lr=1e-3, use_graph=True)
x, y, idx = BN.synth_batch(nb * B, L, w["n_features"], 1, dev)
host = [(x[i * B:(i + 1) * B].cpu(), y[i * B:(i + 1) * B].cpu(), idx[i * B:(i + 1) * B].cpu()) for i in range(nb)]
for i in range(4):
    tr.step(x[:B], y[:B], idx[:B])
torch.cuda.synchronize()


def run(kind):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "resident":
        for i in range(nb):
            tr.step(x[i * B:(i + 1) * B], y[i * B:(i + 1) * B], idx[i * B:(i + 1) * B])
    elif kind == "to_device":
        for xb, yb, ib in host:
            tr.step(xb.to(dev), yb.to(dev), ib.to(dev))
    else:
        for xb, yb, ib in _Prefetcher(host, dev):
            tr.step(xb, yb, ib)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / nb * 1e3


out = {}
for kind in ("resident", "to_device", "prefetcher", "resident", "to_device", "prefetcher"):
    out.setdefault(kind, []).append(round(run(kind), 3))
print(json.dumps({"ms_per_step": out, "items_per_step": B * L}))
