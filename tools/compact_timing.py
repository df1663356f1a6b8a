"""host vs device time of the compact (variable-length) step on the --ragged bench batch.  usage (GPU box):
python tools/compact_timing.py [mode ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from allrank_amd.engine import FusedTrainer

w = bench.WORKLOADS["attn_approxndcg"]
B, L = 256, 240
dev = torch.device("cuda", 0)
x, y, idx = bench.synth_batch(8 * B, L, w["n_features"], 42, dev, ragged=True)
lens = (y != -1).sum(1).cpu()
MODES = sys.argv[1:] or ["padded_graph", "padded_eager", "compact_hostlens", "compact_devcount"]
for mode in MODES:
    model = bench.build_model(w, dev, 0.0)
    tr = FusedTrainer(model, w["loss"], {}, B, L, lr=1e-3, use_graph=(mode == "padded_graph"), compact=mode.startswith("compact"))

    def step(i):
        j = (i % 8) * B
        if mode == "compact_hostlens":
            return tr.step(x[j:j + B], y[j:j + B], lengths=lens[j:j + B])
        return tr.step(x[j:j + B], y[j:j + B])
    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for i in range(20):
        a = time.perf_counter()
        step(i)
        host += time.perf_counter() - a
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    # device time of one step in isolation
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    step(0)
    e1.record()
    torch.cuda.synchronize()
    print("%-18s wall/step %.3f ms   host-in-step %.3f ms   one isolated step (events) %.3f ms   rows %d" %
          (mode, tot / 20 * 1e3, host / 20 * 1e3, e0.elapsed_time(e1), tr.rows))
