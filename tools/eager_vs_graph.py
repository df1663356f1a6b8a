"""explicit step with and without hipGraph capture: GPU ms/step and host enqueue cost (the multi-GPU path runs eager)"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from allrank_amd.engine import FusedTrainer
w = bench.WORKLOADS["attn_approxndcg"]
dev = torch.device("cuda", 0)
for B in (256, 64):
    x, y, idx = bench.synth_batch(B, 240, 136, 1, dev)
    for graph in (True, False):
        m = bench.build_model(w, dev)
        t = FusedTrainer(m, w["loss"], {}, B, 240, use_graph=graph)
        for _ in range(5): t.step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): t.step(x, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        # host-only time: enqueue without sync
        t0 = time.perf_counter()
        for _ in range(20): t.step(x, y)
        host = (time.perf_counter() - t0) / 20
        torch.cuda.synchronize()
        print("B", B, "graph", graph, "ms/step %.3f" % (dt * 1e3), "host enqueue ms %.3f" % (host * 1e3), flush=True)
