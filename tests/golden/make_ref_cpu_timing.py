"""Time the REAL reference training step (allegro/allRank at /root/reference) on this container's CPU cores.

Run where a checkout of the reference is present (the build container: /root/reference; anywhere else: ALLRANK_REFERENCE=<path>):
    python tests/golden/make_ref_cpu_timing.py [--threads 16,64] [--out FILE]
Writes tests/golden/ref_cpu_timing.json (or --out), which bench.py prints beside its live port number as
``cpu_baseline.reference_build_box`` / ``reference_gpu_box`` (round 4: one metered run on the GPU box's host cores with the reference
staged in an ignored scratch directory, tests/golden/ref_cpu_timing_gpubox.json).  What is timed is the body of ``loss_batch`` (allrank/training/train_utils.py:
18-29) exactly as the reference runs it: ``make_model`` of allrank/models/model.py:131-151, ``approxNDCGLoss``,
``loss.backward()``, ``torch.optim.Adam.step()``, ``zero_grad()``, ``loss.item()`` -- on CPU tensors, torch threads = all
cores, BASELINE.json configs[2] (F=136, slate 240, fc[512] + 2x self-attention d512 h8 d_ff2048, dropout 0).
"""
import json
import os
import platform
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="", help="comma-separated torch thread counts to try (default: all cores)")
    ap.add_argument("--out", default=os.path.join(HERE, "ref_cpu_timing.json"))
    ap.add_argument("--device", default="cpu", help="cpu (default) or cuda: the SAME unmodified reference step on the GPU through stock "
                                                      "PyTorch-ROCm (the 'reference on this hardware' figure)")
    ap.add_argument("--slates", default="16,64")
    args = ap.parse_args()
    load_reference(stable_sort=False)
    from allrank.models import losses as RL
    from allrank.models.model import make_model
    from allrank.config import TransformerConfig
    from allrank.training.train_utils import loss_batch

    cores = os.cpu_count()
    thread_counts = [int(t) for t in args.threads.split(",") if t] or [cores]
    torch.manual_seed(42)
    L, F = 240, 136
    out = dict(config="BASELINE.json configs[2]: F=136 L=240 fc[512] + 2x self-attention(d512,h8,d_ff2048) + ApproxNDCG, Adam 1e-3, dropout 0",
               cores=cores, torch=torch.__version__, cpu=platform.processor() or platform.machine(), points=[])
    try:
        with open("/proc/cpuinfo") as fh:
            names = [l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")]
        out["cpu"] = names[0] if names else out["cpu"]
    except OSError:
        pass
    dev = torch.device(args.device)
    out["device"] = args.device if dev.type == "cpu" else torch.cuda.get_device_name(0)
    for B, nthr in [(b_, t_) for t_ in thread_counts for b_ in [int(v) for v in args.slates.split(",")]]:
        torch.set_num_threads(nthr)
        tr = TransformerConfig(N=2, d_ff=2048, h=8, positional_encoding=None, dropout=0.0)
        model = make_model(fc_model=dict(sizes=[512], input_norm=False, activation=None, dropout=0.0), transformer=tr,
                           post_model=dict(d_output=1, output_activation=None), n_features=F)
        model.to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(7)
        xb = torch.randn((B, L, F), generator=g)
        yb = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * L, replacement=True, generator=g).view(B, L).float()
        idx = torch.arange(L).expand(B, L).contiguous()
        xb, yb, idx = xb.to(dev), yb.to(dev), idx.to(dev)
        for _ in range(3 if dev.type == "cuda" else 1):
            loss_batch(model, RL.approxNDCGLoss, xb, yb, idx, None, opt)   # warm-up (loss.item() inside synchronises every step)
        n, t0 = 0, time.perf_counter()
        while True:
            loss_batch(model, RL.approxNDCGLoss, xb, yb, idx, None, opt)
            n += 1
            el = time.perf_counter() - t0
            if el > 20.0 or n >= 40:
                break
        out["points"].append(dict(slates=B, threads=nthr, steps=n, seconds=round(el, 3), items_per_s=round(n * B * L / el, 1)))
        print(out["points"][-1], flush=True)
    out["value"] = max(p["items_per_s"] for p in out["points"])
    out["unit"] = "slate-items/s"
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
