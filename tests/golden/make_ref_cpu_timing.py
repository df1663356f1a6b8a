"""Time the REAL reference training step (allegro/allRank at /root/reference) on this container's CPU cores.

Run in the build container only (the reference does not travel to the GPU box):
    python tests/golden/make_ref_cpu_timing.py
Writes tests/golden/ref_cpu_timing.json, which bench.py prints beside its live numpy-port number as
``cpu_baseline.reference_build_box``.  What is timed is the body of ``loss_batch`` (allrank/training/train_utils.py:
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
    load_reference(stable_sort=False)
    from allrank.models import losses as RL
    from allrank.models.model import make_model
    from allrank.config import TransformerConfig
    from allrank.training.train_utils import loss_batch

    cores = os.cpu_count()
    torch.set_num_threads(cores)
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
    for B in (16, 64):
        tr = TransformerConfig(N=2, d_ff=2048, h=8, positional_encoding=None, dropout=0.0)
        model = make_model(fc_model=dict(sizes=[512], input_norm=False, activation=None, dropout=0.0), transformer=tr,
                           post_model=dict(d_output=1, output_activation=None), n_features=F)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(7)
        xb = torch.randn((B, L, F), generator=g)
        yb = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * L, replacement=True, generator=g).view(B, L).float()
        idx = torch.arange(L).expand(B, L).contiguous()
        loss_batch(model, RL.approxNDCGLoss, xb, yb, idx, None, opt)       # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            loss_batch(model, RL.approxNDCGLoss, xb, yb, idx, None, opt)
            n += 1
            el = time.perf_counter() - t0
            if el > 20.0 or n >= 40:
                break
        out["points"].append(dict(slates=B, steps=n, seconds=round(el, 3), items_per_s=round(n * B * L / el, 1)))
        print(out["points"][-1], flush=True)
    out["value"] = max(p["items_per_s"] for p in out["points"])
    out["unit"] = "slate-items/s"
    with open(os.path.join(HERE, "ref_cpu_timing.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
