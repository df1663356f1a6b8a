"""Same-box A/B of a FusedTrainer switch (default: group_wgrad -- the grouped weight-gradient launch; AB_OPT=relu_bits: the one-bit
ReLU mask) on the bench workload:
   [AB_OPT=name] python tools/wgrad_group_ab.py [slates ...]   -> ms/step with the switch on / off, interleaved, per batch size."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from allrank_amd.model import make_model
    from allrank_amd.engine import FusedTrainer
    sizes = [int(a) for a in sys.argv[1:]] or [64, 256]
    opt = os.environ.get("AB_OPT", "group_wgrad")  # group_wgrad | relu_bits | pad_input | all (the three together)
    dev = "cuda:0"
    L, F = 240, 136
    for B in sizes:
        rng = np.random.default_rng(0)
        x = torch.tensor(rng.standard_normal((B, L, F)).astype(np.float32), device=dev)
        y = torch.tensor(rng.integers(0, 5, (B, L)).astype(np.float32), device=dev)
        trs = {}
        for grouped in (True, False):
            torch.manual_seed(0)
            m = make_model(dict(sizes=[512], input_norm=False, activation=None, dropout=0.0),
                           dict(N=2, d_ff=2048, h=8, positional_encoding=None, dropout=0.0),
                           dict(d_output=1, output_activation=None), F).to(dev)
            trs[grouped] = FusedTrainer(m, "approxNDCGLoss", {}, B, L, lr=1e-3, **({k: grouped for k in ("group_wgrad", "relu_bits", "pad_input")} if opt == "all" else {opt: grouped}))
            for _ in range(5):
                trs[grouped].step(x, y)
        res = {True: [], False: []}
        for rep in range(5):
            for grouped in (True, False):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    trs[grouped].step(x, y)
                torch.cuda.synchronize()
                res[grouped].append((time.perf_counter() - t0) / 20 * 1e3)
        g, s = min(res[True]), min(res[False])
        print("   loss after the timed steps: on %.6f  off %.6f" % (trs[True].step(x, y).item(), trs[False].step(x, y).item()))
        print("slates %4d: %s on %.3f ms/step (%.2f M items/s)   off %.3f ms/step (%.2f M items/s)   gain %.1f %%"
              % (B, opt, g, B * L / g / 1e3, s, B * L / s / 1e3, (s / g - 1) * 100), flush=True)


if __name__ == "__main__":
    main()
