"""Is the 64-slate step limited by the host-side launch of its captured graph?  Times (a) the step as shipped (one captured graph
replayed every step), (b) two captures of the same step replayed alternately, (c) the host time of one replay() call, (d) eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from allrank_amd.model import make_model
from allrank_amd.engine import FusedTrainer
B, L, F = int(os.environ.get("SLATES", 64)), 240, 136
torch.manual_seed(0)
m = make_model(dict(sizes=[512], input_norm=False, activation=None, dropout=0.0), dict(N=2, d_ff=2048, h=8, positional_encoding=None, dropout=0.0),
               dict(d_output=1, output_activation=None), F).to("cuda")
ft = FusedTrainer(m, "approxNDCGLoss", {}, B, L, lr=1e-3)
rng = np.random.default_rng(0)
x = torch.tensor(rng.standard_normal((B, L, F)).astype(np.float32), device="cuda")
y = torch.tensor(rng.integers(0, 5, (B, L)).astype(np.float32), device="cuda")
for _ in range(6):
    ft.step(x, y)
torch.cuda.synchronize()


def timed(fn, n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, t_host * 1e3


a = min(timed(lambda: ft.step(x, y)) for _ in range(3))
segs_a = ft._graphs[next(iter(ft._graphs))]
segs_b = ft._capture()
flip = [0]


def pingpong():
    # (the same ingest launch step() makes, then the replay)
    ft.LB.check(ft.lib.ltrx_ingest_batch(ft.LB.ptr(x), ft.LB.ptr(y), x.numel(), ft.M, ft.x_in.shape[1], ft.x_in.stride(0), -1.0,
                                         ft.LB.ptr(ft.x_in), ft.LB.ptr(ft.y_in), ft.LB.ptr(ft.mask), ft._st()), "ingest")
    for g, after in (segs_a if flip[0] else segs_b):
        g.replay()
    flip[0] ^= 1


b = min(timed(pingpong) for _ in range(3))
g0 = segs_a[0][0]
c = min(timed(g0.replay) for _ in range(3))
ft.use_graph = False
d = min(timed(lambda: ft.step(x, y), 20) for _ in range(2))
print("slates %d: shipped step %.3f ms (host side of the loop %.3f ms/step) | two graphs alternating %.3f ms (host %.3f) | bare replay() %.3f ms (host %.3f) | eager %.3f ms (host %.3f)"
      % (B, a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]))
