import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import model_oracle as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_parity import _make_engine_model, _t
from allrank_amd.engine import FusedTrainer
cfg = dict(n_features=20, fc_sizes=[32], fc_activation=None, fc_input_norm=False, N=2, d_ff=64, h=4, output_activation=None)
rng = np.random.default_rng(12)
B, L = 4, 70
x = rng.standard_normal((B, L, 20)).astype(np.float32)
y = rng.integers(0, 5, (B, L)).astype(np.float32)
y[2, 40:] = -1
x[2, 40:] = 0
xt, yt = _t(x), _t(y)
for name, kw, hack in [("default", {}, None), ("no group", dict(group_wgrad=False), None), ("no fused images", {}, "img"), ("no ingest", {}, "ing")]:
    params = M.init_params(cfg, seed=11)
    m1 = _make_engine_model(cfg, params)
    ft = FusedTrainer(m1, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=False, **kw)
    if hack == "img":
        ft._fused_images = False
    yy = yt.double() if hack == "ing" else yt
    out = [float(ft.step(xt, yy).item()) for _ in range(3)]
    print(name, out, flush=True)
