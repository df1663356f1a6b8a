"""a few eager training steps of the bench model for counter passes: SLATES (64), STEP_OPTS e.g. "group_wgrad=0,relu_bits=0,pad_input=0"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from allrank_amd.model import make_model
from allrank_amd.engine import FusedTrainer
B, L, F = int(os.environ.get("SLATES", 64)), 240, 136
kw = {k: bool(int(v)) for k, v in (kv.split("=") for kv in os.environ.get("STEP_OPTS", "").split(",") if kv)}
torch.manual_seed(0)
m = make_model(dict(sizes=[512], input_norm=False, activation=None, dropout=0.0), dict(N=2, d_ff=2048, h=8, positional_encoding=None, dropout=0.0),
               dict(d_output=1, output_activation=None), F).to("cuda")
ft = FusedTrainer(m, "approxNDCGLoss", {}, B, L, lr=1e-3, use_graph=False, **kw)
rng = np.random.default_rng(0)
x = torch.tensor(rng.standard_normal((B, L, F)).astype(np.float32), device="cuda")
y = torch.tensor(rng.integers(0, 5, (B, L)).astype(np.float32), device="cuda")
for _ in range(int(os.environ.get("STEPS", 4))):
    ft.step(x, y)
torch.cuda.synchronize()
