"""HIP-event timing of the LayerNorm backward kernel alone (rows 61440, d 512).  usage: python tools/ln_time.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
M, d = int(os.environ.get('LN_ROWS', 61440)), 512
f = dict(device="cuda", dtype=torch.float32)
dy, dres, xsum = (torch.randn(M, d, **f) for _ in range(3))
a = torch.ones(d, **f)
dx = torch.empty(M, d, **f)
mean, rstd = torch.zeros(M, **f), torch.ones(M, **f)
da, db = torch.empty(d, **f), torch.empty(d, **f)
ws = torch.empty(max(lib.ltrx_layernorm_bwd_workspace_bytes(M, d), 64), dtype=torch.uint8, device="cuda")
P = LB.ptr
fn = lambda: LB.check(lib.ltrx_layernorm_bwd(P(dy), P(xsum), P(a), P(mean), P(rstd), P(dres), M, d, 1e-6, P(dx), P(da), P(db), P(ws), None), "bwd")
for _ in range(5):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    fn()
e1.record()
torch.cuda.synchronize()
print("layernorm_bwd (+reduce) rows %d: %.1f us" % (M, e0.elapsed_time(e1) * 50))
