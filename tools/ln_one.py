"""LayerNorm forward / backward kernels alone at the bench size (rows 61440, d 512), 5 launches each (for rocprofv3 passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
M, d = int(os.environ.get("GM", "61440")), 512
f = dict(device="cuda", dtype=torch.float32)
x, res, dy, dres = (torch.randn(M, d, **f) for _ in range(4))
a, b = torch.ones(d, **f), torch.zeros(d, **f)
xsum, y, dx = (torch.empty(M, d, **f) for _ in range(3))
mean, rstd = torch.empty(M, **f), torch.empty(M, **f)
da, db = torch.empty(d, **f), torch.empty(d, **f)
ws = torch.empty(max(lib.ltrx_layernorm_bwd_workspace_bytes(M, d), 64), dtype=torch.uint8, device="cuda")
P = LB.ptr
nores = os.environ.get("LN_NORES", "1") == "1"      # the explicit step's form since round 3: one stream in (the residual sum is a GEMM epilogue)
for _ in range(5):
    LB.check(lib.ltrx_layernorm_fwd(P(x), None if nores else P(res), P(a), P(b), M, d, 1e-6, None if nores else P(xsum), P(y), P(mean), P(rstd),
                                    0.0, 0, None, None), "fwd")
xsum = x if nores else xsum
for _ in range(5):
    LB.check(lib.ltrx_layernorm_bwd(P(dy), P(xsum), P(a), P(mean), P(rstd), P(dres), M, d, 1e-6, P(dx), P(da), P(db), P(ws), None), "bwd")
torch.cuda.synchronize()
