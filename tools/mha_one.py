"""the three attention kernels alone at the bench shape (256 slates x 240 items, 8 heads x 64), 5 launches each (rocprofv3 passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
B, L, h, dk = (int(os.environ.get(k, d)) for k, d in (('MB', 256), ('ML', 240), ('MH', 8), ('MDK', 64)))
d = h * dk
dev = "cuda"
qkv = torch.randn(B * L, 3 * d, device=dev)
do = torch.randn(B * L, d, device=dev)
o = torch.empty(B * L, d, device=dev)
lse = torch.empty(B, h, L, device=dev)
dqkv = torch.empty(B * L, 3 * d, device=dev)
mask = torch.zeros(B, L, dtype=torch.uint8, device=dev)
P = LB.ptr
MODE = int(os.environ.get('MMODE', '1'))     # attention arithmetic of the calls: 0 exact fp32, 1 split-bf16, 2 plain bf16
ws = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, L, h, d // h, MODE), 64), dtype=torch.uint8, device=dev)
for _ in range(5):
    LB.check(lib.ltrx_mha_fwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), B, L, h, dk, 3 * d, P(o), d, P(lse), 0.0, 0,
                              None, None, None, MODE, None), "fwd")
for _ in range(5):
    LB.check(lib.ltrx_mha_bwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), P(o), P(do), P(lse), B, L, h, dk, 3 * d, d,
                              P(dqkv), dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d, 3 * d, 0.0, 0, None, None, None, MODE, P(ws), None), "bwd")
torch.cuda.synchronize()
