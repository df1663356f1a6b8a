"""time the attention kernels in both arithmetic modes (0 = exact fp32 MFMA, 1 = split-bf16 MFMA) at config (3)"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB, ops
lib = LB.lib()
DEV = "cuda:0"


def timeit(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


B, L, h, dk = 256, 240, 8, 64
d = h * dk
qkv = torch.randn(B, L, 3 * d, device=DEV, requires_grad=True)
mask = torch.zeros(B, L, dtype=torch.bool, device=DEV)
go = torch.randn(B, L, d, device=DEV)
q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
for mode in (0, 1, 0, 1):
    ops.set_attention_mode(mode)
    with torch.no_grad():
        f = timeit(lambda: ops.attention(q, k, v, mask, h))

    def fb():
        qkv.grad = None
        ops.attention(q, k, v, mask, h).backward(go)
    fbt = timeit(fb)
    print(json.dumps(dict(mode=mode, fwd_us=round(f, 1), fwd_bwd_us=round(fbt, 1))), flush=True)
ops.set_attention_mode(1)
