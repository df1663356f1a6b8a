"""attention kernels on packed batches of UNIFORM slate length: time vs length (how much is per-workgroup overhead).
usage (GPU box): python tools/mha_varlen_timing.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from allrank_amd import _lib as LB
lib = LB.lib()
B, Lmax, h, dk = 256, 240, 8, 64
d = h * dk
dev = "cuda"
for ln in (16, 32, 64, 96, 112, 128, 160, 192, 240):
    n = B * ln
    qkv = torch.randn(n, 3 * d, device=dev)
    do = torch.randn(n, d, device=dev)
    o = torch.empty(n, d, device=dev)
    lse = torch.empty(B, h, Lmax, device=dev)
    dqkv = torch.empty(n, 3 * d, device=dev)
    cu = (torch.arange(B + 1, dtype=torch.int32) * ln).to(dev)
    ws = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, Lmax, h, dk, 1), 64), dtype=torch.uint8, device=dev)
    st = LB.stream_of(qkv)

    def fwd():
        LB.check(lib.ltrx_mha_fwd(LB.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, None, B, Lmax, h, dk, 3 * d, LB.ptr(o), d,
                                  LB.ptr(lse), 0.0, 0, None, LB.ptr(cu), None, 1, st), "fwd")

    def bwd():
        LB.check(lib.ltrx_mha_bwd(LB.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, None, LB.ptr(o), LB.ptr(do), LB.ptr(lse), B,
                                  Lmax, h, dk, 3 * d, d, LB.ptr(dqkv), dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d, 3 * d, 0.0, 0,
                                  None, LB.ptr(cu), None, 1, LB.ptr(ws), st), "bwd")
    res = []
    for fn in (fwd, bwd):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 10 * 1e3)
    qb, kt = -(-ln // 128), -(-ln // 32)
    wq = -(-ln // 32)
    print("len %3d  q-blocks %d  key tiles %d  block-tiles %2d  wave-tiles %2d   fwd %7.1f us   bwd %7.1f us" % (ln, qb, kt, qb * kt, wq * kt, res[0], res[1]))
