"""attention backward at the bench shape as ONE call vs chunks of slates (the dS hand-over of a chunk of 64 slates is 134 MB: does it
stay in the 256-MB Infinity Cache between the dK/dV kernel that writes it and the dQ kernel that reads it?)  ->  us per layer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
B, L, h, dk = 256, 240, 8, 64
d = h * dk
dev = "cuda"
qkv = torch.randn(B * L, 3 * d, device=dev)
do = torch.randn(B * L, d, device=dev)
o = torch.empty(B * L, d, device=dev)
lse = torch.empty(B, h, L, device=dev)
mask = torch.zeros(B, L, dtype=torch.uint8, device=dev)
P = LB.ptr
LB.check(lib.ltrx_mha_fwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), B, L, h, dk, 3 * d, P(o), d, P(lse), 0.0, 0,
                          None, None, None, 1, None), "fwd")
ws = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, L, h, dk, 1), 64), dtype=torch.uint8, device=dev)
filler = torch.empty(64 << 20, device=dev)     # 256 MB written between timed repetitions: the cache starts cold


def run(chunk, out):
    for b0 in range(0, B, chunk):
        r0 = b0 * L
        q = qkv.data_ptr() + 4 * r0 * 3 * d
        dq = out.data_ptr() + 4 * r0 * 3 * d
        LB.check(lib.ltrx_mha_bwd(q, q + 4 * d, q + 8 * d, mask.data_ptr() + r0, o.data_ptr() + 4 * r0 * d, do.data_ptr() + 4 * r0 * d,
                                  lse.data_ptr() + 4 * b0 * h * L, chunk, L, h, dk, 3 * d, d, dq, dq + 4 * d, dq + 8 * d, 3 * d, 0.0, 0,
                                  None, None, None, 1, P(ws), None), "bwd")


ref = torch.empty(B * L, 3 * d, device=dev)
run(B, ref)
for chunk in (256, 128, 64, 32):
    out = torch.empty(B * L, 3 * d, device=dev)
    run(chunk, out)
    same = torch.equal(out, ref)
    ts = []
    for rep in range(6):
        filler.fill_(1.0)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run(chunk, out)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print("chunk %3d slates: %.1f us per layer (min of 6: %s)  bit-identical to one call: %s" % (chunk, min(ts), " ".join("%.0f" % t for t in ts), same), flush=True)
