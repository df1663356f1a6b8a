"""time ltrx_gemm_nt with 64-row tiles / two workgroups per CU (tile 8) against the automatic choice and the 128- and 256-row
large-tile forms (7, 6), with the weight operand from its pre-split image as in the training step; checks tile 8 == tile 0 bits."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()


def ev(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


shapes = [(15360, 512, 512), (15360, 512, 2048), (15360, 512, 1536), (15360, 1536, 512), (15360, 2048, 512), (15360, 512, 160),
          (30720, 512, 512), (30720, 512, 2048), (61440, 512, 512), (61440, 512, 2048), (7680, 512, 2048), (7680, 2048, 512)]
for (m, n, k) in shapes:
    A = torch.randn(m, k, device="cuda"); B = torch.randn(n, k, device="cuda") / k ** 0.5; bias = torch.randn(n, device="cuda")
    img = torch.empty_like(B)
    LB.check(lib.ltrx_split_image(LB.ptr(B), LB.ptr(img), B.numel(), None), "split_image")
    outs = {}
    rec = dict(shape=(m, n, k), tiles256=((m + 255) // 256) * (n // 256), tiles128=((m + 127) // 128) * (n // 256), tiles64=((m + 63) // 64) * (n // 256))
    for v in (0, 8, 7, 6):
        C = torch.empty(m, n, device="cuda")
        call = lambda: lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(B), k, LB.ptr(img), LB.ptr(C), n, m, n, k, LB.ptr(bias), 0, None, 0, 0.0, 0, None, 0, v, None)
        if call() != 0:
            rec["v%d_us" % v] = None
            continue
        rec["v%d_us" % v] = round(ev(call), 1)
        outs[v] = C
    rec["tf_auto"] = round(2.0 * m * n * k / rec["v0_us"] / 1e6, 1)
    if rec.get("v8_us"):
        rec["tf_v8"] = round(2.0 * m * n * k / rec["v8_us"] / 1e6, 1)
        rec["v8_equals_auto_bits"] = bool(torch.equal(outs[8], outs[0]))
    print(json.dumps(rec), flush=True)
