"""time ltrx_gemm_nt per shape with the tile variant forced: 1 = 128x128x32, 6 = 256x256x32, 0 = automatic choice"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()


def ev(fn, iters=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for (m, n, k) in [(15360, 1536, 512), (15360, 2048, 512), (15360, 512, 512), (15360, 512, 2048), (15360, 512, 1536),
                  (30720, 512, 512), (30720, 512, 2048), (61440, 512, 512)]:
    A = torch.randn(m, k, device="cuda"); B = torch.randn(n, k, device="cuda") / k ** 0.5; bias = torch.randn(n, device="cuda")
    C = torch.empty(m, n, device="cuda")
    rec = dict(shape=(m, n, k), tiles256=((m + 255) // 256) * (n // 256))
    for v in (1, 6, 0):
        rec["v%d_us" % v] = round(ev(lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(B), k, None, LB.ptr(C), n, m, n, k, LB.ptr(bias), 0,
                                                                        None, 0, 0.0, 0, None, 0, v, None), "nt")), 1)
    print(json.dumps(rec), flush=True)
