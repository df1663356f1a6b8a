import os, sys, torch, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
def ev(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for (m, n, k) in [(15360, 2048, 512), (15360, 512, 2048), (15360, 512, 512), (122880, 2048, 512)]:
    A = torch.randn(m, k, device="cuda"); B = torch.randn(n, k, device="cuda"); C = torch.empty(m, n, device="cuda")
    row = dict(shape=(m, n, k))
    for v in (1, 101, 3, 103, 5, 105):
        us = ev(lambda: lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(B), k, None, LB.ptr(C), n, m, n, k, None, 0, None, 0, 0.0, 0, None, 0, v, None))
        row["v%d" % v] = "%.1fus %.0fTF" % (us, 2.0 * m * n * k / us / 1e6)
    print(json.dumps(row), flush=True)
