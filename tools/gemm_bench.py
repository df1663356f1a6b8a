"""Microbenchmark of the split-bf16 GEMMs at the config-3 shapes (run via gpurun)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
DEV = "cuda:0"

def ev(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

M = int(os.environ.get("GM", 15360))
shapes = [(M, 512, 136), (M, 1536, 512), (M, 512, 512), (M, 2048, 512), (M, 512, 2048)]
for (m, n, k) in shapes:
    A = torch.randn(m, k, device=DEV); B = torch.randn(n, k, device=DEV) / k ** 0.5; bias = torch.randn(n, device=DEV)
    C = torch.empty(m, n, device=DEV)
    fl = 2.0 * m * n * k
    row = dict(shape=(m, n, k))
    for v in (1, 2, 3, 5):
        us = ev(lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(B), k, None, LB.ptr(C), n, m, n, k, LB.ptr(bias), 0, None, 0, 0.0, 0, None, 0, v, None), "nt"))
        row["nt_v%d" % v] = "%.1fus %.0fTF" % (us, fl / us / 1e6)
    us = ev(lambda: torch.addmm(bias, A, B.t(), out=C))
    row["hipblaslt_fp32"] = "%.1fus %.0fTF" % (us, fl / us / 1e6)
    # wgrad: dW[n,k] = dY[m,n]^T X[m,k]
    dY = torch.randn(m, n, device=DEV); X = torch.randn(m, k, device=DEV)
    gW = torch.empty(n, k, device=DEV); gb = torch.empty(n, device=DEV)
    ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(m, n, k), 64), dtype=torch.uint8, device=DEV)
    us = ev(lambda: LB.check(lib.ltrx_gemm_tn(LB.ptr(dY), n, LB.ptr(X), k, LB.ptr(gW), LB.ptr(gb), m, n, k, 0, 0, LB.ptr(ws), None), "tn"))
    row["tn"] = "%.1fus %.0fTF" % (us, fl / us / 1e6)
    us = ev(lambda: torch.mm(dY.t(), X, out=gW))
    row["hipblaslt_tn"] = "%.1fus %.0fTF" % (us, fl / us / 1e6)
    print(json.dumps(row), flush=True)
