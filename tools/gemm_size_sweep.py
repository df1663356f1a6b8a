"""time per row of the NT and weight-gradient GEMMs over the row count (looking for dispatch cliffs).  usage: python tools/gemm_size_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from allrank_amd import _lib as LB
lib = LB.lib()


def ev(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(8):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 125


Ms = [int(a) for a in sys.argv[1:]] or [7680, 11520, 15360, 19200, 23040, 26880, 30720, 34560, 38400, 42240, 46080, 53760, 61440]
for (N, K) in ((2048, 512), (1536, 512), (512, 2048), (512, 512)):
    row = []
    for M in Ms:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        t = ev(lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(W), K, None, LB.ptr(C), N, M, N, K, LB.ptr(b), 0, None, 0, 0.0, 0, None, 0, 0, None), "nt"))
        row.append("%d:%.0f(%.0fTF)" % (M // 1024, t, 2.0 * M * N * K / t / 1e6))
        del A, W, C
    print("NT N%d K%d  " % (N, K) + "  ".join(row))
for (NP, KP) in ((2048, 512), (1536, 512), (512, 512)):
    row = []
    for M in Ms:
        dY = torch.randn(M, NP, device="cuda"); X = torch.randn(M, KP, device="cuda")
        gW = torch.empty(NP, KP, device="cuda"); gb = torch.empty(NP, device="cuda")
        ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(M, NP, KP), 64), dtype=torch.uint8, device="cuda")
        t = ev(lambda: LB.check(lib.ltrx_gemm_tn(LB.ptr(dY), NP, LB.ptr(X), KP, LB.ptr(gW), LB.ptr(gb), M, NP, KP, 0, 0, LB.ptr(ws), None), "tn"))
        row.append("%d:%.0f(%.0fTF)" % (M // 1024, t, 2.0 * M * NP * KP / t / 1e6))
        del dY, X
    print("TN dW[%d,%d] " % (NP, KP) + "  ".join(row))
