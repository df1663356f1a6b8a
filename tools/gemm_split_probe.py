"""NT GEMM at row counts whose 256-row tiling gives between one and ~1.4 rounds of workgroups (N = 512): one launch as
dispatched vs one full round of 256 x 256 tiles + the remaining rows in a second launch.  usage: python tools/gemm_split_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from allrank_amd import _lib as LB
lib = LB.lib()
N = 512


def call(A, W, C, b, m0, m1):
    K = A.shape[1]
    LB.check(lib.ltrx_gemm_nt(A.data_ptr() + 4 * K * m0, K, LB.ptr(W), K, None, C.data_ptr() + 4 * N * m0, N, m1 - m0, N, K, LB.ptr(b), 0, None, 0,
                              0.0, 0, None, 0, 0, None), "nt")


def ev(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 100


for K in (512, 2048):
    for M in (30720, 33000, 34816, 38400, 40000, 44000, 46080, 50000):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") / K ** 0.5
        b = torch.randn(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        t1 = ev(lambda: call(A, W, C, b, 0, M))
        M1 = (256 // (N // 256)) * 256
        t2 = ev(lambda: (call(A, W, C, b, 0, M1), call(A, W, C, b, M1, M))) if M > M1 else float("nan")
        tiles = -(-M // 256) * (N // 256)
        print("K %4d  M %6d  tiles256 %4d   one launch %7.1f us   split %7.1f us   per-row %.2f / %.2f ns" % (K, M, tiles, t1, t2, t1 * 1e3 / M, t2 * 1e3 / M))
