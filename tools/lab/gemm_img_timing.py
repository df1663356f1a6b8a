"""Same-box A/B of the on-the-fly split GEMMs (ltrx_gemm_nt / ltrx_gemm_tn) and the pre-split image GEMMs
(ltrx_gemm_nt_img / ltrx_gemm_tn_img) at the shapes of one training step (config 3, 256 slates):  python tools/gemm_img_timing.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB  # noqa: E402

lib = LB.lib()
dev = "cuda:0"


def ev(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def img(x):
    o = torch.empty_like(x)
    LB.check(lib.ltrx_to_image(LB.ptr(x), x.stride(0), x.shape[0], x.shape[1], LB.ptr(o), x.shape[1], LB.stream_of(x)), "to_image")
    return o


M = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
print("NT  (M, N, K)          old us    img us   ratio   img TF(alg)")
for (N, K, act, cimg) in [(1536, 512, 0, 0), (512, 512, 0, 0), (2048, 512, 1, 1), (512, 2048, 0, 0), (2048, 512, 2, 1), (512, 1536, 0, 0)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev); aux = torch.randn(M, N, device=dev)
    Ai, Wi, auxi = img(A), img(W), img(aux)
    st = LB.stream_of(A)
    t0 = ev(lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(W), K, LB.ptr(C), N, M, N, K, LB.ptr(b) if act != 2 else None, act,
                                              LB.ptr(aux) if act == 2 else None, N if act == 2 else 0, 0.0, 0, None, 0, st), "nt"))
    t1 = ev(lambda: LB.check(lib.ltrx_gemm_nt_img(LB.ptr(Ai), K, LB.ptr(Wi), K, LB.ptr(C), N, cimg, M, N, K, LB.ptr(b) if act != 2 else None, act,
                                                  LB.ptr(auxi) if act == 2 else None, N if act == 2 else 0, 0.0, 0, None, st), "nt_img"))
    print("NT  %6d %5d %5d act%d img%d  %8.1f  %8.1f  %6.3f  %8.1f" % (M, N, K, act, cimg, t0, t1, t1 / t0, 2.0 * M * N * K / t1 / 1e6))
print("TN  (M, NP, KP)")
for (NP, KP) in [(1536, 512), (512, 512), (2048, 512), (512, 2048)]:
    dY = torch.randn(M, NP, device=dev); X = torch.randn(M, KP, device=dev)
    C = torch.empty(NP, KP, device=dev); gb = torch.empty(NP, device=dev)
    ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(M, NP, KP), 64), dtype=torch.uint8, device=dev)
    dYi, Xi = img(dY), img(X)
    st = LB.stream_of(dY)
    t0 = ev(lambda: LB.check(lib.ltrx_gemm_tn(LB.ptr(dY), NP, LB.ptr(X), KP, LB.ptr(C), LB.ptr(gb), M, NP, KP, 0, LB.ptr(ws), st), "tn"))
    t1 = ev(lambda: LB.check(lib.ltrx_gemm_tn_img(LB.ptr(dYi), NP, LB.ptr(Xi), KP, LB.ptr(C), LB.ptr(gb), M, NP, KP, LB.ptr(ws), st), "tn_img"))
    print("TN  %6d %5d %5d            %8.1f  %8.1f  %6.3f  %8.1f" % (M, NP, KP, t0, t1, t1 / t0, 2.0 * M * NP * KP / t1 / 1e6))
