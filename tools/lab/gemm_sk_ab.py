"""same-box A/B of the split-K form of the NT GEMM (ltrx_gemm_nt_sk) against ltrx_gemm_nt at the N = 512 shapes of a 64-slate step
(M = 15360): microseconds (HIP events, interleaved, best of 3 rounds of 12 launches), algorithmic TF, max error of both against fp64,
run-to-run bit identity of the split form.  usage: python tools/gemm_sk_ab.py > profiles/r04_gemm_b64_ab.md"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from allrank_amd import _lib as LB
lib = LB.lib()


def ev(fn, n=12):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


print("# split-K (x2, in-kernel fix-up) vs one tile per workgroup: NT GEMMs of a 64-slate step (M = 15360), same box\n")
print("| M x N x K | what | epilogue | ltrx_gemm_nt us (TF) | ltrx_gemm_nt_sk us (TF) | speed-up | max err vs fp64: nt / sk (of max abs C) | sk bit-identical over 10 runs |")
print("|---|---|---|---|---|---|---|---|")
g = torch.Generator(device="cuda").manual_seed(5)
for (M, N, K, what) in ((15360, 512, 2048, "ffn2 fwd / ffn1 dgrad"), (15360, 512, 1536, "qkv dgrad"), (15360, 512, 1024, "(K = 1024)"),
                        (15360, 512, 512, "out proj (not selected: K < 1024)"), (16384, 512, 2048, "M = 16384"), (30720, 512, 2048, "128 slates (not selected)")):
    A = torch.randn(M, K, device="cuda", generator=g)
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    bias = torch.randn(N, device="cuda", generator=g)
    aux = torch.randn(M, N, device="cuda", generator=g)
    img = torch.empty_like(W)
    LB.check(lib.ltrx_split_image(LB.ptr(W), LB.ptr(img), W.numel(), None), "img")
    nb = lib.ltrx_gemm_nt_sk_workspace_bytes(M, N, K)
    ws = torch.zeros(max(nb, 64), dtype=torch.uint8, device="cuda")
    ref = None
    for (act, name) in ((1, "bias + ReLU"), (3, "bias + residual"), (2, "ReLU mask")):
        C0, C1 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
        b_ = None if act == 2 else LB.ptr(bias)
        ax = LB.ptr(aux) if act >= 2 else None
        f0 = lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A), K, LB.ptr(W), K, LB.ptr(img), LB.ptr(C0), N, M, N, K, b_, act, ax, N if act >= 2 else 0, 0.0, 0, None, 0, 0, None), "nt")
        f1 = lambda: LB.check(lib.ltrx_gemm_nt_sk(LB.ptr(A), K, LB.ptr(W), K, LB.ptr(img), LB.ptr(C1), N, M, N, K, b_, act, ax, N if act >= 2 else 0, 0.0, 0, None, 0, 0, LB.ptr(ws), ws.numel(), None), "sk")
        t0 = t1 = 1e9
        for _ in range(3):
            t0 = min(t0, ev(f0)); t1 = min(t1, ev(f1))
        if ref is None:
            ref = A.double() @ W.double().t()
        r = ref + (bias.double() if act != 2 else 0)
        if act == 1:
            r = r.clamp(min=0)
        elif act == 3:
            r = r + aux.double()
        elif act == 2:
            r = torch.where(aux > 0, r, torch.zeros_like(r))
        sc = float(r.abs().max())
        e0, e1 = float((C0.double() - r).abs().max()) / sc, float((C1.double() - r).abs().max()) / sc
        same = True
        keep = C1.clone()
        for _ in range(10):
            f1()
            same = same and torch.equal(C1, keep)
        fl = 2.0 * M * N * K
        print("| %d x %d x %d | %s | %s | %.1f (%.0f) | %.1f (%.0f) | %.2fx | %.2e / %.2e | %s |" %
              (M, N, K, what, name, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t0 / t1, e0, e1, same), flush=True)
        assert int(ws[nb - 4 * ((M // 256) * (N // 256)):nb].max()) == 0 if nb else True      # the counters are back at zero
    del A, W, aux, ref
