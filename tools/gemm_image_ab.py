"""A/B of the NT GEMM with the weight operand split on the fly (fp32 B) vs read from the pre-split image (ltrx_split_image) at the shapes
of one training step.  Prints per shape: us per launch (HIP events, 12 launches, variants interleaved, best of two rounds), algorithmic TF,
bit-equality of the outputs.  usage: [GSLATES=256] python tools/gemm_image_ab.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
DEV = "cuda"
M = int(os.environ.get("GSLATES", "256")) * 240


def run(img, A, W, C, bias, act, aux):
    n, k = W.shape
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(W), k, LB.ptr(img), LB.ptr(C), n, A.shape[0], n, k, LB.ptr(bias), act, LB.ptr(aux),
                              n if aux is not None else 0, 0.0, 0, None, 0, 0, None), "nt")


def timeit(fn, iters=12):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for (n, k, act, name) in [(2048, 512, 1, "ffn1 fwd (bias+relu)"), (2048, 512, 2, "ffn2 dgrad (relu mask)"), (512, 2048, 0, "ffn2 fwd / ffn1 dgrad"),
                          (1536, 512, 0, "qkv fwd"), (512, 512, 0, "out proj"), (512, 1536, 0, "qkv dgrad")]:
    A = torch.randn(M, k, device=DEV); W = torch.randn(n, k, device=DEV) / k ** 0.5
    bias = torch.randn(n, device=DEV) if act != 2 else None
    aux = torch.randn(M, n, device=DEV) if act == 2 else None
    img = torch.empty_like(W)
    LB.check(lib.ltrx_split_image(LB.ptr(W), LB.ptr(img), W.numel(), None), "split_image")
    rec = dict(shape=[M, n, k], what=name)
    C0, C1 = torch.empty(M, n, device=DEV), torch.empty(M, n, device=DEV)
    run(None, A, W, C0, bias, act, aux); run(img, A, W, C1, bias, act, aux)
    rec["bit_identical"] = bool(torch.equal(C0, C1))
    for rep in range(2):
        for key, im, C in (("fp32_us", None, C0), ("image_us", img, C1)):
            us = timeit(lambda: run(im, A, W, C, bias, act, aux))
            rec[key] = round(min(us, rec.get(key, 1e9)), 1)
    rec["gain_pct"] = round(100.0 * (1 - rec["image_us"] / rec["fp32_us"]), 1)
    rec["image_tf"] = round(2.0 * M * n * k / rec["image_us"] / 1e6, 1)
    print(json.dumps(rec), flush=True)
    del A, W, aux, C0, C1
