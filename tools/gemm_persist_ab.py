"""A/B of the NT GEMM tile forms at the shapes of one training step: tile 6 = 256x256x32 one tile per workgroup (round 2),
8 / 9 = the persistent form (tile order 0 / 1).  Prints per shape: us per launch (HIP events, 12 launches, variants interleaved),
algorithmic TF, bit-equality of the outputs, and the same on all-zero operands (DVFS probe).  usage: [GSLATES=256] python tools/gemm_persist_ab.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
DEV = "cuda"
M = int(os.environ.get("GSLATES", "256")) * 240
VARS = [int(v) for v in os.environ.get("GVARS", "6,8,9").split(",")]


def run(v, A, W, C, bias, act, aux):
    n, k = W.shape
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(W), k, None, LB.ptr(C), n, A.shape[0], n, k, LB.ptr(bias), act, LB.ptr(aux), n if aux is not None else 0,
                              0.0, 0, None, 0, v, None), "nt v%d" % v)


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
    for zero in (False, True):
        A = torch.randn(M, k, device=DEV); W = torch.randn(n, k, device=DEV) / k ** 0.5
        bias = torch.randn(n, device=DEV) if act != 2 else None
        aux = torch.randn(M, n, device=DEV) if act == 2 else None
        if zero:
            A.zero_(); W.zero_()
        outs, rec = {}, dict(shape=[M, n, k], what=name, operands="zero" if zero else "randn")
        for v in VARS:
            C = torch.empty(M, n, device=DEV)
            run(v, A, W, C, bias, act, aux)
            outs[v] = C
        for rep in range(2):
            for v in VARS:
                us = timeit(lambda: run(v, A, W, outs[v], bias, act, aux))
                rec["v%d_us" % v] = round(min(us, rec.get("v%d_us" % v, 1e9)), 1)
        for v in VARS:
            rec["v%d_tf" % v] = round(2.0 * M * n * k / rec["v%d_us" % v] / 1e6, 1)
            if v != VARS[0]:
                rec["v%d_equal_v%d" % (v, VARS[0])] = bool(torch.equal(outs[v], outs[VARS[0]]))
        print(json.dumps(rec), flush=True)
        del A, W, aux, outs
