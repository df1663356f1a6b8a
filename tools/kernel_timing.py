"""Per-kernel timing on the MI355X (run via gpurun).  Prints one JSON line per kernel: µs per call at given shapes."""
import json
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import losses as E, metrics as EM, ops  # noqa: E402
from tests.golden.make_inputs import make_inputs  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / iters


def main():
    out = []
    for B, L in [(64, 240), (2048, 240), (64, 1024)]:
        s, y = make_inputs(B, L, 1)
        y[:] = np.where(y < 0, 0, y)      # dense slates (headline definition)
        st = torch.tensor(s, device=DEV, requires_grad=True)
        yt = torch.tensor(y, device=DEV)
        perm = torch.randperm(L).to(DEV)
        fns = {
            "listnet": lambda: E.listNet(st, yt),
            "listmle": lambda: E.listMLE(st, yt, perm=perm),
            "approxndcg": lambda: E.approxNDCGLoss(st, yt),
            "lambdarank": lambda: E.lambdaLoss(st, yt, weighing_scheme="lambdaRank_scheme"),
            "ndcgloss2pp": lambda: E.lambdaLoss(st, yt, weighing_scheme="ndcgLoss2PP_scheme"),
            "ndcg@5": lambda: EM.ndcg(st, yt, ats=[5]),
        }
        if L <= 256:
            fns["neuralndcg"] = lambda: E.neuralNDCG(st, yt)
        for name, fn in fns.items():
            it = 5 if (name == "neuralndcg" and B > 64) else 20
            us = timeit(fn, iters=it)
            rec = dict(kernel=name, B=B, L=L, us=round(us, 1), items_per_s=round(B * L / us * 1e6))
            print(json.dumps(rec), flush=True)
            out.append(rec)
    # model kernels at config (3): d=512, h=8
    for B, L, h, dk in [(64, 240, 8, 64), (512, 240, 8, 64)]:
        d = h * dk
        qkv = torch.randn(B, L, 3 * d, device=DEV, requires_grad=True)
        mask = torch.zeros(B, L, dtype=torch.bool, device=DEV)
        go = torch.randn(B, L, d, device=DEV)
        q, k, v = qkv[:, :, :d], qkv[:, :, d:2 * d], qkv[:, :, 2 * d:]
        fl = 4.0 * B * h * L * L * dk
        with torch.no_grad():
            us = timeit(lambda: ops.attention(q, k, v, mask, h))
        print(json.dumps(dict(kernel="mha_fwd", B=B, L=L, us=round(us, 1), tflops=round(fl / us / 1e6, 2))), flush=True)

        def fb():
            qkv.grad = None
            o = ops.attention(q, k, v, mask, h)
            o.backward(go)
        us2 = timeit(fb)
        print(json.dumps(dict(kernel="mha_fwd+bwd", B=B, L=L, us=round(us2, 1), tflops=round(3.5 * fl / us2 / 1e6, 2))), flush=True)
        # torch reference attention for comparison (materialised)
        def tref():
            qh = q.reshape(B, L, h, dk).transpose(1, 2)
            kh = k.reshape(B, L, h, dk).transpose(1, 2)
            vh = v.reshape(B, L, h, dk).transpose(1, 2)
            sc = qh @ kh.transpose(-1, -2) / 8.0
            return torch.softmax(sc, -1) @ vh
        with torch.no_grad():
            us3 = timeit(tref)
        print(json.dumps(dict(kernel="torch_attn_fwd_materialised", B=B, L=L, us=round(us3, 1))), flush=True)
        x = torch.randn(B * L, d, device=DEV, requires_grad=True)
        r = torch.randn(B * L, d, device=DEV)
        a = torch.ones(d, device=DEV, requires_grad=True)
        bb = torch.zeros(d, device=DEV, requires_grad=True)
        with torch.no_grad():
            us = timeit(lambda: ops.layer_norm_residual(x, r, a, bb))
        by = B * L * d * 4 * 4
        print(json.dumps(dict(kernel="layernorm_res_fwd", rows=B * L, us=round(us, 1), GBps=round(by / us / 1e3, 1))), flush=True)
        # GEMM sanity: what torch (hipBLASLt/rocBLAS) gives on the FFN shape in fp32 and bf16
        W = torch.randn(2048, d, device=DEV)
        X = torch.randn(B * L, d, device=DEV)
        with torch.no_grad():
            us = timeit(lambda: X @ W.t())
            fl2 = 2.0 * B * L * d * 2048
            print(json.dumps(dict(kernel="torch_gemm_fp32_ffn1", M=B * L, us=round(us, 1), tflops=round(fl2 / us / 1e6, 1))), flush=True)
            Xb, Wb = X.bfloat16(), W.bfloat16()
            us = timeit(lambda: Xb @ Wb.t())
            print(json.dumps(dict(kernel="torch_gemm_bf16_ffn1", M=B * L, us=round(us, 1), tflops=round(fl2 / us / 1e6, 1))), flush=True)


if __name__ == "__main__":
    main()
