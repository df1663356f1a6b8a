"""torch.autograd bindings of the scoring-model kernels of libltrx.so (custom LayerNorm, fused masked attention).

PyTorch owns the tensors and the autograd graph edges; all arithmetic happens in the HIP kernels behind the C ABI
(include/ltrx.h).  Device tensors only, fp32, no CPU fallback.
"""
import torch

from . import _lib as L


class _LayerNormFn(torch.autograd.Function):
    """y = a*(xsum-mean)/(std_unbiased+eps)+b with xsum = x (+ res).  Returns (y, xsum).
    allrank/models/transformer.py:73-81 (+ the residual sum of :105)."""

    @staticmethod
    def forward(ctx, x, res, a, b, eps):
        L.require_device(x, res, a, b)
        shape = x.shape
        D = shape[-1]
        x2 = L.f32c(x).view(-1, D)
        r2 = L.f32c(res).view(-1, D) if res is not None else None
        a = L.f32c(a)
        b = L.f32c(b)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        xsum = torch.empty_like(x2) if r2 is not None else x2
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(L.lib().ltrx_layernorm_fwd(L.ptr(x2), L.ptr(r2), L.ptr(a), L.ptr(b), rows, D, float(eps),
                                           L.ptr(xsum) if r2 is not None else None, L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                           0.0, 0, None, L.stream_of(x2)), "layernorm_fwd")
        ctx.save_for_backward(xsum, a, mean, rstd)
        ctx.eps = float(eps)
        ctx.has_res = res is not None
        ctx.shape = shape
        # without a residual input xsum IS x: do not hand an input back as an output
        return y.view(shape), (xsum.view(shape) if r2 is not None else None)

    @staticmethod
    def backward(ctx, dy, dxsum):
        xsum, a, mean, rstd = ctx.saved_tensors
        D = xsum.shape[-1]
        rows = xsum.shape[0]
        dy2 = L.f32c(dy).view(-1, D)
        dres = L.f32c(dxsum).view(-1, D) if dxsum is not None else None
        dx = torch.empty_like(xsum)
        da = torch.empty(D, dtype=torch.float32, device=xsum.device)
        db = torch.empty(D, dtype=torch.float32, device=xsum.device)
        lib = L.lib()
        ws = L.workspace(lib.ltrx_layernorm_bwd_workspace_bytes(rows, D), xsum)
        L.check(lib.ltrx_layernorm_bwd(L.ptr(dy2), L.ptr(xsum), L.ptr(a), L.ptr(mean), L.ptr(rstd), L.ptr(dres), rows, D,
                                       ctx.eps, L.ptr(dx), L.ptr(da), L.ptr(db), L.ptr(ws), L.stream_of(xsum)),
                "layernorm_bwd")
        dx = dx.view(ctx.shape)
        return dx, (dx if ctx.has_res else None), da, db, None


def layer_norm_residual(x, res, a, b, eps=1e-6):
    """(LN(x + res), x + res): the pre-norm residual stream step.  ``res`` may be None."""
    return _LayerNormFn.apply(x, res, a, b, eps)


def layer_norm(x, a, b, eps=1e-6):
    return _LayerNormFn.apply(x, None, a, b, eps)[0]


class _AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d_k) + key padding mask) v per (slate, head); q,k,v: [B, L, h*d_k] (possibly strided
    views of one fused projection).  allrank/models/transformer.py:137-156, :193-203."""

    @staticmethod
    def forward(ctx, q, k, v, key_pad_mask, h, p_drop=0.0, seed=0):
        L.require_device(q, k, v, key_pad_mask)
        B, SL, d = q.shape
        dk = d // h
        for t in (q, k, v):
            if t.dtype != torch.float32 or t.stride(2) != 1 or t.stride(0) != SL * t.stride(1):
                raise ValueError("q/k/v must be fp32 [B,L,h*d_k] with unit inner stride and dense slates")
        rs = q.stride(1)
        if k.stride(1) != rs or v.stride(1) != rs:
            raise ValueError("q, k, v must share one row stride")
        mask = key_pad_mask.to(torch.uint8).contiguous()
        o = torch.empty((B, SL, d), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, h, SL), dtype=torch.float32, device=q.device)
        L.check(L.lib().ltrx_mha_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), B, SL, h, dk, rs, L.ptr(o), d, L.ptr(lse),
                                     float(p_drop), int(seed) & 0xFFFFFFFF, None, None, None, L.stream_of(q)), "mha_fwd")
        ctx.save_for_backward(q, k, v, mask, o, lse)
        ctx.h = h
        ctx.p_drop, ctx.seed = float(p_drop), int(seed) & 0xFFFFFFFF
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, mask, o, lse = ctx.saved_tensors
        B, SL, d = o.shape
        h = ctx.h
        dk_ = d // h
        do = L.f32c(do)
        # one fused [B, L, 3d] gradient buffer: dq | dk | dv are column slices (what a fused QKV projection consumes)
        dqkv = torch.empty((B, SL, 3 * d), dtype=torch.float32, device=o.device)
        dq, dkk, dv = dqkv[:, :, 0:d], dqkv[:, :, d:2 * d], dqkv[:, :, 2 * d:3 * d]
        lib = L.lib()
        ws = L.workspace(lib.ltrx_mha_bwd_workspace_bytes(B, SL, h), o)
        L.check(lib.ltrx_mha_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(o), L.ptr(do), L.ptr(lse), B, SL, h, dk_,
                                 q.stride(1), d, L.ptr(dq), L.ptr(dkk), L.ptr(dv), 3 * d, ctx.p_drop, ctx.seed, None, None, None, L.ptr(ws),
                                 L.stream_of(o)), "mha_bwd")
        return dq, dkk, dv, None, None, None, None


def attention(q, k, v, key_pad_mask, h, p_drop=0.0, seed=None):
    """fused masked self-attention; with p_drop > 0 the softmax probabilities are dropped out inside the kernel (the seed
    is drawn from torch's CPU generator unless given, so torch.manual_seed controls it)."""
    if p_drop and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _AttentionFn.apply(q, k, v, key_pad_mask, h, float(p_drop or 0.0), int(seed or 0))


def mfma_selftest(A, Bm):
    """D = A[32,2] @ B[2,32] through one MFMA with the lane layout the attention kernels assume."""
    D = torch.empty((32, 32), dtype=torch.float32, device=A.device)
    L.check(L.lib().ltrx_selftest_mfma32x32x2(L.ptr(L.f32c(A)), L.ptr(L.f32c(Bm)), L.ptr(D), L.stream_of(A)), "selftest")
    return D
