"""torch.autograd bindings of the scoring-model kernels of libltrx.so (custom LayerNorm, fused masked attention, nn.Linear).

PyTorch owns the tensors and the autograd graph edges; all arithmetic happens in the HIP kernels behind the C ABI
(include/ltrx.h).  Device tensors only, fp32, no CPU fallback.
"""
import contextlib
import threading

import torch

from . import _lib as L

# ------------------------------------------------------------------------------------------------------------------
# Arithmetic of the nn.Module path.  The C ABI keeps no mode of its own (every ltrx_* call carries its precision / mode
# argument, include/ltrx.h); what the autograd bindings below pass is a PYTHON-side setting: a process default
# (set_linear_backend / set_attention_mode) that a thread can override for a region with ``arithmetic(...)`` -- so the
# reference's DataParallel replica threads, or a trainer beside a validation pass, never see each other's choice.
#   attention mode 1 (default): three bf16 products per fp32 product on the bf16 MFMA (fp32-class, like the projections);
#   0: exact fp32 MFMA everywhere (the strict reference); 2: ONE bf16 product (throughput mode, outside the parity contract).
# ------------------------------------------------------------------------------------------------------------------
_DEFAULT = {"linear": "split_bf16", "attention": 1}
_tls = threading.local()


def _setting(key):
    ov = getattr(_tls, "override", None)
    if ov and ov.get(key) is not None:
        return ov[key]
    return _DEFAULT[key]


@contextlib.contextmanager
def arithmetic(linear=None, attention=None):
    """thread-local override of the nn.Module path's arithmetic for the enclosed region, e.g.
    ``with ops.arithmetic(linear="hipblaslt", attention=0): ...`` = exact-fp32 projections and attention (the strict bar)."""
    if linear is not None and linear not in ("split_bf16", "hipblaslt"):
        raise ValueError("linear backend must be split_bf16 or hipblaslt")
    if attention is not None and attention not in (0, 1, 2):
        raise ValueError("attention mode must be 0 (exact fp32), 1 (split-bf16) or 2 (plain bf16)")
    prev = getattr(_tls, "override", None)
    cur = dict(prev or {})
    if linear is not None:
        cur["linear"] = linear
    if attention is not None:
        cur["attention"] = attention
    _tls.override = cur
    try:
        yield
    finally:
        _tls.override = prev


def set_attention_mode(mode):
    """process default of the attention arithmetic of the nn.Module path (see ``arithmetic`` for a scoped, per-thread override)"""
    if mode not in (0, 1, 2):
        raise ValueError("attention mode must be 0, 1 or 2")
    _DEFAULT["attention"] = int(mode)


def get_attention_mode():
    return int(_setting("attention"))


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
        mode = get_attention_mode()
        L.check(L.lib().ltrx_mha_fwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), B, SL, h, dk, rs, L.ptr(o), d, L.ptr(lse),
                                     float(p_drop), int(seed) & 0xFFFFFFFF, None, None, None, mode, L.stream_of(q)), "mha_fwd")
        ctx.save_for_backward(q, k, v, mask, o, lse)
        ctx.h, ctx.mode = h, mode
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
        ws = L.workspace(lib.ltrx_mha_bwd_workspace_bytes(B, SL, h, dk_, ctx.mode), o)
        L.check(lib.ltrx_mha_bwd(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(mask), L.ptr(o), L.ptr(do), L.ptr(lse), B, SL, h, dk_,
                                 q.stride(1), d, L.ptr(dq), L.ptr(dkk), L.ptr(dv), 3 * d, ctx.p_drop, ctx.seed, None, None, None, ctx.mode, L.ptr(ws),
                                 L.stream_of(o)), "mha_bwd")
        return dq, dkk, dv, None, None, None, None


class _AttentionPackedFn(torch.autograd.Function):
    """the same attention on ONE packed projection qkv [B, L, 3 h d_k] (q | k | v column blocks, what MultiHeadedAttention's fused
    QKV GEMM produces): one input, one gradient -- autograd does not have to rebuild d qkv from three zero-padded slice gradients
    (3 fills + 3 full-size adds per layer in the sliced form)."""

    @staticmethod
    def forward(ctx, qkv, key_pad_mask, h, p_drop=0.0, seed=0):
        L.require_device(qkv, key_pad_mask)
        qkv = L.f32c(qkv)
        B, SL, d3 = qkv.shape
        d = d3 // 3
        mask = key_pad_mask.to(torch.uint8).contiguous()
        o = torch.empty((B, SL, d), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, h, SL), dtype=torch.float32, device=qkv.device)
        mode = get_attention_mode()
        L.check(L.lib().ltrx_mha_fwd(L.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, L.ptr(mask), B, SL, h, d // h, d3, L.ptr(o),
                                     d, L.ptr(lse), float(p_drop), int(seed) & 0xFFFFFFFF, None, None, None, mode, L.stream_of(qkv)), "mha_fwd")
        ctx.save_for_backward(qkv, mask, o, lse)
        ctx.h, ctx.p_drop, ctx.seed, ctx.mode = h, float(p_drop), int(seed) & 0xFFFFFFFF, mode
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, mask, o, lse = ctx.saved_tensors
        B, SL, d = o.shape
        h = ctx.h
        do = L.f32c(do)
        dqkv = torch.empty((B, SL, 3 * d), dtype=torch.float32, device=o.device)
        lib = L.lib()
        ws = L.workspace(lib.ltrx_mha_bwd_workspace_bytes(B, SL, h, d // h, ctx.mode), o)
        L.check(lib.ltrx_mha_bwd(L.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, L.ptr(mask), L.ptr(o), L.ptr(do), L.ptr(lse),
                                 B, SL, h, d // h, 3 * d, d, L.ptr(dqkv), dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d, 3 * d,
                                 ctx.p_drop, ctx.seed, None, None, None, ctx.mode, L.ptr(ws), L.stream_of(o)), "mha_bwd")
        return dqkv, None, None, None, None


def attention_packed(qkv, key_pad_mask, h, p_drop=0.0, seed=None):
    """``attention(qkv[..., :d], qkv[..., d:2d], qkv[..., 2d:], ...)`` with one input and one gradient (see _AttentionPackedFn)"""
    if p_drop and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _AttentionPackedFn.apply(qkv, key_pad_mask, h, float(p_drop or 0.0), int(seed or 0))


def attention(q, k, v, key_pad_mask, h, p_drop=0.0, seed=None):
    """fused masked self-attention; with p_drop > 0 the softmax probabilities are dropped out inside the kernel (the seed
    is drawn from torch's CPU generator unless given, so torch.manual_seed controls it)."""
    if p_drop and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _AttentionFn.apply(q, k, v, key_pad_mask, h, float(p_drop or 0.0), int(seed or 0))


# ------------------------------------------------------------------------------------------------------------------
# nn.Linear on the split-bf16 GEMMs (the drop-in nn.Module path: allrank's own fit() / loss_batch, model.score(), validation)
# ------------------------------------------------------------------------------------------------------------------
def set_linear_backend(name):
    """process default: "split_bf16" -- ``linear()`` runs ltrx_gemm_nt / ltrx_gemm_tn (fp32-accurate three-product bf16 MFMA GEMMs,
    the arithmetic of the explicit step); "hipblaslt": torch's F.linear (exact-fp32 library GEMMs, 2-2.5x slower on MI355X).
    ``arithmetic(linear=...)`` overrides it for one thread and region."""
    if name not in ("split_bf16", "hipblaslt"):
        raise ValueError("linear backend must be split_bf16 or hipblaslt")
    _DEFAULT["linear"] = name


def _split_linear():
    return _setting("linear") == "split_bf16"


class _LinearFn(torch.autograd.Function):
    """y = act(x w^T + b)   (nn.Linear, model.py:35-44, transformer.py:193-203, 221-227; act 1 = the ReLU of :227 in the epilogue)"""

    @staticmethod
    def forward(ctx, x, w, b, act):
        L.require_device(x, w, b)
        N, K = w.shape
        x2 = L.f32c(x).reshape(-1, K)
        w = L.f32c(w)
        b = L.f32c(b) if b is not None else None
        M = x2.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        L.check(L.lib().ltrx_gemm_nt(L.ptr(x2), K, L.ptr(w), K, None, L.ptr(y), N, M, N, K, L.ptr(b), int(act), None, 0, 0.0, 0, None, 0,
                                     0, L.stream_of(x2)), "gemm_nt(linear fwd)")
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.has_bias, ctx.act, ctx.xshape = b is not None, int(act), x.shape
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        N, K = w.shape
        M = x2.shape[0]
        dy2 = L.f32c(dy).reshape(M, N)
        if ctx.act:                                              # ReLU backward
            dy2 = dy2 * (y > 0)
        lib = L.lib()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if N % 4 == 0:
                wT = w.t().contiguous()                          # [K, N]: the NT kernel wants both operands contraction-contiguous
                dx = torch.empty((M, K), dtype=torch.float32, device=dy2.device)
                L.check(lib.ltrx_gemm_nt(L.ptr(dy2), N, L.ptr(wT), N, None, L.ptr(dx), K, M, K, N, None, 0, None, 0, 0.0, 0, None, 0,
                                         0, L.stream_of(dy2)), "gemm_nt(linear dgrad)")
            else:
                dx = dy2 @ w
            dx = dx.view(ctx.xshape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty((N, K), dtype=torch.float32, device=dy2.device)
            db = torch.empty(N, dtype=torch.float32, device=dy2.device) if ctx.has_bias else None
            ws = torch.empty(max(int(lib.ltrx_gemm_tn_workspace_bytes(M, N, K)), 64), dtype=torch.uint8, device=dy2.device)
            L.check(lib.ltrx_gemm_tn(L.ptr(dy2), N, L.ptr(x2), K, L.ptr(dw), L.ptr(db), M, N, K, 0, 0, L.ptr(ws), L.stream_of(dy2)),
                    "gemm_tn(linear wgrad)")
        return dx, dw, db, None


def linear(x, w, b=None, act=0):
    """F.linear(x, w, b) (followed by ReLU when act == 1) for device tensors on the split-bf16 GEMMs; shapes the kernels do not
    take (in_features not a multiple of 4) and the "hipblaslt" backend go through torch."""
    if not _split_linear() or not x.is_cuda or w.shape[1] % 4 != 0 or x.dtype != torch.float32:
        y = torch.nn.functional.linear(x, w, b)
        return torch.relu(y) if act else y
    return _LinearFn.apply(x, w, b, act)


class _FFNFn(torch.autograd.Function):
    """w_2(dropout(relu(w_1 x)))   (PositionwiseFeedForward.forward, transformer.py:221-227) as ONE autograd node: ReLU and
    dropout live in the epilogue of the first GEMM (mask regenerated from (seed, element index), nothing stored), and the
    backward applies the ReLU(+dropout) mask in the epilogue of the input-gradient GEMM of w_2 -- the block of the explicit step."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, p, seed):
        L.require_device(x, w1, b1, w2, b2)
        F_, K = w1.shape
        N = w2.shape[0]
        x2 = L.f32c(x).reshape(-1, K)
        w1, b1, w2, b2 = L.f32c(w1), L.f32c(b1), L.f32c(w2), L.f32c(b2)
        M = x2.shape[0]
        lib, st = L.lib(), L.stream_of(x2)
        r = torch.empty((M, F_), dtype=torch.float32, device=x.device)
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        L.check(lib.ltrx_gemm_nt(L.ptr(x2), K, L.ptr(w1), K, None, L.ptr(r), F_, M, F_, K, L.ptr(b1), 1, None, 0, float(p), int(seed), None, 0, 0, st),
                "gemm_nt(ffn w_1)")
        L.check(lib.ltrx_gemm_nt(L.ptr(r), F_, L.ptr(w2), F_, None, L.ptr(y), N, M, N, F_, L.ptr(b2), 0, None, 0, 0.0, 0, None, 0, 0, st),
                "gemm_nt(ffn w_2)")
        ctx.save_for_backward(x2, w1, w2, r)
        ctx.p, ctx.xshape = float(p), x.shape
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w1, w2, r = ctx.saved_tensors
        F_, K = w1.shape
        N = w2.shape[0]
        M = x2.shape[0]
        dy2 = L.f32c(dy).reshape(M, N)
        lib, st, dev = L.lib(), L.stream_of(dy2), dy2.device

        def wgrad(a, b_, n, k):
            gw = torch.empty((n, k), dtype=torch.float32, device=dev)
            gb = torch.empty(n, dtype=torch.float32, device=dev)
            ws = torch.empty(max(int(lib.ltrx_gemm_tn_workspace_bytes(M, n, k)), 64), dtype=torch.uint8, device=dev)
            L.check(lib.ltrx_gemm_tn(L.ptr(a), n, L.ptr(b_), k, L.ptr(gw), L.ptr(gb), M, n, k, 0, 0, L.ptr(ws), st), "gemm_tn(ffn wgrad)")
            return gw, gb

        dw2, db2 = wgrad(dy2, r, N, F_)
        w2T = w2.t().contiguous()                                # [F, N]
        dr = torch.empty((M, F_), dtype=torch.float32, device=dev)
        # d r = (dy w_2) * [r > 0] / (1 - p): r is the post-ReLU, post-dropout activation, so its sign pattern IS the combined mask
        L.check(lib.ltrx_gemm_nt(L.ptr(dy2), N, L.ptr(w2T), N, None, L.ptr(dr), F_, M, F_, N, None, 2, L.ptr(r), F_, ctx.p, 0, None, 0, 0, st),
                "gemm_nt(ffn dgrad w_2)")
        dw1, db1 = wgrad(dr, x2, F_, K)
        dx = None
        if ctx.needs_input_grad[0]:
            w1T = w1.t().contiguous()                            # [K, F]
            dx = torch.empty((M, K), dtype=torch.float32, device=dev)
            L.check(lib.ltrx_gemm_nt(L.ptr(dr), F_, L.ptr(w1T), F_, None, L.ptr(dx), K, M, K, F_, None, 0, None, 0, 0.0, 0, None, 0, 0, st),
                    "gemm_nt(ffn dgrad w_1)")
            dx = dx.view(ctx.xshape)
        return dx, dw1, db1, dw2, db2, None, None


def feed_forward(x, w1, b1, w2, b2, p_drop=0.0, seed=None):
    """w_2(dropout_p(relu(w_1 x))) for device tensors; ``seed`` keys the dropout mask (default: drawn from torch's generator)."""
    dims_ok = w1.shape[1] % 4 == 0 and w1.shape[0] % 4 == 0 and w2.shape[0] % 4 == 0
    if not _split_linear() or not x.is_cuda or not dims_ok or x.dtype != torch.float32:
        h = torch.relu(torch.nn.functional.linear(x, w1, b1))
        return torch.nn.functional.linear(torch.nn.functional.dropout(h, p_drop, p_drop > 0), w2, b2)
    if p_drop and seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    return _FFNFn.apply(x, w1, b1, w2, b2, float(p_drop), int(seed or 0))


class _ScoreHeadFn(torch.autograd.Function):
    """OutputLayer with d_output == 1 (model.py:111-117): s[m] = <x[m, :], w> + b as one pass over x (a GEMV is HBM-bound)"""

    @staticmethod
    def forward(ctx, x, w, b):
        L.require_device(x, w, b)
        D = x.shape[-1]
        x2 = L.f32c(x).reshape(-1, D)
        w, b = L.f32c(w).reshape(-1), L.f32c(b).reshape(-1)
        M = x2.shape[0]
        sc = torch.empty(M, dtype=torch.float32, device=x.device)
        L.check(L.lib().ltrx_score_head_fwd(L.ptr(x2), L.ptr(w), L.ptr(b), M, D, L.ptr(sc), L.stream_of(x2)), "score_head_fwd")
        ctx.save_for_backward(x2, w)
        ctx.xshape, ctx.wshape, ctx.bshape = x.shape, None, None
        return sc.view(x.shape[:-1])

    @staticmethod
    def backward(ctx, ds):
        x2, w = ctx.saved_tensors
        M, D = x2.shape
        ds = L.f32c(ds).reshape(-1)
        lib = L.lib()
        dx = torch.empty_like(x2)
        dw = torch.empty(D, dtype=torch.float32, device=x2.device)
        db = torch.empty(1, dtype=torch.float32, device=x2.device)
        ws = torch.empty(max(int(lib.ltrx_score_head_bwd_workspace_bytes(M, D)), 64), dtype=torch.uint8, device=x2.device)
        L.check(lib.ltrx_score_head_bwd(L.ptr(ds), L.ptr(x2), L.ptr(w), M, D, L.ptr(dx), L.ptr(dw), L.ptr(db), L.ptr(ws), L.stream_of(x2)),
                "score_head_bwd")
        return dx.view(ctx.xshape), dw.view(1, D), db


def score_head(x, w, b):
    """nn.Linear(d, 1)(x).squeeze(-1) for device tensors: w [1, d], b [1]"""
    if not _split_linear() or not x.is_cuda or x.dtype != torch.float32 or w.shape[0] != 1 or b is None:
        return torch.nn.functional.linear(x, w, b).squeeze(-1)
    return _ScoreHeadFn.apply(x, w, b)


def mfma_selftest(A, Bm):
    """D = A[32,2] @ B[2,32] through one MFMA with the lane layout the attention kernels assume."""
    D = torch.empty((32, 32), dtype=torch.float32, device=A.device)
    L.check(L.lib().ltrx_selftest_mfma32x32x2(L.ptr(L.f32c(A)), L.ptr(L.f32c(Bm)), L.ptr(D), L.stream_of(A)), "selftest")
    return D
