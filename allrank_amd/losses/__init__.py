"""MI355X-native listwise losses with the plugin signatures of ``allrank.models.losses``.

Every function here has the signature, defaults, error behaviour and semantics of its namesake in the reference
(allrank/models/losses/__init__.py:3-12; selected by name at allrank/main.py:83) and returns a 0-dim tensor that
supports ``.backward()`` / ``.item()``.  The arithmetic runs in ONE fused HIP kernel per loss (forward and
d loss / d y_pred together; libltrx.so, include/ltrx.h); the autograd node only scales the stored gradient.
Inputs are never mutated.  Device tensors only -- there is no CPU fallback.
"""
import contextlib
import threading

import torch

from .. import _lib as L
from .. import sharding

DEFAULT_EPS = 1e-10        # allrank/models/losses/__init__.py:1
PADDED_Y_VALUE = -1        # allrank/data/dataset_loading.py:15

__all__ = ["DEFAULT_EPS", "PADDED_Y_VALUE", "listNet", "listMLE", "approxNDCGLoss", "lambdaLoss", "neuralNDCG",
           "neuralNDCG_transposed", "sinkhorn_iterations_used", "rankNet", "rankNet_weightByGTDiff",
           "rankNet_weightByGTDiff_pow", "bce", "ordinal", "with_ordinals", "pointwise_rmse", "binary_listNet"]

_SCHEMES = {None: 0, "ndcgLoss1_scheme": 1, "ndcgLoss2_scheme": 2, "lambdaRank_scheme": 3, "ndcgLoss2PP_scheme": 4,
            "rankNet_scheme": 5, "rankNetWeightedByGTDiff_scheme": 6, "rankNetWeightedByGTDiffPowed_scheme": 7}


class _FusedLoss(torch.autograd.Function):
    """The kernel already produced d loss / d y_pred; backward is a scale by the incoming gradient."""

    @staticmethod
    def forward(ctx, y_pred, loss, grad):
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def _prep(y_pred, y_true):
    if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
        raise ValueError("y_pred and y_true must both be [batch_size, slate_length]")
    L.require_device(y_pred, y_true)
    yp = L.f32c(y_pred.detach())
    yt = L.f32c(y_true.detach())
    need_grad = torch.is_grad_enabled() and y_pred.requires_grad
    B, SL = yp.shape
    loss = torch.empty(1, dtype=torch.float32, device=yp.device)
    grad = torch.empty_like(yp) if need_grad else None
    return yp, yt, B, SL, loss, grad, need_grad


def _finish(y_pred, loss, grad, need_grad):
    if need_grad:
        return _FusedLoss.apply(y_pred, loss, grad.to(y_pred.dtype))
    return loss.view(())


def listNet(y_pred, y_true, eps=DEFAULT_EPS, padded_value_indicator=PADDED_Y_VALUE):
    """ListNet (allrank/models/losses/listNet.py:8-30): -mean_b sum_i softmax(y_true)_i log(softmax(y_pred)_i + eps)."""
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_listnet_workspace_bytes(B, SL), yp)
    L.check(lib.ltrx_listnet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(eps), float(padded_value_indicator),
                                     sharding.batch_divisor(B), L.ptr(loss), None, L.ptr(grad), L.ptr(ws),
                                     L.stream_of(yp)), "listnet")
    return _finish(y_pred, loss, grad, ng)


def listMLE(y_pred, y_true, eps=DEFAULT_EPS, padded_value_indicator=PADDED_Y_VALUE, perm=None, generator=None):
    """ListMLE (allrank/models/losses/listMLE.py:7-38).  The reference shuffles the columns with
    ``torch.randperm(L)`` from the global CPU generator (listMLE.py:17) for randomised tie resolution; so does this
    function unless ``perm`` (an int64 permutation of range(L), any device) is given.  Ties among equal labels are
    then resolved by a STABLE sort in the shuffled order."""
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    if perm is None:
        perm = torch.randperm(SL, generator=generator)              # CPU generator, like the reference
    perm = perm.to(device=yp.device, dtype=torch.int64).contiguous()
    if perm.numel() != SL:
        raise ValueError("perm must be a permutation of range(slate_length)")
    lib = L.lib()
    ws = L.workspace(lib.ltrx_listmle_workspace_bytes(B, SL), yp)
    L.check(lib.ltrx_listmle_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(perm), B, SL, float(eps), float(padded_value_indicator),
                                     sharding.batch_divisor(B), L.ptr(loss), None, L.ptr(grad), None, L.ptr(ws),
                                     L.stream_of(yp)), "listmle")
    return _finish(y_pred, loss, grad, ng)


def approxNDCGLoss(y_pred, y_true, eps=DEFAULT_EPS, padded_value_indicator=PADDED_Y_VALUE, alpha=1.):
    """ApproxNDCG (allrank/models/losses/approxNDCG.py:7-53); no truncation, sigmoid temperature ``alpha``."""
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_approxndcg_workspace_bytes(B, SL), yp)
    L.check(lib.ltrx_approxndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(eps), float(padded_value_indicator),
                                        float(alpha), sharding.batch_divisor(B), L.ptr(loss), None, L.ptr(grad),
                                        L.ptr(ws), L.stream_of(yp)), "approxndcg")
    return _finish(y_pred, loss, grad, ng)


def lambdaLoss(y_pred, y_true, eps=DEFAULT_EPS, padded_value_indicator=PADDED_Y_VALUE, weighing_scheme=None, k=None,
               sigma=1., mu=10., reduction="sum", reduction_log="binary"):
    """LambdaLoss framework (allrank/models/losses/lambdaLoss.py:7-114) with its 7 weighing schemes."""
    if weighing_scheme not in _SCHEMES:
        raise KeyError(weighing_scheme)                                  # reference: globals()[weighing_scheme] (:61)
    if reduction_log not in ("natural", "binary"):
        raise ValueError("Reduction logarithm base can be either natural or binary")   # lambdaLoss.py:72
    if reduction not in ("sum", "mean"):
        raise ValueError("Reduction method can be either sum or mean")                  # lambdaLoss.py:79
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_lambdaloss_workspace_bytes(B, SL), yp)
    kk = 0 if k is None else int(k)
    red = 0 if reduction == "sum" else 1
    lg = 0 if reduction_log == "binary" else 1
    ext = None
    if red == 1 and sharding.active():
        # global pair count first (loss-only pass), then the real pass normalised by it
        cnt = torch.empty(1, dtype=torch.float32, device=yp.device)
        L.check(lib.ltrx_lambdaloss_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(eps), float(padded_value_indicator),
                                            _SCHEMES[weighing_scheme], kk, float(sigma), float(mu), 0, lg, None,
                                            L.ptr(loss), L.ptr(cnt), None, None, L.ptr(ws), L.stream_of(yp)),
                "lambdaloss(count)")
        ext = sharding.allreduce_sum_(cnt)
    L.check(lib.ltrx_lambdaloss_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(eps), float(padded_value_indicator),
                                        _SCHEMES[weighing_scheme], kk, float(sigma), float(mu), red, lg, L.ptr(ext),
                                        L.ptr(loss), None, L.ptr(grad), None, L.ptr(ws), L.stream_of(yp)), "lambdaloss")
    return _finish(y_pred, loss, grad, ng)


_last_iters = {"t": None}
_neural_tls = threading.local()


@contextlib.contextmanager
def neural_kernel_path(path):
    """test hook (thread-local, scoped): 1 = run the general L2-streaming Sinkhorn kernels even where the register-resident
    ones apply (L <= 240); the value is passed as the ``path`` argument of every ltrx_neuralndcg_fwd_bwd call in the region."""
    prev = getattr(_neural_tls, "path", 0)
    _neural_tls.path = int(path)
    try:
        yield
    finally:
        _neural_tls.path = prev


def _neural_path():
    return getattr(_neural_tls, "path", 0)


def sinkhorn_iterations_used():
    """number of Sinkhorn iterations the last neuralNDCG* call ran (device->host sync; diagnostics only)."""
    t = _last_iters["t"]
    return None if t is None else int(t.item())


def _neural_call(yp, yt, idcg, cnt, kk, k_rows, padded_value_indicator, temperature, powered_relevancies, transposed,
                 max_iter, tol, ng):
    """one fused NeuralSort + Sinkhorn + value + gradient launch sequence over the slates of (yp, yt)"""
    lib = L.lib()
    B, SL = yp.shape
    ws = L.workspace(lib.ltrx_neuralndcg_workspace_bytes(B, SL, int(max_iter)), yp)
    loss = torch.empty(1, dtype=torch.float32, device=yp.device)
    grad = torch.empty_like(yp) if ng else None
    iters = torch.empty(1, dtype=torch.int32, device=yp.device)
    L.check(lib.ltrx_neuralndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(idcg), L.ptr(cnt), B, SL,
                                        float(padded_value_indicator), float(temperature),
                                        1 if powered_relevancies else 0, kk, L.ptr(k_rows), 1 if transposed else 0,
                                        int(max_iter), float(tol), L.ptr(loss), None, L.ptr(grad), L.ptr(iters), _neural_path(), L.ptr(ws),
                                        L.stream_of(yp)), "neuralndcg")
    _last_iters["t"] = iters
    return loss, grad


def sample_gumbel(samples_shape, device, eps=1e-10):
    """loss_utils.py:70-81"""
    U = torch.rand(samples_shape, device=device)
    return -torch.log(-torch.log(U + eps) + eps)


def _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, transposed,
            max_iter, tol, n_samples=32, beta=0.1, log_scores=True, gumbel=None):
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    kk = 0 if k is None else int(k)
    ws = L.workspace(lib.ltrx_neuralndcg_workspace_bytes(B, SL, int(max_iter)), yp)
    idcg = torch.empty(B, dtype=torch.float32, device=yp.device)
    cnt = torch.empty(1, dtype=torch.float32, device=yp.device)
    idcg_powered = 1 if (powered_relevancies or transposed) else 0      # neuralNDCG.py:55-58 vs :118-126
    st = L.stream_of(yp)
    L.check(lib.ltrx_neuralndcg_prepare(L.ptr(yt), B, SL, float(padded_value_indicator), kk, idcg_powered, L.ptr(idcg),
                                        L.ptr(cnt), L.ptr(ws), st), "neuralndcg_prepare")
    sharding.allreduce_sum_(cnt)                                         # global normaliser (neuralNDCG.py:69)
    if not stochastic:
        loss, grad = _neural_call(yp, yt, idcg, cnt, kk, None, padded_value_indicator, temperature, powered_relevancies,
                                  transposed, max_iter, tol, ng)
        return _finish(y_pred, loss, grad, ng)
    # ---- stochastic NeuralSort (loss_utils.py:84-112): n_samples Gumbel-perturbed copies of every slate ----
    # The perturbation is a handful of elementwise torch ops (autograd carries d s_perturb / d y_pred, including the path
    # through the batch-global min); the n_samples * B perturbed slates then go through the SAME fused kernels as one batch.
    S = int(n_samples)
    s = y_pred.to(torch.float32)
    s_pos = s + torch.abs(s.min())                                       # :102 (min over the whole batch, padded slots too)
    if gumbel is None:
        gumbel = sample_gumbel([S, B, SL, 1], device=yp.device)          # :103
    samples = float(beta) * gumbel.to(device=yp.device, dtype=torch.float32).reshape(S, B, SL)
    if log_scores:
        s_pos = torch.log(s_pos + 1e-10)                                 # :104-105
    s_pert = (s_pos.unsqueeze(0) + samples).reshape(S * B, SL)           # pseudo slate i = sample i // B of slate i % B
    # The reference sorts pseudo slate i under the padding mask of slate i // n_samples (mask.repeat_interleave, :108 and
    # neuralNDCG.py:41/:106) but reads the result out with the labels of slate i % B.  Reproduced exactly through the labels
    # handed to the kernel:
    #   plain (:44-51: rows and columns are masked by the TRUE slate's padding, padded gains are 0):
    #     padding where the sort mask pads; the true label where both are valid; label 0 (= gain 0) where only the sort
    #     mask is valid; ranks beyond the true slate's length carry no discount (k_rows).
    #   transposed (:116-124: no read-out mask, gains are the RAW labels, i.e. 2^-1 - 1 or -1 on truly padded items):
    #     the raw labels wherever the sort mask is valid, under a private padding sentinel.
    idx = torch.arange(S * B, device=yp.device)
    sort_src, true_src = idx // S, idx % B
    pad = float(padded_value_indicator)
    sort_pad = (yt[sort_src] == pad)
    true_pad = (yt[true_src] == pad)
    if transposed:
        pad_k = -1.0e30
        y_ps = torch.where(sort_pad, torch.full((), pad_k, device=yp.device), yt[true_src]).contiguous()
        k_rows = None
    else:
        pad_k = pad
        y_ps = torch.where(true_pad, torch.zeros((), device=yp.device), yt[true_src])
        y_ps = torch.where(sort_pad, torch.full((), pad, device=yp.device), y_ps).contiguous()
        k_rows = (~true_pad).sum(1).to(torch.int32).contiguous()
    cnt_s = cnt * float(S)                                               # neuralNDCG.py:69: (#idcg != 0) * n_samples
    sp = s_pert.detach().contiguous()
    ng2 = torch.is_grad_enabled() and s_pert.requires_grad
    loss, grad = _neural_call(sp, y_ps, idcg[true_src].contiguous(), cnt_s, kk, k_rows, pad_k, temperature,
                              powered_relevancies, transposed, max_iter, tol, ng2)
    return _finish(s_pert, loss, grad, ng2)


def neuralNDCG(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE, temperature=1., powered_relevancies=True, k=None,
               stochastic=False, n_samples=32, beta=0.1, log_scores=True, gumbel=None):
    """NeuralNDCG (allrank/models/losses/neuralNDCG.py:10-70): NeuralSort (deterministic, or stochastic with ``n_samples``
    Gumbel-perturbed copies per slate) + Sinkhorn (50 its, tol 1e-6).  ``gumbel`` ([n_samples, B, L, 1], optional) injects
    the Gumbel noise the reference draws with torch.rand (loss_utils.py:80) -- parity tests pass the same draw to both."""
    return _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, False, 50, 1e-6,
                   n_samples, beta, log_scores, gumbel)


def neuralNDCG_transposed(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE, temperature=1.,
                          powered_relevancies=True, k=None, stochastic=False, n_samples=32, beta=0.1, log_scores=True,
                          max_iter=50, tol=1e-6, gumbel=None):
    """NeuralNDCG transposed (allrank/models/losses/neuralNDCG.py:73-136)."""
    return _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, True,
                   max_iter, tol, n_samples, beta, log_scores, gumbel)


# ----------------------------------------------------------------------------------------------------------------
# pointwise / pairwise losses
# ----------------------------------------------------------------------------------------------------------------
def _count_normalised(y_pred, y_true, name, launch, out_shape=None):
    """shared driver of the losses whose divisor is a batch-global count: under slate sharding a loss-only pass produces
    this rank's count, the counts are all-reduced, and the real pass divides by the global one (cf. lambdaLoss 'mean')."""
    L.require_device(y_pred, y_true)
    yp = L.f32c(y_pred.detach())
    yt = L.f32c(y_true.detach())
    ng = torch.is_grad_enabled() and y_pred.requires_grad
    loss = torch.empty(1, dtype=torch.float32, device=yp.device)
    grad = torch.empty_like(yp) if ng else None
    ext = None
    if sharding.active():
        cnt = torch.empty(1, dtype=torch.float32, device=yp.device)
        L.check(launch(yp, yt, None, loss, cnt, None), name + "(count)")
        ext = sharding.allreduce_sum_(cnt)
    L.check(launch(yp, yt, ext, loss, None, grad), name)
    return _finish(y_pred, loss, grad, ng)


def rankNet(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE, weight_by_diff=False, weight_by_diff_powed=False):
    """RankNet (allrank/models/losses/rankNet.py:31-79): BCE-with-logits over the pairs y_i > y_j, mean over all pairs of
    the batch; optional weights |y_i - y_j| or |y_i^2 - y_j^2|."""
    if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
        raise ValueError("y_pred and y_true must both be [batch_size, slate_length]")
    lib = L.lib()
    B, SL = y_pred.shape
    mode = 1 if weight_by_diff else (2 if weight_by_diff_powed else 0)           # rankNet.py:63-70 (elif order)

    def launch(yp, yt, ext, loss, cnt, grad):
        ws = L.workspace(lib.ltrx_ranknet_workspace_bytes(B, SL), yp)
        return lib.ltrx_ranknet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(padded_value_indicator), mode, L.ptr(ext), L.ptr(loss),
                                        L.ptr(cnt), L.ptr(grad), L.ptr(ws), L.stream_of(yp))
    return _count_normalised(y_pred, y_true, "ranknet", launch)


def rankNet_weightByGTDiff(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE):
    """rankNet.py:8-16"""
    return rankNet(y_pred, y_true, padded_value_indicator, weight_by_diff=True)


def rankNet_weightByGTDiff_pow(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE):
    """rankNet.py:19-28"""
    return rankNet(y_pred, y_true, padded_value_indicator, weight_by_diff=False, weight_by_diff_powed=True)


def bce(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE):
    """Binary cross-entropy on probabilities (allrank/models/losses/bce.py:8-32): sum over valid items / number of slates
    that contain a valid item."""
    if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
        raise ValueError("y_pred and y_true must both be [batch_size, slate_length]")
    lib = L.lib()
    B, SL = y_pred.shape

    def launch(yp, yt, ext, loss, cnt, grad):
        ws = L.workspace(lib.ltrx_bce_workspace_bytes(B, SL, 0), yp)
        return lib.ltrx_bce_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, 0, float(padded_value_indicator), L.ptr(ext), L.ptr(loss),
                                    L.ptr(cnt), L.ptr(grad), L.ptr(ws), L.stream_of(yp))
    return _count_normalised(y_pred, y_true, "bce", launch)


def with_ordinals(y, n, padded_value_indicator=PADDED_Y_VALUE):
    """ordinal.py:8-22: labels -> [B, L, n] ordinal targets (kept for API parity; the fused loss does not need it)."""
    one_to_n = torch.arange(start=1, end=n + 1, dtype=torch.float, device=y.device)
    unsq = y.unsqueeze(2).repeat(1, 1, n)
    out = (unsq >= one_to_n).type(torch.float)
    out[unsq == padded_value_indicator] = padded_value_indicator
    return out


def ordinal(y_pred, y_true, n, padded_value_indicator=PADDED_Y_VALUE):
    """Ordinal loss (allrank/models/losses/ordinal.py:25-50): y_pred [B, L, n] probabilities, BCE against the ordinal
    targets [y_true >= 1..n], summed / number of valid items."""
    n = int(n)
    if y_pred.dim() != 3 or y_pred.shape[:2] != y_true.shape or y_pred.shape[2] != n:
        raise ValueError("y_pred must be [batch_size, slate_length, n] and y_true [batch_size, slate_length]")
    lib = L.lib()
    B, SL = y_true.shape

    def launch(yp, yt, ext, loss, cnt, grad):
        ws = L.workspace(lib.ltrx_bce_workspace_bytes(B, SL, n), yp)
        return lib.ltrx_bce_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, n, float(padded_value_indicator), L.ptr(ext), L.ptr(loss),
                                    L.ptr(cnt), L.ptr(grad), L.ptr(ws), L.stream_of(yp))
    return _count_normalised(y_pred, y_true, "ordinal", launch)


def pointwise_rmse(y_pred, y_true, no_of_levels, padded_value_indicator=PADDED_Y_VALUE):
    """Pointwise RMSE (allrank/models/losses/pointwise.py:6-32): mean over slates of sqrt(mean (y - levels * p)^2)."""
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_pointwise_rmse_workspace_bytes(B, SL), yp)
    L.check(lib.ltrx_pointwise_rmse_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(no_of_levels), float(padded_value_indicator),
                                            sharding.batch_divisor(B), L.ptr(loss), L.ptr(grad), L.ptr(ws), L.stream_of(yp)),
            "pointwise_rmse")
    return _finish(y_pred, loss, grad, ng)


def binary_listNet(y_pred, y_true, eps=DEFAULT_EPS, padded_value_indicator=PADDED_Y_VALUE):
    """ListNet for binary labels (allrank/models/losses/binary_listNet.py:8-33): target distribution y / sum(y)."""
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_binary_listnet_workspace_bytes(B, SL), yp)
    L.check(lib.ltrx_binary_listnet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, float(eps), float(padded_value_indicator),
                                            sharding.batch_divisor(B), L.ptr(loss), L.ptr(grad), L.ptr(ws), L.stream_of(yp)),
            "binary_listnet")
    return _finish(y_pred, loss, grad, ng)


# ----------------------------------------------------------------------------------------------------------------
# Allocation-free launchers for the explicit training step (allrank_amd.engine.FusedTrainer): same kernels, persistent
# output / workspace buffers, no autograd node -> capturable in a hipGraph.
# ----------------------------------------------------------------------------------------------------------------
class FusedLoss(object):
    """``run(scores[B,L], y[B,L], batch_divisor)`` -> (loss[1], dloss/dscores[B,L]) on persistent device buffers."""

    def __init__(self, name, B, SL, device, **args):
        self.name, self.B, self.SL, self.args = name, B, SL, dict(args)
        lib = L.lib()
        self.loss = torch.zeros(1, dtype=torch.float32, device=device)
        self.grad = torch.zeros((B, SL), dtype=torch.float32, device=device)
        pad = float(args.get("padded_value_indicator", PADDED_Y_VALUE))
        eps = float(args.get("eps", DEFAULT_EPS))
        self.pad, self.eps = pad, eps
        if name == "listNet":
            nb = lib.ltrx_listnet_workspace_bytes(B, SL)
        elif name == "listMLE":
            nb = lib.ltrx_listmle_workspace_bytes(B, SL)
            self.perm = torch.arange(SL, dtype=torch.int64, device=device)
        elif name == "approxNDCGLoss":
            nb = lib.ltrx_approxndcg_workspace_bytes(B, SL)
        elif name == "lambdaLoss":
            if args.get("weighing_scheme") not in _SCHEMES:
                raise KeyError(args.get("weighing_scheme"))
            if args.get("reduction_log", "binary") not in ("natural", "binary"):
                raise ValueError("Reduction logarithm base can be either natural or binary")
            if args.get("reduction", "sum") not in ("sum", "mean"):
                raise ValueError("Reduction method can be either sum or mean")
            nb = lib.ltrx_lambdaloss_workspace_bytes(B, SL)
            self.cnt = torch.zeros(1, dtype=torch.float32, device=device)
        elif name in ("neuralNDCG", "neuralNDCG_transposed"):
            if args.get("stochastic", False):
                raise NotImplementedError("stochastic NeuralSort draws fresh noise per step: use the autograd Trainer")
            self.max_iter = int(args.get("max_iter", 50))
            nb = lib.ltrx_neuralndcg_workspace_bytes(B, SL, self.max_iter)
            self.idcg = torch.zeros(B, dtype=torch.float32, device=device)
            self.cnt = torch.zeros(1, dtype=torch.float32, device=device)
        elif name in ("rankNet", "rankNet_weightByGTDiff", "rankNet_weightByGTDiff_pow"):
            nb = lib.ltrx_ranknet_workspace_bytes(B, SL)
            self.cnt = torch.zeros(1, dtype=torch.float32, device=device)
            self.mode = (1 if (name == "rankNet_weightByGTDiff" or args.get("weight_by_diff")) else
                         2 if (name == "rankNet_weightByGTDiff_pow" or args.get("weight_by_diff_powed")) else 0)
        elif name in ("bce", "ordinal"):
            # bce: scores [B, SL] probabilities; ordinal: scores [B, SL, n] (n = the OutputLayer's d_output, ordinal.py:25-50)
            self.n_ord = int(args["n"]) if name == "ordinal" else 0
            nb = lib.ltrx_bce_workspace_bytes(B, SL, self.n_ord)
            self.cnt = torch.zeros(1, dtype=torch.float32, device=device)
            if self.n_ord:
                self.grad = torch.zeros((B, SL, self.n_ord), dtype=torch.float32, device=device)
        elif name == "binary_listNet":
            nb = lib.ltrx_binary_listnet_workspace_bytes(B, SL)
        elif name == "pointwise_rmse":
            nb = lib.ltrx_pointwise_rmse_workspace_bytes(B, SL)
            self.levels = float(args["no_of_levels"])
        else:
            raise KeyError("no fused launcher for loss %r" % (name,))
        self.ws = torch.empty(max(int(nb), 64), dtype=torch.uint8, device=device)

    def set_perm(self, perm):
        self.perm.copy_(perm.to(self.perm.device))

    def run(self, yp, yt, batch_divisor=None):
        lib = L.lib()
        B, SL, a = self.B, self.SL, self.args
        div = float(batch_divisor if batch_divisor is not None else B)
        st = L.stream_of(yp)
        n = self.name
        if n == "listNet":
            rc = lib.ltrx_listnet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.eps, self.pad, div, L.ptr(self.loss), None,
                                          L.ptr(self.grad), L.ptr(self.ws), st)
        elif n == "listMLE":
            rc = lib.ltrx_listmle_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(self.perm), B, SL, self.eps, self.pad, div,
                                          L.ptr(self.loss), None, L.ptr(self.grad), None, L.ptr(self.ws), st)
        elif n == "approxNDCGLoss":
            rc = lib.ltrx_approxndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.eps, self.pad, float(a.get("alpha", 1.)), div,
                                             L.ptr(self.loss), None, L.ptr(self.grad), L.ptr(self.ws), st)
        elif n == "lambdaLoss":
            k = a.get("k")
            red = 0 if a.get("reduction", "sum") == "sum" else 1
            lg = 0 if a.get("reduction_log", "binary") == "binary" else 1
            ext = None
            if red == 1 and sharding.active():
                rc = lib.ltrx_lambdaloss_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.eps, self.pad, _SCHEMES[a.get("weighing_scheme")],
                                                 0 if k is None else int(k), float(a.get("sigma", 1.)), float(a.get("mu", 10.)), 0, lg,
                                                 None, L.ptr(self.loss), L.ptr(self.cnt), None, None, L.ptr(self.ws), st)
                L.check(rc, "lambdaloss(count)")
                ext = sharding.allreduce_sum_(self.cnt)
            rc = lib.ltrx_lambdaloss_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.eps, self.pad, _SCHEMES[a.get("weighing_scheme")],
                                             0 if k is None else int(k), float(a.get("sigma", 1.)), float(a.get("mu", 10.)), red, lg,
                                             L.ptr(ext), L.ptr(self.loss), None, L.ptr(self.grad), None, L.ptr(self.ws), st)
        elif n.startswith("rankNet"):
            ext = None
            if sharding.active():
                L.check(lib.ltrx_ranknet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.pad, self.mode, None, L.ptr(self.loss),
                                                 L.ptr(self.cnt), None, L.ptr(self.ws), st), "ranknet(count)")
                ext = sharding.allreduce_sum_(self.cnt)
            rc = lib.ltrx_ranknet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.pad, self.mode, L.ptr(ext), L.ptr(self.loss), None,
                                          L.ptr(self.grad), L.ptr(self.ws), st)
        elif n in ("bce", "ordinal"):
            ext = None
            if sharding.active():
                L.check(lib.ltrx_bce_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.n_ord, self.pad, None, L.ptr(self.loss), L.ptr(self.cnt), None,
                                             L.ptr(self.ws), st), n + "(count)")
                ext = sharding.allreduce_sum_(self.cnt)
            rc = lib.ltrx_bce_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.n_ord, self.pad, L.ptr(ext), L.ptr(self.loss), None, L.ptr(self.grad),
                                      L.ptr(self.ws), st)
        elif n == "binary_listNet":
            rc = lib.ltrx_binary_listnet_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.eps, self.pad, div, L.ptr(self.loss),
                                                 L.ptr(self.grad), L.ptr(self.ws), st)
        elif n == "pointwise_rmse":
            rc = lib.ltrx_pointwise_rmse_fwd_bwd(L.ptr(yp), L.ptr(yt), B, SL, self.levels, self.pad, div, L.ptr(self.loss),
                                                 L.ptr(self.grad), L.ptr(self.ws), st)
        else:
            tr = n == "neuralNDCG_transposed"
            pw = bool(a.get("powered_relevancies", True))
            k = a.get("k")
            kk = 0 if k is None else int(k)
            rc = lib.ltrx_neuralndcg_prepare(L.ptr(yt), B, SL, self.pad, kk, 1 if (pw or tr) else 0, L.ptr(self.idcg),
                                             L.ptr(self.cnt), L.ptr(self.ws), st)
            L.check(rc, "neuralndcg_prepare")
            sharding.allreduce_sum_(self.cnt)
            rc = lib.ltrx_neuralndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(self.idcg), L.ptr(self.cnt), B, SL, self.pad,
                                             float(a.get("temperature", 1.)), 1 if pw else 0, kk, None, 1 if tr else 0, self.max_iter,
                                             float(a.get("tol", 1e-6)), L.ptr(self.loss), None, L.ptr(self.grad), None,
                                             _neural_path(), L.ptr(self.ws), st)
        L.check(rc, n)
        return self.loss, self.grad
