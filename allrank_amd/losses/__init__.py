"""MI355X-native listwise losses with the plugin signatures of ``allrank.models.losses``.

Every function here has the signature, defaults, error behaviour and semantics of its namesake in the reference
(allrank/models/losses/__init__.py:3-12; selected by name at allrank/main.py:83) and returns a 0-dim tensor that
supports ``.backward()`` / ``.item()``.  The arithmetic runs in ONE fused HIP kernel per loss (forward and
d loss / d y_pred together; libltrx.so, include/ltrx.h); the autograd node only scales the stored gradient.
Inputs are never mutated.  Device tensors only -- there is no CPU fallback.
"""
import torch

from .. import _lib as L
from .. import sharding

DEFAULT_EPS = 1e-10        # allrank/models/losses/__init__.py:1
PADDED_Y_VALUE = -1        # allrank/data/dataset_loading.py:15

__all__ = ["DEFAULT_EPS", "PADDED_Y_VALUE", "listNet", "listMLE", "approxNDCGLoss", "lambdaLoss", "neuralNDCG",
           "neuralNDCG_transposed", "sinkhorn_iterations_used"]

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


def sinkhorn_iterations_used():
    """number of Sinkhorn iterations the last neuralNDCG* call ran (device->host sync; diagnostics only)."""
    t = _last_iters["t"]
    return None if t is None else int(t.item())


def _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, transposed,
            max_iter, tol):
    if stochastic:
        raise NotImplementedError("stochastic NeuralSort (loss_utils.py:84-112) is not part of the MI355X hot path yet "
                                  "(SURVEY.md §8a row a19: deferred); use stochastic=False")
    yp, yt, B, SL, loss, grad, ng = _prep(y_pred, y_true)
    lib = L.lib()
    kk = 0 if k is None else int(k)
    ws = L.workspace(lib.ltrx_neuralndcg_workspace_bytes(B, SL, int(max_iter)), yp)
    idcg = torch.empty(B, dtype=torch.float32, device=yp.device)
    cnt = torch.empty(1, dtype=torch.float32, device=yp.device)
    iters = torch.empty(1, dtype=torch.int32, device=yp.device)
    idcg_powered = 1 if (powered_relevancies or transposed) else 0      # neuralNDCG.py:55-58 vs :118-126
    st = L.stream_of(yp)
    L.check(lib.ltrx_neuralndcg_prepare(L.ptr(yt), B, SL, float(padded_value_indicator), kk, idcg_powered, L.ptr(idcg),
                                        L.ptr(cnt), L.ptr(ws), st), "neuralndcg_prepare")
    sharding.allreduce_sum_(cnt)                                         # global normaliser (neuralNDCG.py:69)
    L.check(lib.ltrx_neuralndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(idcg), L.ptr(cnt), B, SL,
                                        float(padded_value_indicator), float(temperature),
                                        1 if powered_relevancies else 0, kk, 1 if transposed else 0, int(max_iter),
                                        float(tol), L.ptr(loss), None, L.ptr(grad), L.ptr(iters), L.ptr(ws), st),
            "neuralndcg")
    _last_iters["t"] = iters
    return _finish(y_pred, loss, grad, ng)


def neuralNDCG(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE, temperature=1., powered_relevancies=True, k=None,
               stochastic=False, n_samples=32, beta=0.1, log_scores=True):
    """NeuralNDCG (allrank/models/losses/neuralNDCG.py:10-70), deterministic NeuralSort + Sinkhorn (50 its, tol 1e-6)."""
    return _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, False, 50, 1e-6)


def neuralNDCG_transposed(y_pred, y_true, padded_value_indicator=PADDED_Y_VALUE, temperature=1.,
                          powered_relevancies=True, k=None, stochastic=False, n_samples=32, beta=0.1, log_scores=True,
                          max_iter=50, tol=1e-6):
    """NeuralNDCG transposed (allrank/models/losses/neuralNDCG.py:73-136)."""
    return _neural(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k, stochastic, True,
                   max_iter, tol)


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
                raise NotImplementedError("stochastic NeuralSort is not on the MI355X hot path yet")
            self.max_iter = int(args.get("max_iter", 50))
            nb = lib.ltrx_neuralndcg_workspace_bytes(B, SL, self.max_iter)
            self.idcg = torch.zeros(B, dtype=torch.float32, device=device)
            self.cnt = torch.zeros(1, dtype=torch.float32, device=device)
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
                                             float(a.get("temperature", 1.)), 1 if pw else 0, kk, 1 if tr else 0, self.max_iter,
                                             float(a.get("tol", 1e-6)), L.ptr(self.loss), None, L.ptr(self.grad), None,
                                             L.ptr(self.ws), st)
        L.check(rc, n)
        return self.loss, self.grad
