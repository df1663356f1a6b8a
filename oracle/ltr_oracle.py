"""CPU oracle for the listwise-LTR hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``allrank_amd``) never imports it and has no CPU fallback.

What this is: a numpy restatement of the *algorithms* of allegro/allRank's losses and metrics
(reference mounted at /root/reference; citations are ``file:line`` into it), plus hand-derived
analytic gradients (the reference gets its gradients from torch autograd, which this oracle does
not use).  Every function takes/returns numpy arrays; arithmetic runs in ``dtype`` (float32 by
default, like the reference; float64 is available as a "ground truth" for tolerance studies).

Parity pinning (see tests/test_oracle_pinned.py, tests/golden/):
  * the literal known-answer constants of the reference's own tests
    (tests/losses/test_{approxndcg,lambdaloss,listmle,listnet,neuralndcg,ndcg}.py);
  * golden vectors produced by importing the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz), values AND autograd gradients.

Tie policy: every sort here is *stable descending* (lower original index first
among equals).  torch's CPU sort is not stable for L>16, so "bit-exact indices" is defined under
this policy; the golden generator patches ``Tensor.sort`` to ``stable=True``.
"""
import numpy as np

DEFAULT_EPS = 1e-10          # allrank/models/losses/__init__.py:1
PADDED_Y_VALUE = -1          # allrank/data/dataset_loading.py:15

LAMBDA_SCHEMES = (None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme",
                  "rankNet_scheme", "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme")


def _f(x, dtype):
    return np.asarray(x, dtype=dtype)


def stable_argsort_desc(x):
    """Stable descending argsort along the last axis (ties: lower index first)."""
    return np.argsort(-x, axis=-1, kind="stable")


def _softmax_rows(x):
    m = np.max(x, axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=-1, keepdims=True)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------------------------
# ListNet  (allrank/models/losses/listNet.py:8-30)
# ----------------------------------------------------------------------------------------------
def listnet(y_pred, y_true, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, dtype=np.float32):
    """returns (loss, dloss/dy_pred, per_slate_loss)."""
    s = _f(y_pred, dtype).copy()
    t = _f(y_true, dtype).copy()
    B = s.shape[0]
    mask = t == pad                                   # listNet.py:20
    s[mask] = -np.inf                                 # :21
    t[mask] = -np.inf                                 # :22
    with np.errstate(invalid="ignore", divide="ignore"):
        P = _softmax_rows(s)                          # :24
        T = _softmax_rows(t)                          # :25
        logp = np.log(P + dtype(eps))                 # :27-28
        per = -np.sum(T * logp, axis=1)               # :30
    loss = np.mean(per)
    # d/ds_k = (1/B) [ P_k * sum_i T_i P_i/(P_i+eps) - T_k P_k/(P_k+eps) ]
    r = P / (P + dtype(eps))
    grad = (P * np.sum(T * r, axis=1, keepdims=True) - T * r) / dtype(B)
    grad[mask] = 0
    return dtype(loss), grad.astype(dtype), per.astype(dtype)


# ----------------------------------------------------------------------------------------------
# ListMLE  (allrank/models/losses/listMLE.py:7-38); the random column shuffle (:17) is an input.
# ----------------------------------------------------------------------------------------------
def listmle(y_pred, y_true, perm, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, dtype=np.float32):
    """returns (loss, dloss/dy_pred, per_slate_loss, order) where ``order[b, r]`` is the ORIGINAL item
    index placed at sorted position r (perm composed with the stable label sort)."""
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    B, L = s.shape
    perm = np.asarray(perm, dtype=np.int64)
    s_sh = s[:, perm]                                  # listMLE.py:18
    t_sh = t[:, perm]                                  # :19
    idx = stable_argsort_desc(t_sh)                    # :21 (stable policy)
    t_sorted = np.take_along_axis(t_sh, idx, axis=1)
    mask = t_sorted == pad                             # :23
    x = np.take_along_axis(s_sh, idx, axis=1).copy()   # :25
    x[mask] = -np.inf                                  # :26
    m = np.max(x, axis=1, keepdims=True)               # :28
    xm = x - m                                         # :30
    with np.errstate(invalid="ignore", divide="ignore"):
        e = np.exp(xm)
        cums = np.cumsum(e[:, ::-1], axis=1, dtype=dtype)[:, ::-1]   # :32
        obs = np.log(cums + dtype(eps)) - xm           # :34
    obs[mask] = 0                                      # :36
    per = np.sum(obs, axis=1)
    loss = np.mean(per)                                # :38
    # gradient w.r.t. xm:  g_k = e_k * sum_{i<=k, i valid} 1/(C_i+eps) - 1   (valid k)
    with np.errstate(invalid="ignore", divide="ignore"):
        inv = np.where(mask, 0, 1.0 / (cums + dtype(eps))).astype(dtype)
    pref = np.cumsum(inv, axis=1, dtype=dtype)
    g = np.where(mask, 0, e * pref - 1).astype(dtype)
    # the max-shift: xm = x - x[argmax]  ->  argmax receives -sum(g)
    am = np.argmax(x, axis=1)
    g[np.arange(B), am] -= np.sum(g, axis=1)
    g /= dtype(B)
    order = np.take_along_axis(np.broadcast_to(perm, (B, L)), idx, axis=1)
    grad = np.zeros_like(s)
    np.put_along_axis(grad, order, g, axis=1)
    return dtype(loss), grad.astype(dtype), per.astype(dtype), order


# ----------------------------------------------------------------------------------------------
# ApproxNDCG  (allrank/models/losses/approxNDCG.py:7-53)
# ----------------------------------------------------------------------------------------------
def approxndcg(y_pred, y_true, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, alpha=1.0, dtype=np.float32):
    """returns (loss, dloss/dy_pred, per_slate_approx_ndcg)."""
    s = _f(y_pred, dtype).copy()
    t = _f(y_true, dtype).copy()
    B, L = s.shape
    pm = t == pad                                      # approxNDCG.py:22
    s[pm] = -np.inf
    t[pm] = -np.inf
    ip = stable_argsort_desc(s)                        # :27
    s_sorted = np.take_along_axis(s, ip, axis=1)
    t_sorted = -np.sort(-t, axis=1, kind="stable")     # :28
    tsp = np.take_along_axis(t, ip, axis=1)            # :31
    with np.errstate(invalid="ignore"):
        td = tsp[:, :, None] - tsp[:, None, :]         # :32
    pmask = np.isfinite(td)                            # :33
    pmask[:, np.arange(L), np.arange(L)] = False       # :34
    tsp = np.maximum(tsp, 0)                           # :37
    t_sorted = np.maximum(t_sorted, 0)                 # :38
    pos = np.arange(1, L + 1, dtype=dtype)
    D = np.log2(dtype(1) + pos)[None, :]               # :42
    maxdcg = np.maximum(np.sum((np.power(dtype(2), t_sorted) - 1) / D, axis=-1), dtype(eps))   # :43
    G = (np.power(dtype(2), tsp) - 1) / maxdcg[:, None]                                       # :44
    with np.errstate(invalid="ignore"):
        sd = s_sorted[:, :, None] - s_sorted[:, None, :]   # :47
    sd[~pmask] = 0                                     # :48
    sig = _sigmoid(-dtype(alpha) * sd).astype(dtype)
    c = np.maximum(sig, dtype(eps))
    pm_f = pmask.astype(dtype)
    apos = 1 + np.sum(pm_f * c, axis=-1)               # :49
    aD = np.log2(1 + apos)                             # :50
    andcg = np.sum(G / aD, axis=-1)                    # :51
    loss = -np.mean(andcg)                             # :53
    # ---- analytic gradient (sorted space) ----
    # loss_b = -sum_i G_i / log2(1+p_i);  dloss_b/dp_i = G_i / (log2(1+p_i)^2 (1+p_i) ln2) =: w_i
    w = G / (aD * aD * (1 + apos) * dtype(np.log(2.0)))
    # c_ij = max(sigmoid(z_ij), eps), z_ij = -alpha (s_i - s_j); dc_ij/ds_i = -alpha sig', dc_ij/ds_j = +alpha sig'
    dsig = sig * (1 - sig) * (sig >= dtype(eps)) * pm_f
    g_sorted = -dtype(alpha) * w * np.sum(dsig, axis=2) + dtype(alpha) * np.sum(w[:, :, None] * dsig, axis=1)
    g_sorted = g_sorted / dtype(B)
    grad = np.zeros_like(s)
    np.put_along_axis(grad, ip, g_sorted.astype(dtype), axis=1)
    grad[pm] = 0
    return dtype(loss), grad.astype(dtype), andcg.astype(dtype)


# ----------------------------------------------------------------------------------------------
# LambdaLoss  (allrank/models/losses/lambdaLoss.py:7-114)
# ----------------------------------------------------------------------------------------------
def _lambda_weights(scheme, G, D, mu, tsp, dtype):
    """weights [B,L,L] (or scalar 1) in sorted-by-prediction space. lambdaLoss.py:84-114."""
    B, L = G.shape
    if scheme is None or scheme == "rankNet_scheme":
        return dtype(1.0)
    if scheme == "ndcgLoss1_scheme":                   # :84-85
        return np.broadcast_to((G / D)[:, :, None], (B, L, L)).astype(dtype)

    def ndcg2():
        pos = np.arange(1, L + 1)
        delta = np.abs(pos[:, None] - pos[None, :])    # :89-90
        Dv = D[0]
        deltas = np.abs(np.power(np.abs(Dv[delta - 1]), dtype(-1.0)) - np.power(np.abs(Dv[np.minimum(delta, L - 1)]), dtype(-1.0)))
        # NOTE: D[0, delta_idxs] with delta == L never occurs (max delta = L-1); min() is a no-op guard.
        deltas[np.arange(L), np.arange(L)] = 0         # :92
        return deltas[None, :, :] * np.abs(G[:, :, None] - G[:, None, :])   # :94

    def lrank():
        return np.abs(np.power(D[:, :, None], dtype(-1.0)) - np.power(D[:, None, :], dtype(-1.0))) * \
            np.abs(G[:, :, None] - G[:, None, :])      # :97-98

    if scheme == "ndcgLoss2_scheme":
        return ndcg2().astype(dtype)
    if scheme == "lambdaRank_scheme":
        return lrank().astype(dtype)
    if scheme == "ndcgLoss2PP_scheme":                 # :101-102
        return (dtype(mu) * ndcg2() + lrank()).astype(dtype)
    if scheme == "rankNetWeightedByGTDiff_scheme":     # :109-110
        return np.abs(tsp[:, :, None] - tsp[:, None, :]).astype(dtype)
    if scheme == "rankNetWeightedByGTDiffPowed_scheme":   # :113-114
        return np.abs(np.power(tsp[:, :, None], 2) - np.power(tsp[:, None, :], 2)).astype(dtype)
    raise ValueError("unknown weighing scheme %r" % (scheme,))


def lambdaloss(y_pred, y_true, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, weighing_scheme=None, k=None, sigma=1.0, mu=10.0,
               reduction="sum", reduction_log="binary", dtype=np.float32):
    """returns (loss, dloss/dy_pred, n_selected_pairs, order) -- order = stable-desc argsort of masked preds."""
    if reduction_log not in ("natural", "binary"):
        raise ValueError("Reduction logarithm base can be either natural or binary")      # lambdaLoss.py:72
    if reduction not in ("sum", "mean"):
        raise ValueError("Reduction method can be either sum or mean")                     # :79
    s = _f(y_pred, dtype).copy()
    t = _f(y_true, dtype).copy()
    B, L = s.shape
    pm = t == pad
    s[pm] = -np.inf                                    # :29
    t[pm] = -np.inf                                    # :30
    ip = stable_argsort_desc(s)                        # :33
    s_sorted = np.take_along_axis(s, ip, axis=1)
    t_sorted = -np.sort(-t, axis=1, kind="stable")     # :34
    tsp = np.take_along_axis(t, ip, axis=1)            # :37
    with np.errstate(invalid="ignore"):
        td = tsp[:, :, None] - tsp[:, None, :]         # :38
    pmask = np.isfinite(td)                            # :39
    if weighing_scheme != "ndcgLoss1_scheme":
        with np.errstate(invalid="ignore"):
            pmask = pmask & (td > 0)                   # :41-42
    kmask = np.zeros((L, L), dtype=bool)               # :44-45  ([:None,:None] == all)
    kmask[:k, :k] = True
    tsp = np.maximum(tsp, 0)                           # :48
    t_sorted = np.maximum(t_sorted, 0)                 # :49
    pos = np.arange(1, L + 1, dtype=dtype)
    D = np.log2(dtype(1) + pos)[None, :]               # :53
    maxdcg = np.maximum(np.sum(((np.power(dtype(2), t_sorted) - 1) / D)[:, :k], axis=-1), dtype(eps))   # :54
    G = (np.power(dtype(2), tsp) - 1) / maxdcg[:, None]                                               # :55
    W = _lambda_weights(weighing_scheme, G, D, mu, tsp, dtype)                                          # :58-61
    with np.errstate(invalid="ignore"):
        sd = np.clip(s_sorted[:, :, None] - s_sorted[:, None, :], dtype(-1e8), dtype(1e8))            # :64
    sel = pmask & kmask[None]
    sd = np.where(sel, sd, 0)       # unselected pairs never reach the sum (:75); avoids NaN noise
    sig = _sigmoid(dtype(sigma) * sd).astype(dtype)
    q = np.maximum(sig, dtype(eps))
    with np.errstate(invalid="ignore", divide="ignore"):
        u = np.power(q, W).astype(dtype)               # :66
    lnb = dtype(1.0) if reduction_log == "natural" else dtype(np.log(2.0))
    # log(clamp(q**W, eps)) evaluated as max(W*log(q), log(eps)): the same function, but without the fp32
    # cancellation of log(1 - tiny) that pow-then-log suffers when W ~ 1e-4 (the torch reference happens to
    # average that noise out; numpy's float32 pow does not -- 3e-5 relative at L=240 with ndcgLoss2).
    with np.errstate(invalid="ignore", divide="ignore"):
        losses = np.maximum(W * np.log(q), np.log(dtype(eps))).astype(dtype) / lnb     # :66-72
    n_sel = int(np.sum(sel))
    total = np.sum(np.where(sel, losses, 0), dtype=dtype)
    denom = dtype(1.0) if reduction == "sum" else dtype(max(n_sel, 1))
    loss = -total / denom                              # :74-79
    # ---- analytic gradient: d losses_ij / d(sd_ij) = W * sigma * (1 - sig) / ln(base)  where neither clamp is active
    live = sel & (sig >= dtype(eps)) & (u >= dtype(eps))
    dl = np.where(live, W * dtype(sigma) * (1 - sig) / lnb, 0).astype(dtype)
    g_sorted = -(np.sum(dl, axis=2) - np.sum(dl, axis=1)) / denom
    grad = np.zeros_like(s)
    np.put_along_axis(grad, ip, g_sorted.astype(dtype), axis=1)
    grad[pm] = 0
    if reduction == "mean" and n_sel == 0:
        loss = dtype(np.nan)                           # torch.mean of an empty selection
    return dtype(loss), grad.astype(dtype), n_sel, ip


# ----------------------------------------------------------------------------------------------
# metrics.dcg / ndcg  (allrank/models/metrics.py:7-77)
# ----------------------------------------------------------------------------------------------
def dcg(y_pred, y_true, ats=None, powered=True, pad=PADDED_Y_VALUE, dtype=np.float32, gain_function=None):
    """returns (dcg[B, len(ats)], order[B, L]) -- order = stable-desc argsort of the masked predictions.
    ``gain_function`` (metrics.py:42,67): any elementwise callable on the labels gathered in predicted order -- padded items were
    set to label 0 first (:35), so they carry gain_function(0) at the tail positions."""
    s = _f(y_pred, dtype).copy()
    t = _f(y_true, dtype).copy()
    B, L = t.shape
    if ats is None:
        ats = [L]                                      # metrics.py:58-59
    ats = [min(int(a), L) for a in ats]                # :60
    mask = t == pad                                    # :32
    s[mask] = -np.inf                                  # :34
    t[mask] = 0                                        # :35
    order = stable_argsort_desc(s)                     # :37
    tsp = np.take_along_axis(t, order, axis=1)         # :38
    disc = (dtype(1) / np.log2(np.arange(L, dtype=dtype) + dtype(2.0))).astype(dtype)   # :64-65
    if gain_function is not None:
        gains = np.asarray(gain_function(tsp), dtype=dtype)                             # :67
    else:
        gains = (np.power(dtype(2), tsp) - 1) if powered else tsp
    dg = (gains * disc)[:, :max(ats)]                  # :69
    cum = np.cumsum(dg, axis=1, dtype=dtype)           # :71
    return cum[:, np.asarray(ats) - 1].astype(dtype), order    # :73-75


def ndcg(y_pred, y_true, ats=None, pad=PADDED_Y_VALUE, filler_value=1.0, dtype=np.float32, gain_function=None):
    """returns (ndcg[B, len(ats)], order[B, L]).  metrics.py:7-28 (idcg == 0 -> filler_value); the ideal ranking sorts by LABEL
    (dcg(y_true, y_true, ...), :21) whatever the gain function."""
    idcg, _ = dcg(y_true, y_true, ats, True, pad, dtype, gain_function)       # :21
    d, order = dcg(y_pred, y_true, ats, True, pad, dtype, gain_function)
    with np.errstate(invalid="ignore", divide="ignore"):
        out = d / idcg                                         # :22
    out[idcg == 0] = filler_value                              # :23-24
    return out.astype(dtype), order


# ----------------------------------------------------------------------------------------------
# NeuralSort + Sinkhorn + NeuralNDCG  (loss_utils.py:8-67, neuralNDCG.py:10-136), deterministic variant
# ----------------------------------------------------------------------------------------------
def deterministic_neural_sort(s, tau, mask, dtype=np.float32):
    """s [B,L], mask [B,L] bool -> P_hat [B,L,L] (row = soft rank, col = item).  loss_utils.py:34-67."""
    s = _f(s, dtype)
    B, n = s.shape
    sm = np.where(mask, dtype(-1e8), s)                                    # :48
    A = np.abs(sm[:, :, None] - sm[:, None, :])                            # :49
    A = np.where(mask[:, :, None] | mask[:, None, :], 0, A)                # :50
    Bsum = np.sum(A, axis=2, dtype=dtype)                                  # :52  (A @ ones(n,n): every column = row sum)
    nvalid = n - mask.sum(axis=1)                                          # :54
    i = np.arange(n)
    scaling = np.where(i[None, :] < nvalid[:, None], nvalid[:, None] + 1 - 2 * (i[None, :] + 1), 0).astype(dtype)   # :54-57
    s0 = np.where(mask, 0, s)                                              # :59
    C = s0[:, :, None] * scaling[:, None, :]                               # :60  C[b, j, i]
    P_max = np.transpose(C - Bsum[:, :, None], (0, 2, 1))                  # :62  P_max[b, i, j]
    both = mask[:, :, None] & mask[:, None, :]
    either = mask[:, :, None] | mask[:, None, :]
    P_max = np.where(either, -np.inf, P_max)                               # :63
    P_max = np.where(both, 1.0, P_max).astype(dtype)                       # :64
    with np.errstate(invalid="ignore"):
        return _softmax_rows(P_max / dtype(tau)).astype(dtype)             # :65-66


def sinkhorn_scaling(mat, mask=None, tol=1e-6, max_iter=50, dtype=np.float32, return_norms=False):
    """loss_utils.py:8-31.  The early exit (:25) is batch-global.  Returns the scaled matrices (and, when
    return_norms, the list of (axis, clamped normaliser) pairs actually applied, for the backward pass)."""
    mat = _f(mat, dtype).copy()
    if mask is not None:
        either = mask[:, None, :] | mask[:, :, None]
        both = mask[:, None, :] & mask[:, :, None]
        mat = np.where(either, 0, mat)                                     # :17
        mat = np.where(both, 1, mat).astype(dtype)                         # :18
    norms = []
    for _ in range(max_iter):                                              # :20
        c = np.maximum(mat.sum(axis=1, keepdims=True, dtype=dtype), dtype(DEFAULT_EPS))
        mat = mat / c                                                      # :21
        r = np.maximum(mat.sum(axis=2, keepdims=True, dtype=dtype), dtype(DEFAULT_EPS))
        mat = mat / r                                                      # :22
        norms.append((c, r))
        if np.max(np.abs(mat.sum(axis=2) - 1.0)) < tol and np.max(np.abs(mat.sum(axis=1) - 1.0)) < tol:   # :25
            break
    if mask is not None:
        mat = np.where(either, 0, mat).astype(dtype)                       # :28-29
    if return_norms:
        return mat, norms
    return mat


def _sinkhorn_backward(P0, norms, gbar, dtype):
    """Reverse-mode through sinkhorn_scaling given the initial (masked) matrix P0, the recorded clamped
    normalisers and the adjoint ``gbar`` of the output.  Re-runs the forward to recover the intermediates."""
    eps = dtype(DEFAULT_EPS)
    states = []
    mat = P0
    for (c, r) in norms:
        y1 = mat / c
        y2 = y1 / r
        states.append((y1, y2))
        mat = y2
    g = gbar
    for (c, r), (y1, y2) in zip(reversed(norms), reversed(states)):
        # y2 = y1 / r,  r = max(rowsum(y1), eps)
        d = np.sum(g * y2, axis=2, keepdims=True, dtype=dtype) * (r > eps)
        g = (g - d) / r
        # y1 = x / c,   c = max(colsum(x), eps)
        d = np.sum(g * y1, axis=1, keepdims=True, dtype=dtype) * (c > eps)
        g = (g - d) / c
    return g.astype(dtype)


def _neural_core(s, t, sort_mask, true_mask, temperature, powered_relevancies, k, transposed, max_iter, tol, idcg, cnt,
                 dtype):
    """shared body of neuralndcg / neuralndcg_stochastic over a batch of (possibly perturbed) slates.
    ``sort_mask`` pads NeuralSort + Sinkhorn (neuralNDCG.py:36-42), ``true_mask`` pads the read-out (:46-47); they differ
    only in the stochastic variant (mask.repeat_interleave vs the sample-major view, loss_utils.py:107-108).
    returns (ndcg values [N], d loss / d s [N, L], iterations run), loss = -sum(values)/cnt."""
    N, L = s.shape
    P_hat = deterministic_neural_sort(s, temperature, sort_mask, dtype)    # :38
    P_s, norms = sinkhorn_scaling(P_hat, sort_mask, tol=tol, max_iter=max_iter, dtype=dtype, return_norms=True)   # :41-42
    either_s = sort_mask[:, :, None] | sort_mask[:, None, :]
    if transposed:
        # :102-124: no read-out mask at all -- the gains are the RAW labels (padding -1 -> 2^-1 - 1 or -1), which meet the zero
        # columns Sinkhorn's own mask leaves (identical to the plain variant unless sort_mask != true_mask)
        either_t = np.zeros_like(either_s)
        g = (np.power(dtype(2), t) - 1) if powered_relevancies else t.copy()
    else:
        either_t = true_mask[:, :, None] | true_mask[:, None, :]
        P_s = np.where(either_t, 0, P_s).astype(dtype)                     # :46
        tm = np.where(true_mask, 0, t)                                     # :47
        g = (np.power(dtype(2), tm) - 1) if powered_relevancies else tm   # :48-49
    disc = (dtype(1) / np.log2(np.arange(L, dtype=dtype) + dtype(2.0))).astype(dtype)   # :52 / :108
    disc_k = disc.copy()
    disc_k[k:] = 0                                                         # :55 / :111
    gt = np.einsum("bij,bj->bi", P_s, g).astype(dtype)                     # :51
    dgain = np.sum(gt * disc_k[None, :], axis=1, dtype=dtype)              # :53,:60-61
    nd = dgain / (idcg + dtype(DEFAULT_EPS))                               # :61
    zero = idcg == 0                                                       # :62
    nd = np.where(zero, 0, nd)                                             # :63
    n_iter = len(norms)
    if cnt == 0:
        return nd, np.zeros_like(s), n_iter
    # ---- backward ----
    coef = np.where(zero, 0, -1.0 / (cnt * (idcg + dtype(DEFAULT_EPS)))).astype(dtype)    # dloss/d dgain_b
    gbar = coef[:, None, None] * disc_k[None, :, None] * g[:, None, :]     # adjoint of P_s[b,i,j]
    gbar = np.where(either_t | either_s, 0, gbar).astype(dtype)            # final masks (:46 and loss_utils.py:28-29)
    either = either_s
    mask = sort_mask
    P0 = np.where(either, 0, P_hat)
    P0 = np.where(mask[:, :, None] & mask[:, None, :], 1, P0).astype(dtype)
    g0 = _sinkhorn_backward(P0, norms, gbar, dtype)
    g0 = np.where(either, 0, g0)                                           # masked_fill at loss_utils.py:17-18 blocks the gradient
    # softmax over j of P_max/tau
    inner = np.sum(g0 * P_hat, axis=2, keepdims=True, dtype=dtype)
    gz = P_hat * (g0 - inner) / dtype(temperature)                         # adjoint of P_max[b,i,j]
    gz = np.where(either, 0, gz).astype(dtype)
    # P_max[b,i,j] = scaling_i * s_j - Bsum_j ;  Bsum_j = sum_k |s_j - s_k| over valid pairs
    nvalid = L - mask.sum(axis=1)
    i = np.arange(L)
    scaling = np.where(i[None, :] < nvalid[:, None], nvalid[:, None] + 1 - 2 * (i[None, :] + 1), 0).astype(dtype)
    grad = np.einsum("bij,bi->bj", gz, scaling)
    Q = gz.sum(axis=1, dtype=dtype)                                        # [B, j]
    sm = np.where(mask, 0, s)
    sgn = np.sign(sm[:, :, None] - sm[:, None, :])
    sgn = np.where(either, 0, sgn)
    grad = grad - Q * sgn.sum(axis=2) + np.einsum("bj,bjm->bm", Q, sgn)
    grad = np.where(mask, 0, grad)
    return nd, grad.astype(dtype), n_iter


def neuralndcg(y_pred, y_true, pad=PADDED_Y_VALUE, temperature=1.0, powered_relevancies=True, k=None,
               transposed=False, max_iter=50, tol=1e-6, dtype=np.float32):
    """Deterministic NeuralNDCG / NeuralNDCG-transposed.  returns (loss, dloss/dy_pred, n_iter_run).

    neuralNDCG.py:10-70 (plain) and :73-136 (transposed).  The two variants compute the same number
    (sum_i disc_i sum_j P_ij g_j); they differ in (a) the transposed variant exposes max_iter/tol, and
    (b) with powered_relevancies=False the transposed variant STILL normalises by the powered idcg
    (neuralNDCG.py:126, reference quirk, kept).
    """
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    B, L = s.shape
    if k is None:
        k = L                                                              # :29-30
    mask = t == pad                                                        # :32
    idcg_powered = powered_relevancies or transposed                       # :55-58 vs :118-126
    idcg, _ = dcg(t, t, [k], idcg_powered, pad, dtype)
    idcg = idcg[:, 0]
    zero = idcg == 0
    cnt = dtype((~zero).sum())
    nd, grad, n_iter = _neural_core(s, t, mask, mask, temperature, powered_relevancies, k, transposed, max_iter, tol, idcg, cnt,
                                    dtype)
    if zero.all():
        return dtype(0.0), np.zeros_like(s), n_iter                        # :66-67
    loss = -np.sum(nd, dtype=dtype) / cnt                                  # :69-70
    return dtype(loss), grad.astype(dtype), n_iter


def neuralndcg_stochastic(y_pred, y_true, gumbel, pad=PADDED_Y_VALUE, temperature=1.0, powered_relevancies=True, k=None,
                          transposed=False, beta=0.1, log_scores=True, max_iter=50, tol=1e-6, dtype=np.float32):
    """Stochastic NeuralNDCG (neuralNDCG.py:35-37 + loss_utils.py:84-112) for a GIVEN Gumbel draw ``gumbel``
    [n_samples, B, L] (the reference: sample_gumbel -> torch.rand, loss_utils.py:80).  returns (loss, dloss/dy_pred).
    Reference quirk kept: the perturbed copy i = sample*B + b is sorted under the padding mask of slate i // n_samples
    (mask.repeat_interleave, loss_utils.py:108, neuralNDCG.py:41) and read out under the mask of slate b (:44-47)."""
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    G = _f(gumbel, dtype)
    S, B, L = G.shape
    if k is None:
        k = L
    mask = t == pad
    smin = s.min()
    s_pos = s + np.abs(smin)                                               # loss_utils.py:102
    w = np.ones_like(s)
    if log_scores:
        w = (dtype(1) / (s_pos + dtype(1e-10))).astype(dtype)
        s_pos = np.log(s_pos + dtype(1e-10))                               # :104-105
    sp = (s_pos[None] + dtype(beta) * G).reshape(S * B, L).astype(dtype)   # :103,:107
    idx = np.arange(S * B)
    sort_mask = mask[idx // S]                                             # :108
    true_mask = np.tile(mask, (S, 1))                                      # neuralNDCG.py:46 (mask[None] broadcast)
    tt = np.tile(t, (S, 1))
    idcg_powered = powered_relevancies or transposed
    idcg, _ = dcg(t, t, [k], idcg_powered, pad, dtype)
    idcg = idcg[:, 0]
    zero = idcg == 0
    if zero.all():
        return dtype(0.0), np.zeros_like(s)
    cnt = dtype((~zero).sum() * S)                                         # :69
    nd, gp, _ = _neural_core(sp, tt, sort_mask, true_mask, temperature, powered_relevancies, k, transposed, max_iter, tol,
                             np.tile(idcg, S), cnt, dtype)
    loss = -np.sum(nd, dtype=dtype) / cnt
    gs = gp.reshape(S, B, L).sum(axis=0) * w                               # d loss / d s_pos . d s_pos / d s (elementwise part)
    grad = gs.copy()
    ties = s == smin                                                       # through |min(s)| (:102); torch's min() backward
    grad[ties] += np.sign(smin) * gs.sum(dtype=dtype) / ties.sum()         # spreads the gradient evenly over tied minima
    return dtype(loss), grad.astype(dtype)


# ------------------------------------------------------------------------------------------------------------------
# pointwise / pairwise losses and MRR
# ------------------------------------------------------------------------------------------------------------------
def ranknet(y_pred, y_true, pad=PADDED_Y_VALUE, weight_by_diff=False, weight_by_diff_powed=False, dtype=np.float32):
    """rankNet.py:31-79.  returns (loss, grad): BCEWithLogits(target 1, weight) over pairs y_i > y_j, mean over all pairs."""
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    valid = t != pad
    pair = valid[:, :, None] & valid[:, None, :] & (t[:, :, None] > t[:, None, :])        # :57-61
    d = (s[:, :, None] - s[:, None, :]).astype(dtype)
    if weight_by_diff:
        w = np.abs(t[:, :, None] - t[:, None, :])                                          # :64-66
    elif weight_by_diff_powed:
        w = np.abs(t[:, :, None] ** 2 - t[:, None, :] ** 2)                                # :67-70
    else:
        w = np.ones_like(d)
    n = pair.sum()
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        l = np.where(pair, w * (np.maximum(-d, 0) + np.log1p(np.exp(-np.abs(d)))), 0).astype(dtype)
        loss = dtype(l.sum(dtype=np.float64) / n) if n else dtype(np.nan)                  # :79 (mean of an empty selection)
        gpair = np.where(pair, -w / (1 + np.exp(d)), 0)                                    # d l_ij / d d_ij
    grad = (gpair.sum(axis=2) - gpair.sum(axis=1)) / max(int(n), 1)
    return loss, grad.astype(dtype)


def _bce_terms(p, tgt, dtype):
    with np.errstate(divide="ignore"):
        l = -(tgt * np.maximum(np.log(p), -100) + (1 - tgt) * np.maximum(np.log(1 - p), -100))     # torch BCELoss
    g = (p - tgt) / np.maximum((1 - p) * p, 1e-12)
    return l.astype(dtype), g.astype(dtype)


def bce(y_pred, y_true, pad=PADDED_Y_VALUE, dtype=np.float32):
    """bce.py:8-32: y_pred are probabilities.  returns (loss, grad)."""
    p = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    valid = t != pad
    l, g = _bce_terms(p, t, dtype)
    l = np.where(valid, l, 0)                                                              # :24-25
    cnt = (valid.sum(axis=1) > 0).sum()                                                    # :28
    loss = dtype(l.sum(dtype=np.float64) / cnt)
    return loss, (np.where(valid, g, 0) / cnt).astype(dtype)


def ordinal(y_pred, y_true, n, pad=PADDED_Y_VALUE, dtype=np.float32):
    """ordinal.py:25-50: y_pred [B, L, n] probabilities.  returns (loss, grad [B, L, n])."""
    p = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    valid = t != pad
    tgt = (t[:, :, None] >= np.arange(1, n + 1, dtype=dtype)[None, None, :]).astype(dtype)  # :17-20
    l, g = _bce_terms(p, tgt, dtype)
    l = np.where(valid[:, :, None], l, 0)
    cnt = valid.sum()                                                                      # :46 (documents with a valid ordinal)
    loss = dtype(l.sum(dtype=np.float64) / cnt)
    return loss, (np.where(valid[:, :, None], g, 0) / cnt).astype(dtype)


def pointwise_rmse(y_pred, y_true, no_of_levels, pad=PADDED_Y_VALUE, dtype=np.float32):
    """pointwise.py:6-32.  returns (loss, grad)."""
    p = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    valid = t != pad
    e = np.where(valid, t - dtype(no_of_levels) * p, 0).astype(dtype)                      # :23-26
    nv = valid.sum(axis=1).astype(dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        rm = np.sqrt((e * e).sum(axis=1, dtype=dtype) / nv)                                # :28-30
        loss = dtype(rm.mean(dtype=np.float64))
        grad = -dtype(no_of_levels) * e / (nv * rm)[:, None] / dtype(len(t))
    return loss, np.where(valid, grad, 0).astype(dtype)


def binary_listnet(y_pred, y_true, eps=DEFAULT_EPS, pad=PADDED_Y_VALUE, dtype=np.float32):
    """binary_listNet.py:8-33.  returns (loss, grad)."""
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    mask = t == pad
    s = np.where(mask, -np.inf, s)
    t = np.where(mask, 0, t)
    norm = t.sum(axis=1, keepdims=True)
    norm = np.where(norm == 0, 1, norm)                                                    # :24
    T = (t / norm).astype(dtype)
    P = _softmax_rows(s).astype(dtype)
    loss = dtype(np.mean(-np.sum(T * np.log(P + dtype(eps)), axis=1, dtype=dtype), dtype=np.float64))
    r = P / (P + dtype(eps))
    grad = (P * np.sum(T * r, axis=1, keepdims=True) - T * r) / dtype(len(s))
    return loss, np.where(mask, 0, grad).astype(dtype)


def mrr(y_pred, y_true, ats=None, pad=PADDED_Y_VALUE, dtype=np.float32):
    """metrics.py:80-113.  returns [B, len(ats)]."""
    s = _f(y_pred, dtype)
    t = _f(y_true, dtype)
    B, L = s.shape
    if ats is None:
        ats = [L]
    mask = t == pad
    s = np.where(mask, -np.inf, s)
    t = np.where(mask, 0, t)
    order = np.stack([stable_argsort_desc(r) for r in s])
    ts = np.take_along_axis(t, order, axis=1)
    vals = ts.max(axis=1)
    idx = ts.argmax(axis=1).astype(dtype)                                                  # first maximum (:100)
    res = (dtype(1) / (idx + dtype(1)))[:, None].repeat(len(ats), axis=1)
    if vals.sum() == 0:                                                                    # :108-109 (0-dim mask)
        res[:] = 0
    within = (idx[:, None] < np.asarray(ats, dtype=dtype)[None, :]).astype(dtype)
    return (res * within).astype(dtype)


