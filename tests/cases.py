"""Shared enumeration of the golden loss cases (tests/golden/losses_golden.npz, made by make_golden.py)."""
LAMBDA_SCHEMES = [None, "ndcgLoss1_scheme", "ndcgLoss2_scheme", "lambdaRank_scheme", "ndcgLoss2PP_scheme",
                  "rankNet_scheme", "rankNetWeightedByGTDiff_scheme", "rankNetWeightedByGTDiffPowed_scheme"]


def close(a, b, rtol=1e-5, atol=1e-5):
    """|a-b| <= atol + rtol*|b| -- the 1e-5 fp32 bar of BASELINE.json's north_star, relative for large values."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)      # e.g. lambdaLoss(reduction="mean") over an empty pair selection
    return bool(np.all(both_nan | (np.abs(a - b) <= atol + rtol * np.abs(b))))


def grad_close(g, gref, rtol=2e-4):
    """gradient check: max abs error relative to the largest reference gradient entry of the tensor."""
    import numpy as np
    g = np.asarray(g, dtype=np.float64)
    gref = np.asarray(gref, dtype=np.float64)
    scale = max(float(np.abs(gref).max()), 1e-6)
    return float(np.abs(g - gref).max()) <= rtol * scale + 1e-7


def iter_loss_cases(gold):
    """yields (name, kind, kwargs, s, y, ref_loss, ref_grad)"""
    n = int(gold["n_cases"])
    for ci in range(n):
        pre = "c%d." % ci
        s, y = gold[pre + "s"], gold[pre + "y"]
        yield (pre + "listnet", "listnet", {}, s, y, gold[pre + "listnet.loss"], gold[pre + "listnet.grad"])
        for a in (1.0, 2.5):
            k = pre + "approxndcg.a%g" % a
            yield (k, "approxndcg", dict(alpha=a), s, y, gold[k + ".loss"], gold[k + ".grad"])
        yield (pre + "listmle", "listmle", dict(perm=gold[pre + "listmle.perm"]), s, y,
               gold[pre + "listmle.loss"], gold[pre + "listmle.grad"])
        for si, sch in enumerate(LAMBDA_SCHEMES):
            for kk in (None, 5):
                for red, lg in (("sum", "binary"), ("mean", "natural")):
                    k = pre + "lambda.s%d.k%s.%s.%s" % (si, kk, red, lg)
                    yield (k, "lambdaloss", dict(weighing_scheme=sch, k=kk, reduction=red, reduction_log=lg, sigma=1.3, mu=7.0),
                           s, y, gold[k + ".loss"], gold[k + ".grad"])
        for tr in (False, True):
            for tau in (1.0, 0.1):
                for kk in (None, 5):
                    for pw in (True, False):
                        k = pre + "neural.t%d.tau%g.k%s.p%d" % (int(tr), tau, kk, int(pw))
                        yield (k, "neuralndcg", dict(transposed=tr, temperature=tau, k=kk, powered_relevancies=pw),
                               s, y, gold[k + ".loss"], gold[k + ".grad"])


# ---- SURVEY.md section 8f row 4 fixtures (tests/golden/extra_golden.npz, made by make_golden_extra.py) ----
STOCH = [dict(tr=False, tau=1.0, k=None, pw=True, log=True, beta=0.1), dict(tr=False, tau=0.5, k=5, pw=False, log=False, beta=0.3),
         dict(tr=True, tau=1.0, k=None, pw=True, log=True, beta=0.1), dict(tr=True, tau=2.0, k=7, pw=False, log=True, beta=1.0)]
N_ORD = 4


def iter_extra_cases(gold):
    """yields (name, kind, kwargs, y_pred, y_true, ref_loss, ref_grad) for the pointwise / pairwise losses"""
    for ci in range(int(gold["n_cases"])):
        pre = "c%d." % ci
        s, y = gold[pre + "s"], gold[pre + "y"]
        for m, kw in enumerate((dict(), dict(weight_by_diff=True), dict(weight_by_diff_powed=True))):
            yield (pre + "ranknet.m%d" % m, "ranknet", kw, s, y, gold[pre + "ranknet.m%d.loss" % m], gold[pre + "ranknet.m%d.grad" % m])
        yield (pre + "bce.nopad", "bce", {}, gold[pre + "pe"], gold[pre + "ybnp"], gold[pre + "bce.nopad.loss"], gold[pre + "bce.nopad.grad"])
        yield (pre + "bce.pad", "bce", {}, gold[pre + "p"], gold[pre + "yb"], gold[pre + "bce.pad.loss"], gold[pre + "bce.pad.grad"])
        yield (pre + "ordinal.nopad", "ordinal", dict(n=N_ORD), gold[pre + "p3e"], gold[pre + "ynp"],
               gold[pre + "ordinal.nopad.loss"], gold[pre + "ordinal.nopad.grad"])
        yield (pre + "ordinal.pad", "ordinal", dict(n=N_ORD), gold[pre + "p3"], y, gold[pre + "ordinal.pad.loss"], gold[pre + "ordinal.pad.grad"])
        yield (pre + "rmse", "pointwise_rmse", dict(no_of_levels=4), gold[pre + "p"], y, gold[pre + "rmse.loss"], gold[pre + "rmse.grad"])
        yield (pre + "blistnet", "binary_listnet", {}, s, gold[pre + "yb"], gold[pre + "blistnet.loss"], gold[pre + "blistnet.grad"])


def iter_stochastic_cases(gold):
    """yields (name, cfg, s, y, gumbel[S,B,L], ref_loss, ref_grad, strict_mask): ``strict_mask`` excludes the batch-minimum
    score(s) when log_scores is on -- there d log(s - min + 1e-10)/ds = 1e10 amplifies fp32 round-off in the reference too."""
    for ci in range(int(gold["n_cases"])):
        pre = "c%d." % ci
        s, y = gold[pre + "s"], gold[pre + "y"]
        for si, c in enumerate(STOCH):
            strict = (s != s.min()) if c["log"] else (s == s)
            yield (pre + "stoch%d" % si, c, s, y, gold[pre + "gumbel"][..., 0], gold[pre + "stoch%d.loss" % si],
                   gold[pre + "stoch%d.grad" % si], strict)


def row4_kats():
    """the known-answer tests of /root/reference/tests/losses/test_{ranknet,mrr,loss_ordinal,loss_pointwise,binary_listnet}.py
    restated as (kind, kwargs, y_pred, y_true, expected): expected values come from the same independent closed forms the
    reference tests use (BCE-with-logits of the score differences, cross-entropy by hand, ...)."""
    import math
    import numpy as np

    def bcel(ds, ws=None):                                     # BCEWithLogitsLoss(weight)(ds, ones), mean
        ws = ws or [1.0] * len(ds)
        return float(np.mean([w * math.log1p(math.exp(-d)) for d, w in zip(ds, ws)]))

    def xe(t, p):
        return -t * math.log(p) - (1 - t) * math.log(1 - p)

    def sm(v):
        e = np.exp(np.asarray(v, np.float64) - np.max(v))
        return e / e.sum()

    P = -1.0
    k = []
    # test_ranknet.py:29-115
    k += [("ranknet", {}, [0.5, 0.2], [1.0, 0.0], bcel([0.3])), ("ranknet", {}, [0.2, 0.5], [1.0, 0.0], bcel([-0.3])),
          ("ranknet", {}, [0.5, 0.2, 0.1], [1.0, 0.0, 0.0], bcel([0.3, 0.4])), ("ranknet", {}, [0.2, 0.5], [0.0, 1.0], bcel([0.3])),
          ("ranknet", {}, [0.2, 0.5], [0.0, 2.0], bcel([0.3])), ("ranknet", {}, [0.5, 0.2, 0.66], [1.0, 0.0, P], bcel([0.3])),
          ("ranknet", dict(weight_by_diff=True), [0.5, 0.2, 0.1], [2.0, 1.0, 0.0], bcel([0.3, 0.4, 0.1], [1.0, 2.0, 1.0])),
          ("ranknet", dict(weight_by_diff_powed=True), [0.5, 0.2, 0.1], [2.0, 1.0, 0.0], bcel([0.3, 0.4, 0.1], [3.0, 4.0, 1.0]))]
    # test_loss_ordinal.py:28-58 (n = 2)
    k += [("ordinal", dict(n=2), [[0.8, 0.6]], [1.0], xe(1, 0.8) + xe(0, 0.6)),
          ("ordinal", dict(n=2), [[0.8, 0.7], [0.4, 0.3], [0.2, 0.1]], [2.0, 1.0, 0.0],
           float(np.mean([xe(1, 0.8) + xe(1, 0.7), xe(1, 0.4) + xe(0, 0.3), xe(0, 0.2) + xe(0, 0.1)]))),
          ("ordinal", dict(n=2), [[0.8, 0.6], [0.2, 0.1]], [1.0, P], xe(1, 0.8) + xe(0, 0.6))]
    # test_loss_pointwise.py:16-47
    k += [("pointwise_rmse", dict(no_of_levels=1), [0.5, 0.2], [1.0, 0.0], math.sqrt(np.mean([0.25, 0.04]))),
          ("pointwise_rmse", dict(no_of_levels=1), [0.5, 0.2, 0.5], [1.0, 0.0, P], math.sqrt(np.mean([0.25, 0.04]))),
          ("pointwise_rmse", dict(no_of_levels=3), [0.5, 0.2, 0.7, 0.8], [1.0, 0.0, 2.0, 3.0],
           math.sqrt(np.mean([0.25, 0.36, 0.01, 0.36])))]
    # test_binary_listnet.py:16-47
    k += [("binary_listnet", dict(eps=0.0), [0.5, 0.2], [1.0, 0.0], float(-np.log(sm([0.5, 0.2])[0]))),
          ("binary_listnet", {}, [0.5, -1e30], [1.0, 0.0], float(-np.log(sm([0.5, -1e30])[0] + 1e-10))),
          ("binary_listnet", {}, [0.5, 0.2, 0.5], [1.0, 0.0, P], float(-np.log(sm([0.5, 0.2])[0] + 1e-10)))]
    return k


def mrr_kats():
    """test_mrr.py:19-105 as (y_pred rows, y_true rows, ats, expected matrix)"""
    P = -1.0
    return [([[0.5, 0.2]], [[1.0, 0.0]], [10], [[1.0]]), ([[0.5, 0.2]], [[1.0, 0.0]], None, [[1.0]]),
            ([[0.5, 0.2]], [[0.0, 1.0]], [10], [[0.5]]),
            ([[0.2, 0.5], [0.5, 0.2]], [[0.0, 1.0], [0.0, 1.0]], [10], [[1.0], [0.5]]),
            ([[0.5, 0.2]], [[0.0, 1.0]], [1, 2], [[0.0, 0.5]]),
            ([[0.2, 0.5], [0.5, 0.2]], [[0.0, 1.0], [0.0, 1.0]], [1, 2], [[1.0, 1.0], [0.0, 0.5]]),
            ([[0.5, 0.2]], [[0.0, 0.0]], [10], [[0.0]]),
            ([[0.5, 0.2, 1.0]], [[1.0, 0.0, P]], [10], [[1.0]]), ([[0.5, 0.2, 1.0]], [[0.0, 1.0, P]], [10], [[0.5]])]
