"""MI355X-native ranking metrics with the signatures of ``allrank.models.metrics`` (metrics.py:7-77; looked up by
name at allrank/training/train_utils.py:50).  ``ndcg``/``dcg``/``mrr`` run as one HIP kernel per call (libltrx.so)."""
import ctypes

import torch

from . import _lib as L

PADDED_Y_VALUE = -1


def _run(y_pred, y_true, ats, padding_indicator, filler_value, want_order, gain_function=None):
    if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
        raise ValueError("y_pred and y_true must both be [batch_size, slate_length]")
    L.require_device(y_pred, y_true)
    yp = L.f32c(y_pred.detach())
    yt = L.f32c(y_true.detach())
    B, SL = yp.shape
    if ats is None:
        ats = [SL]                                    # metrics.py:58-59
    ats = [min(int(a), SL) for a in ats]              # metrics.py:60
    n = len(ats)
    nd = torch.empty((B, n), dtype=torch.float32, device=yp.device)
    dc = torch.empty((B, n), dtype=torch.float32, device=yp.device)
    order = torch.empty((B, SL), dtype=torch.int64, device=yp.device) if want_order else None
    arr = (ctypes.c_int * n)(*ats)
    if gain_function is None:                         # the default gain 2^x - 1, evaluated in the kernel
        L.check(L.lib().ltrx_ndcg_at(L.ptr(yp), L.ptr(yt), B, SL, arr, n, float(padding_indicator), float(filler_value),
                                     L.ptr(nd), L.ptr(dc), L.ptr(order), None, L.stream_of(yp)), "ndcg_at")
    else:
        # metrics.py:67: gains = gain_function(labels gathered in predicted order) -- an elementwise callable commutes with the
        # gather, so it is evaluated once per item on the masked labels (padded -> 0, metrics.py:35) with torch on the device and
        # the kernel ranks, discounts and accumulates the pre-computed gains (ltrx_ndcg_at_gains)
        gains = gain_function(torch.where(yt == padding_indicator, torch.zeros_like(yt), yt))
        if not torch.is_tensor(gains) or gains.shape != yt.shape:
            raise ValueError("gain_function must map a [batch_size, slate_length] tensor of labels to a tensor of the same shape")
        gains = L.f32c(gains.detach())
        L.check(L.lib().ltrx_ndcg_at_gains(L.ptr(yp), L.ptr(yt), L.ptr(gains), B, SL, arr, n, float(padding_indicator),
                                           float(filler_value), L.ptr(nd), L.ptr(dc), L.ptr(order), None, L.stream_of(yp)),
                "ndcg_at_gains")
    return nd, dc, order


def ndcg(y_pred, y_true, ats=None, gain_function=None, padding_indicator=PADDED_Y_VALUE, filler_value=1.0,
         return_order=False):
    """NDCG@ats (metrics.py:7-28): [batch, len(ats)]; slates without a relevant item get ``filler_value`` (1.0).
    ``gain_function=None`` is the reference's default gain 2^x - 1 (in the kernel); any other elementwise callable -- the
    reference itself passes the identity, losses/neuralNDCG.py:58 -- is applied to the labels on the device first."""
    nd, _, order = _run(y_pred, y_true, ats, padding_indicator, filler_value, return_order, gain_function)
    return (nd, order) if return_order else nd


def dcg(y_pred, y_true, ats=None, gain_function=None, padding_indicator=PADDED_Y_VALUE):
    """DCG@ats (metrics.py:41-77); ``gain_function`` as in ``ndcg``."""
    return _run(y_pred, y_true, ats, padding_indicator, 1.0, False, gain_function)[1]


def mrr(y_pred, y_true, ats=None, padding_indicator=PADDED_Y_VALUE):
    """MRR@ats (metrics.py:80-113): [batch, len(ats)] -- 1 / (1 + rank of the first item carrying the slate's maximum
    label) when that rank is below the cut-off, else 0; the reference's batch-level zeroing (all maxima 0) is kept."""
    if y_pred.dim() != 2 or y_pred.shape != y_true.shape:
        raise ValueError("y_pred and y_true must both be [batch_size, slate_length]")
    L.require_device(y_pred, y_true)
    yp = L.f32c(y_pred.detach())
    yt = L.f32c(y_true.detach())
    B, SL = yp.shape
    if ats is None:
        ats = [SL]
    ats = [int(a) for a in ats]
    n = len(ats)
    out = torch.empty((B, n), dtype=torch.float32, device=yp.device)
    lib = L.lib()
    ws = L.workspace(lib.ltrx_mrr_workspace_bytes(B, SL, n), yp)
    arr = (ctypes.c_int * n)(*ats)
    L.check(lib.ltrx_mrr_at(L.ptr(yp), L.ptr(yt), B, SL, arr, n, float(padding_indicator), L.ptr(out), L.ptr(ws),
                            L.stream_of(yp)), "mrr_at")
    return out
