"""ctypes binding of libltrx.so (include/ltrx.h).  No fallback: if the library is missing, importing the
kernels fails loudly (the product path never routes through a CPU implementation)."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTRX_LIB_PATH") or os.path.join(_HERE, "libltrx.so")     # (override: A/B runs of two builds)

_c_float_p = ctypes.c_void_p   # device pointers are passed as raw addresses
_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

SIGNATURES = {
    "ltrx_version": (_i, []),
    "ltrx_listnet_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_listnet_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_listmle_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_listmle_fwd_bwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_approxndcg_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_approxndcg_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_lambdaloss_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_lambdaloss_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_neuralndcg_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltrx_neuralndcg_prepare": (_i, [_vp, _i, _i, _f, _i, _i, _vp, _vp, _vp, _vp]),
    "ltrx_neuralndcg_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _i, _i, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ltrx_ranknet_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_ranknet_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_bce_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltrx_bce_fwd_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_pointwise_rmse_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_pointwise_rmse_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ltrx_binary_listnet_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_binary_listnet_fwd_bwd": (_i, [_vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ltrx_mrr_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltrx_mrr_at": (_i, [_vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_int), _i, _f, _vp, _vp, _vp]),
    "ltrx_ndcg_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_ndcg_at": (_i, [_vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_int), _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_ndcg_at_gains": (_i, [_vp, _vp, _vp, _i, _i, ctypes.POINTER(ctypes.c_int), _i, _f, _f, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, ctypes.c_uint32, _vp, _vp]),
    "ltrx_layernorm_fwd_image": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _f, ctypes.c_uint32, _vp, _vp]),
    "ltrx_layernorm_bwd_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_layernorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_layernorm_bwd_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "ltrx_mha_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _f, ctypes.c_uint32, _vp, _vp, _vp, _i, _vp]),
    "ltrx_mha_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ltrx_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _i, _vp, _f, _vp, _vp]),
    "ltrx_sgd_step": (_i, [_vp, _vp, _vp, _sz, _f, _f, _i, _f, _f, _vp, _vp]),
    "ltrx_clip_workspace_bytes": (_sz, [_sz]),
    "ltrx_clip_grad_norm_scale": (_i, [_vp, _sz, _f, _vp, _vp, _vp, _vp]),
    "ltrx_colsum_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_colsum": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "ltrx_relu_bwd": (_i, [_vp, _vp, _sz, _f, _vp]),
    "ltrx_dropout_apply": (_i, [_vp, _vp, _sz, _f, ctypes.c_uint32, _vp, _vp]),
    "ltrx_bump_u32": (_i, [_vp, _vp]),
    "ltrx_first_nonfinite": (_i, [_vp, _sz, _vp, _i, _vp, _vp]),
    "ltrx_gather_rows": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp]),
    "ltrx_packed_row_index": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "ltrx_scatter_rows": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp]),
    "ltrx_transpose_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "ltrx_bias_act": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ltrx_score_head_fwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ltrx_score_head_bwd_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_score_head_bwd": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_gemm_nt_relu_bits_bytes": (_sz, [_i, _i, _i]),
    "ltrx_gemm_nt": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _f, ctypes.c_uint32, _vp, _i, _i, _vp]),
    "ltrx_split_image": (_i, [_vp, _vp, _sz, _vp]),
    "ltrx_gemm_nt_image_ok": (_i, [_i, _i, _i]),
    "ltrx_gemm_nt_img": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _f, ctypes.c_uint32, _vp, _i, _i, _i, _vp]),
    "ltrx_weight_images": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ltrx_ingest_batch": (_i, [_vp, _vp, _sz, _sz, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "ltrx_gemm_tn_workspace_bytes": (_sz, [_i, _i, _i]),
    "ltrx_gemm_tn_splits": (_i, [_i, _i, _i]),
    "ltrx_gemm_tn": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ltrx_gemm_tn_group_workspace_bytes": (_sz, [_i, _i, _vp, _vp]),
    "ltrx_gemm_tn_group": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp]),
    "ltrx_gemm_tn_group_img": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_reduce_group": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ltrx_debug_tn_group_map": (_i, [_i, _i, _vp, _vp, _vp, _vp]),
    "ltrx_layernorm_torch_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "ltrx_posenc_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp]),
    "ltrx_posenc_table_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "ltrx_scale_inplace": (_i, [_vp, _sz, _f, _vp]),
    "ltrx_out_act_fwd": (_i, [_vp, _sz, _i, _vp, _vp]),
    "ltrx_out_act_bwd": (_i, [_vp, _vp, _sz, _i, _vp, _vp]),
    "ltrx_fixlength_positions": (_i, [_vp, _vp, _vp, _i, _i, _i, ctypes.c_uint64, _vp, _vp]),
    "ltrx_assemble_batch": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ltrx_libsvm_parse": (_i, [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ltrx_selftest_mfma32x32x2": (_i, [_vp, _vp, _vp, _vp]),
    "ltrx_fc_listnet_supported": (_i, [_i, _i, _i]),
    "ltrx_fc_listnet_workspace_bytes": (_sz, [_i, _i, _i, _i, _sz]),
    "ltrx_fc_linear_listnet_workspace_bytes": (_sz, [_i, _i]),
    "ltrx_fc_linear_listnet_step": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _sz, _sz, _sz, _sz, _f, _f, _f, _vp, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "ltrx_fc_listnet_step": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _sz, _sz, _sz, _sz, _f, _f, _f, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _vp, _vp]),
    "ltrx_mha_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, ctypes.c_uint32, _vp, _vp, _vp, _i, _vp, _vp]),
}

_lib = None
MAX_SLATE_LEN = 2048            # LTRX_MAX_SLATE_LEN (include/ltrx.h; checked against the header by tests/test_abi.py)
MAX_LONG_SLATE_LEN = 16384      # LTRX_MAX_LONG_SLATE_LEN: listNet / listMLE / approxNDCG / lambdaLoss (work arrays in the workspace beyond LDS)
MAX_METRIC_SLATE_LEN = 8192     # LTRX_MAX_METRIC_SLATE_LEN


class _Stream(ctypes.c_void_p):
    """a hipStream_t that remembers the device it belongs to (``launch_stream`` / ``stream_of``)"""
    dev = None


class _Bound(object):
    """namespace of the bound entry points (attribute access like a ctypes.CDLL)"""


def _on_stream_device(fn):
    """Entry points that take a stream launch on it; HIP requires the stream's device to be the CURRENT device.  The stream
    argument produced by ``stream_of`` / ``launch_stream`` carries its device; when that is not the current device (explicit
    cuda:1 tensors while cuda:0 is current -- plain user code after the reference's DataParallel gather, or replica threads)
    the launch is wrapped in a device guard.  Every launch made with that stream object is guarded, not only the first
    (ADVICE r2).  A raw / NULL stream is left alone."""
    def call(*args):
        dev = getattr(args[-1], "dev", None) if args else None
        if dev is None or dev == torch.cuda.current_device():
            return fn(*args)
        with torch.cuda.device(dev):
            return fn(*args)
    call.__name__ = fn.__name__
    return call


def lib():
    """The loaded library (loads on first use).  Raises if libltrx.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "allrank_amd: %s is missing -- build it with `python -m allrank_amd.build` "
                "(there is no CPU fallback for the HIP kernels)" % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        b = _Bound()
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)     # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            launches = res is _i and args and args[-1] is _vp and not name.endswith("_bytes")
            setattr(b, name, _on_stream_device(fn) if launches else fn)
        _lib = b
    return _lib


def check(rc, what):
    if rc != 0:
        kinds = {-1: "invalid argument",
                 -2: "unsupported shape (slate length above LTRX_MAX_SLATE_LEN = %d for a loss -- LTRX_MAX_LONG_SLATE_LEN = %d for listNet / "
                     "listMLE / approxNDCGLoss / lambdaLoss -- or LTRX_MAX_METRIC_SLATE_LEN = %d for a metric, or an alignment the kernel "
                     "needs -- see include/ltrx.h; there is no fallback path)" % (MAX_SLATE_LEN, MAX_LONG_SLATE_LEN, MAX_METRIC_SLATE_LEN)}
        msg = kinds.get(rc, "HIP error %d" % (-rc - 1000) if rc <= -1000 else "error")
        raise RuntimeError("libltrx %s failed: %s (code %d)" % (what, msg, rc))


def ptr(t):
    """raw device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def launch_stream(device):
    """torch's current stream on ``device`` as a hipStream_t that carries its device for the launch guard of ``lib()``."""
    device = torch.device(device)
    st = _Stream(torch.cuda.current_stream(device).cuda_stream)
    st.dev = device.index if device.index is not None else torch.cuda.current_device()
    return st


def stream_of(t):
    return launch_stream(t.device)


def require_device(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("allrank_amd kernels run on the MI355X only: got a %s tensor (no CPU fallback; "
                               "use the reference implementation on CPU)" % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("allrank_amd kernels need all tensors of a call on one device: got %s and %s" % (dev, t.device))


def workspace(nbytes, like):
    return torch.empty(max(int(nbytes), 64), dtype=torch.uint8, device=like.device)


def f32c(t):
    """contiguous float32 view/copy"""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
