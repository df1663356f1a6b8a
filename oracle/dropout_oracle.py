"""The engine's counter-based dropout masks, restated in numpy  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference draws its dropout masks from torch's generator (nn.Dropout: model.py:43, transformer.py:105,155,227), so a
training-mode comparison with it can only be statistical.  The engine's masks are pure functions of (site seed,
device step word, element index) -- allrank_amd/csrc/ltrx_device.h ``drop_keep_scale``, ltrx_mha_res.hip ``drop_row_seed`` /
``drop_scale_rk`` -- so the ORACLE can be handed exactly the masks a step used and the step compared with it to round-off
(tests/test_gpu_parity.py::test_fused_step_with_dropout_matches_the_fp64_oracle_under_the_same_masks).  The functions here are
pinned to the kernels bit for bit by the same test file (ltrx_dropout_apply of a tensor of ones; an attention call whose
probabilities are uniform and whose values are one-hot rows).
"""
import numpy as np

U32 = np.uint32
M32 = np.uint64(0xFFFFFFFF)


def _u32(x):
    return (np.asarray(x, dtype=np.uint64) & M32).astype(np.uint64)


def _mul(a, b):
    return (_u32(a) * np.uint64(b)) & M32


def spec(p, seed, step_word):
    """(seed of this step, threshold, 1/(1-p)) as ltrx_make_drop + the kernels' ``seed ^= step * 0x9E3779B9`` compute them"""
    p32 = np.float32(p)
    thresh = int(np.float32(p32 * np.float32(16777216.0))) if p > 0 else 0
    inv_keep = np.float32(1.0) / (np.float32(1.0) - p32) if p > 0 else np.float32(1.0)
    s = (int(seed) ^ ((int(step_word) * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return s, thresh, inv_keep


def keep_scale(p, seed, step_word, shape):
    """multipliers (0 or 1/(1-p), float32) of the elements 0 .. prod(shape)-1 of a row-major tensor: the GEMM-epilogue, LayerNorm
    residual and ltrx_dropout_apply masks (ltrx_device.h: murmur3 finaliser over the folded 64-bit element index)"""
    if p <= 0:
        return np.ones(shape, np.float32)
    s, thresh, inv_keep = spec(p, seed, step_word)
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64)
    x = (idx & M32) ^ _mul(idx >> np.uint64(32), 0x9E3779B9) ^ np.uint64(s)
    x ^= x >> np.uint64(16)
    x = _mul(x, 0x85EBCA6B)
    x ^= x >> np.uint64(13)
    x = _mul(x, 0xC2B2AE35)
    x ^= x >> np.uint64(16)
    keep = (x >> np.uint64(8)) >= np.uint64(thresh)
    return np.where(keep, inv_keep, np.float32(0)).astype(np.float32).reshape(shape)


def attention_keep_scale(p, seed, step_word, B, H, L):
    """[B, H, L(query), L(key)] multipliers of the attention probabilities (ltrx_mha_res.hip / ltrx_mha.hip: a row seed per
    (slate, head, query), one multiply-xorshift round per key)"""
    if p <= 0:
        return np.ones((B, H, L, L), np.float32)
    s, thresh, inv_keep = spec(p, seed, step_word)
    bh = np.arange(B * H, dtype=np.uint64)[:, None]
    q = np.arange(L, dtype=np.uint64)[None, :]
    x = np.uint64(s) ^ _mul((bh * np.uint64(L) + q) & M32, 0x9E3779B9)
    x ^= x >> np.uint64(16)
    x = _mul(x, 0x85EBCA6B)
    x ^= x >> np.uint64(13)
    x = _mul(x, 0xC2B2AE35)
    x ^= x >> np.uint64(16)
    row_seed = x[:, :, None]                                          # [B H, L, 1]
    key = np.arange(L, dtype=np.uint64)[None, None, :]
    y = _mul(row_seed ^ key, 0x9E3779B1)
    y ^= y >> np.uint64(16)
    y = _mul(y, 0x85EBCA6B)
    keep = (y >> np.uint64(8)) >= np.uint64(thresh)
    return np.where(keep, inv_keep, np.float32(0)).astype(np.float32).reshape(B, H, L, L)


def engine_masks(trainer, step_word):
    """the masks of every dropout site of a FusedTrainer step (non-compact) as the dict oracle/model_oracle.forward(drop=) takes:
    {"fc": [per FC layer [B, L, size]], "layers": [{"att": [B, H, L, L], "ff": [B, L, d_ff], "s0": [B, L, d], "s1": [B, L, d]}]}"""
    t = trainer
    B, L, d = t.B, t.L, t.d
    out = {"fc": [keep_scale(t.p_fc, t._site(1000 + i), step_word, (B, L, s)) for i, s in enumerate(t.fc_sizes[1:])], "layers": []}
    for st in t.layers:
        out["layers"].append({
            "att": attention_keep_scale(st["p_att"], st["s_att"], step_word, B, t.h, L),
            "ff": keep_scale(st["p_ff"], st["s_ff"], step_word, (B, L, t.dff)),
            "s0": keep_scale(st["p_s0"], st["s_s0"], step_word, (B, L, d)),
            "s1": keep_scale(st["p_s1"], st["s_s1"], step_word, (B, L, d))})
    return out
