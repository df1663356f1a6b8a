"""MI355X-native scoring model with the plugin surface of ``allrank.models.model`` / ``allrank.models.transformer``.

``make_model(fc_model, transformer, post_model, n_features)`` (allrank/models/model.py:131-151, called at
allrank/main.py:75) returns an ``LTRModel`` with ``forward(x, mask, indices)`` / ``score(x, mask, indices)``
(model.py:72-92), the same ``state_dict`` keys and shapes as the reference, and -- given the same
torch RNG state -- bit-identical initial weights: the construction below draws from the generator in the same
order as the reference (one prototype nn.Linear deep-copied 4x per attention block, the encoder layer cloned N
times, Xavier-uniform over parameters() with dim > 1 in registration order).

Arithmetic: every contraction -- the dense projections (``ops.linear`` -> ltrx_gemm_nt / ltrx_gemm_tn) and the per-slate
self-attention (``ops.attention_packed`` -> ltrx_mha_fwd / ltrx_mha_bwd) -- runs on the bf16 matrix cores as three bf16 products
per fp32 product with fp32 accumulation ("split-bf16", fp32-class accuracy; ``ops.arithmetic(linear="hipblaslt", attention=0)``
selects torch's fp32 library GEMMs and the exact-fp32 MFMA attention kernels instead); Q, K and V come from ONE [3d, d] GEMM so that
the attention kernel reads q / k / v as strided views of a single buffer.  The custom LayerNorm (+ fused residual add) is a
hand-written kernel too (allrank_amd/ops.py -> libltrx.so).  Dropout follows torch semantics (nn.Dropout on the residual branches
/ activations); the attention-probability dropout of transformer.py:154-155 happens inside the fused attention kernel
(counter-based mask regenerated in the backward; same distribution, different random stream).
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _instantiate_activation(name):
    """reference: instantiate_class("torch.nn.modules.activation", name) (model.py:28-29,106-107)"""
    if name is None:
        return nn.Identity()
    import torch.nn.modules.activation as A
    return getattr(A, name)()


def first_arg_id(x, *y):       # model.py:8-9
    return x


class FCModel(nn.Module):
    """model.py:12-44: input_norm -> [Linear -> activation -> dropout] per layer (also after the last one)."""

    def __init__(self, sizes, input_norm, activation, dropout, n_features):
        super(FCModel, self).__init__()
        sizes.insert(0, n_features)                      # the reference mutates the config list too (model.py:25)
        layers = [nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])]
        self.input_norm = nn.LayerNorm(n_features) if input_norm else nn.Identity()
        self.activation = _instantiate_activation(activation)
        self.dropout = nn.Dropout(dropout or 0.0)
        self.output_size = sizes[-1]
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        x = self.input_norm(x)
        for layer in self.layers:
            x = self.dropout(self.activation(ops.linear(x, layer.weight, layer.bias)))
        return x


class LayerNorm(nn.Module):
    """transformer.py:59-81 -- unbiased std, eps added to std.  Parameters a_2 / b_2 like the reference."""

    def __init__(self, features, eps=1e-6):
        super(LayerNorm, self).__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.a_2, self.b_2, self.eps)


class SublayerConnection(nn.Module):
    """transformer.py:84-106; only a parameter/dropout holder here -- EncoderLayer fuses add+norm."""

    def __init__(self, size, dropout):
        super(SublayerConnection, self).__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        return x + self.dropout(sublayer(self.norm(x)))


class MultiHeadedAttention(nn.Module):
    """transformer.py:159-203.  linears[0..2] = q/k/v projections, linears[3] = output projection."""

    def __init__(self, h, d_model, dropout=0.1):
        super(MultiHeadedAttention, self).__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        proto = nn.Linear(d_model, d_model)
        self.linears = nn.ModuleList([copy.deepcopy(proto) for _ in range(4)])
        self.attn = None                                  # the reference keeps p_attn here; never materialised now
        self.dropout = nn.Dropout(p=dropout)

    def forward(self, query, key, value, mask=None):
        if query is not key or key is not value:
            raise NotImplementedError("only self-attention (query is key is value) is on the MI355X hot path")
        B, SL, d = query.shape
        if mask is None:
            mask = torch.zeros((B, SL), dtype=torch.bool, device=query.device)
        mask = mask.reshape(B, -1)[:, -SL:] if mask.dim() > 2 else mask
        w = torch.cat([self.linears[i].weight for i in range(3)], dim=0)
        b = torch.cat([self.linears[i].bias for i in range(3)], dim=0)
        qkv = ops.linear(query, w, b)                                      # [B, L, 3d]
        p_drop = self.dropout.p if self.training else 0.0          # transformer.py:154-155, inside the fused kernel
        o = ops.attention_packed(qkv, mask, self.h, p_drop)
        return ops.linear(o, self.linears[3].weight, self.linears[3].bias)


class PositionwiseFeedForward(nn.Module):
    """transformer.py:206-227"""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super(PositionwiseFeedForward, self).__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        # one autograd node: ReLU + dropout in the epilogue of the first GEMM, their backward in the epilogue of w_2's input gradient
        return ops.feed_forward(x, self.w_1.weight, self.w_1.bias, self.w_2.weight, self.w_2.bias,
                                self.dropout.p if self.training else 0.0)


class EncoderLayer(nn.Module):
    """transformer.py:109-134: x = x + drop(attn(norm0(x)));  x = x + drop(ffn(norm1(x)))."""

    def __init__(self, size, self_attn, feed_forward, dropout):
        super(EncoderLayer, self).__init__()
        self.self_attn = self_attn
        self.feed_forward = feed_forward
        self.sublayer = nn.ModuleList([copy.deepcopy(SublayerConnection(size, dropout)) for _ in range(2)])
        self.size = size

    def forward(self, x, mask):
        s0, s1 = self.sublayer[0], self.sublayer[1]
        xn = s0.norm(x)
        a = s0.dropout(self.self_attn(xn, xn, xn, mask))
        # fused: x1 = x + a ; xn1 = LN(x1)
        xn1, x1 = ops.layer_norm_residual(x, a, s1.norm.a_2, s1.norm.b_2, s1.norm.eps)
        return x1 + s1.dropout(self.feed_forward(xn1))


class Encoder(nn.Module):
    """transformer.py:28-56"""

    def __init__(self, layer, N, position):
        super(Encoder, self).__init__()
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(N)])
        self.norm = LayerNorm(layer.size)
        self.position = position

    def forward(self, x, mask, indices):
        if self.position:
            x = self.position(x, mask, indices)
        for layer in self.layers:
            x = layer(x, mask)
        return self.norm(x)


class FixedPositionalEncoding(nn.Module):
    """allrank/models/positional.py:15-50: sin/cos table of max_len rows + one all-zero padding row; the original rank
    of an item (``indices``) selects the row, padded items and ranks beyond max_len get the zero row;
    x <- sqrt(d_model) * x + pe[row].  (Plain torch gather: one pass over [B,L,d])"""

    def __init__(self, d_model, max_len=5000):
        super(FixedPositionalEncoding, self).__init__()
        import math
        pos = torch.arange(0.0, max_len).unsqueeze(1)
        freq = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(10000.0) / d_model))
        table = torch.zeros(max_len + 1, d_model)
        table[:max_len, 0::2] = torch.sin(pos * freq)
        table[:max_len, 1::2] = torch.cos(pos * freq)
        self.padding_idx = max_len
        self.register_buffer("pe", table)

    def forward(self, x, mask, indices):
        rows = indices.masked_fill(mask, self.padding_idx).clamp(max=self.padding_idx)
        return (self.pe.shape[1] ** 0.5) * x + self.pe[rows]


class LearnedPositionalEncoding(nn.Module):
    """allrank/models/positional.py:53-77: an Embedding of max_len + 1 rows whose last row is the padding row."""

    def __init__(self, d_model, max_len=5000):
        super(LearnedPositionalEncoding, self).__init__()
        self.pe = nn.Embedding(max_len + 1, d_model, padding_idx=-1)

    def forward(self, x, mask, indices):
        pad = self.pe.padding_idx
        rows = indices.masked_fill(mask, pad).clamp(max=pad)
        return (self.pe.embedding_dim ** 0.5) * x + self.pe(rows)


def _make_positional_encoding(d_model, positional_encoding):
    """positional.py:80-94; accepts the reference's attrs PositionalEncoding object or a dict."""
    if positional_encoding is None:
        return None
    get = positional_encoding.get if isinstance(positional_encoding, dict) else lambda k: getattr(positional_encoding, k)
    strategy, max_indices = get("strategy"), get("max_indices")
    if strategy == "fixed":
        return FixedPositionalEncoding(d_model, max_len=max_indices)
    if strategy == "learned":
        return LearnedPositionalEncoding(d_model, max_len=max_indices)
    raise ValueError("Invalid positional encoding type: {}".format(strategy))        # positional.py:94


def make_transformer(N=6, d_ff=2048, h=8, dropout=0.1, n_features=136, positional_encoding=None):
    """transformer.py:230-247 (same construction order: attention, feed-forward, position, encoder -> same RNG draws)"""
    attn = MultiHeadedAttention(h, n_features, dropout)
    ff = PositionwiseFeedForward(n_features, d_ff, dropout)
    position = _make_positional_encoding(n_features, positional_encoding)
    return Encoder(EncoderLayer(n_features, copy.deepcopy(attn), copy.deepcopy(ff), dropout), N, position)


class OutputLayer(nn.Module):
    """model.py:95-128"""

    def __init__(self, d_model, d_output, output_activation=None):
        super(OutputLayer, self).__init__()
        self.activation = _instantiate_activation(output_activation)
        self.d_output = d_output
        self.w_1 = nn.Linear(d_model, d_output)

    def forward(self, x):
        if self.d_output == 1 and x.dim() == 3:
            return self.activation(ops.score_head(x, self.w_1.weight, self.w_1.bias))
        return self.activation(self.w_1(x).squeeze(dim=2))

    def score(self, x):
        if self.d_output > 1:
            return self.forward(x).sum(-1)
        return self.forward(x)


class LTRModel(nn.Module):
    """model.py:47-92"""

    def __init__(self, input_layer, encoder, output_layer):
        super(LTRModel, self).__init__()
        self.input_layer = input_layer if input_layer else nn.Identity()
        self.encoder = encoder if encoder else first_arg_id
        self.output_layer = output_layer

    def prepare_for_output(self, x, mask, indices):
        return self.encoder(self.input_layer(x), mask, indices)

    def forward(self, x, mask, indices):
        return self.output_layer(self.prepare_for_output(x, mask, indices))

    def score(self, x, mask, indices):
        return self.output_layer.score(self.prepare_for_output(x, mask, indices))


def _as_dict(cfg):
    """accepts the reference's attrs TransformerConfig (config.py:8-15), a plain dict, or any object with fields."""
    if cfg is None or isinstance(cfg, dict):
        return cfg
    try:
        from attr import asdict
        return asdict(cfg, recurse=False)
    except Exception:
        return {k: getattr(cfg, k) for k in ("N", "d_ff", "h", "positional_encoding", "dropout")}


def make_model(fc_model, transformer, post_model, n_features):
    """model.py:131-151"""
    if fc_model:
        fc_model = FCModel(**fc_model, n_features=n_features)
    d_model = n_features if not fc_model else fc_model.output_size
    if transformer:
        transformer = make_transformer(n_features=d_model, **_as_dict(transformer))
    model = LTRModel(fc_model, transformer, OutputLayer(d_model, **post_model))
    for p in model.parameters():            # Glorot / fan_avg (model.py:148-150)
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
