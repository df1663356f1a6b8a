"""CPU oracle for the scoring model of the hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement (forward, hand-derived backward, Adam) of allRank's LTRModel:
FCModel -> N x pre-norm Transformer encoder layers over the slate -> custom LayerNorm -> OutputLayer
(allrank/models/model.py:12-151, allrank/models/transformer.py:28-247).  Dropout is modelled as INJECTED masks
(``forward(..., drop=)``: one multiplier array -- 0 or 1/(1-p) -- per nn.Dropout site of the reference: model.py:43,
transformer.py:105,155,227; None = every dropout off, the parity default): the reference draws its masks from
torch's generator, the engine's are counter-based and reproducible (oracle/dropout_oracle.py).  Parameters are a dict keyed exactly like the
reference ``state_dict`` so weights can be exchanged with the real thing.

Pinned by tests/test_oracle_pinned.py against golden vectors generated from the reference itself
(tests/golden/make_golden.py): forward scores and autograd gradients of every parameter.
Used as ``cpu_baseline`` (kind "port") by bench.py: matmuls go to numpy's BLAS threads.
"""
import math
import numpy as np

ACTS = {
    None: (lambda x: x, lambda x, y, g: g),
    "ReLU": (lambda x: np.maximum(x, 0), lambda x, y, g: g * (x > 0)),
    "Tanh": (lambda x: np.tanh(x), lambda x, y, g: g * (1 - y * y)),
    "Sigmoid": (lambda x: 1 / (1 + np.exp(-x)), lambda x, y, g: g * y * (1 - y)),
}


def xavier_uniform(rng, out_f, in_f, dtype=np.float32):
    a = math.sqrt(6.0 / (in_f + out_f))                         # nn.init.xavier_uniform_ (model.py:148-150)
    return rng.uniform(-a, a, size=(out_f, in_f)).astype(dtype)


def init_params(cfg, seed=0, dtype=np.float32):
    """cfg: dict(n_features, fc_sizes, fc_activation, fc_input_norm, N, d_ff, h, d_output, output_activation).
    Same shapes/keys as the reference state_dict; biases get torch's default U(-1/sqrt(in), 1/sqrt(in))."""
    rng = np.random.default_rng(seed)
    p = {}

    def lin(name, out_f, in_f):
        p[name + ".weight"] = xavier_uniform(rng, out_f, in_f, dtype)
        b = 1.0 / math.sqrt(in_f)
        p[name + ".bias"] = rng.uniform(-b, b, size=(out_f,)).astype(dtype)

    sizes = [cfg["n_features"]] + list(cfg.get("fc_sizes") or [])
    if cfg.get("fc_input_norm"):
        p["input_layer.input_norm.weight"] = np.ones(sizes[0], dtype)
        p["input_layer.input_norm.bias"] = np.zeros(sizes[0], dtype)
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        lin("input_layer.layers.%d" % i, b, a)
    d = sizes[-1]
    for n in range(cfg.get("N", 0)):
        pre = "encoder.layers.%d." % n
        for j in range(4):
            lin(pre + "self_attn.linears.%d" % j, d, d)
        lin(pre + "feed_forward.w_1", cfg["d_ff"], d)
        lin(pre + "feed_forward.w_2", d, cfg["d_ff"])
        for j in range(2):
            p[pre + "sublayer.%d.norm.a_2" % j] = np.ones(d, dtype)
            p[pre + "sublayer.%d.norm.b_2" % j] = np.zeros(d, dtype)
    if cfg.get("N", 0):
        p["encoder.norm.a_2"] = np.ones(d, dtype)
        p["encoder.norm.b_2"] = np.zeros(d, dtype)
    lin("output_layer.w_1", cfg.get("d_output", 1), d)
    return p


# ---- primitives (forward returns (out, cache); backward returns input grad and accumulates param grads) ----
def linear_fwd(x, W, b):
    return x @ W.T + b


def linear_bwd(x, W, gy, grads, name):
    x2 = x.reshape(-1, x.shape[-1])
    g2 = gy.reshape(-1, gy.shape[-1])
    grads[name + ".weight"] = grads.get(name + ".weight", 0) + g2.T @ x2
    grads[name + ".bias"] = grads.get(name + ".bias", 0) + g2.sum(0)
    return gy @ W


def custom_ln_fwd(x, a, b, eps=1e-6):
    """transformer.py:73-81: a*(x-mean)/(std+eps)+b with the UNBIASED std and eps added to std."""
    n = x.shape[-1]
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    std = np.sqrt((xc * xc).sum(-1, keepdims=True) / (n - 1))
    r = 1.0 / (std + x.dtype.type(eps))
    xhat = xc * r
    return a * xhat + b, (xc, std, r, xhat)


def custom_ln_bwd(cache, a, gy, grads, name):
    xc, std, r, xhat = cache
    n = xc.shape[-1]
    grads[name + ".a_2"] = grads.get(name + ".a_2", 0) + (gy * xhat).reshape(-1, n).sum(0)
    grads[name + ".b_2"] = grads.get(name + ".b_2", 0) + gy.reshape(-1, n).sum(0)
    g = gy * a
    gm = g.mean(-1, keepdims=True)
    dot = (g * xc).sum(-1, keepdims=True)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = np.where(std > 0, r * r * dot / ((n - 1) * std), 0)
    return r * (g - gm) - t * xc


def torch_ln_fwd(x, w, b, eps=1e-5):
    """nn.LayerNorm (FCModel.input_norm, model.py:27): biased variance, eps inside the sqrt."""
    mean = x.mean(-1, keepdims=True)
    xc = x - mean
    var = (xc * xc).mean(-1, keepdims=True)
    r = 1.0 / np.sqrt(var + x.dtype.type(eps))
    xhat = xc * r
    return xhat * w + b, (xhat, r)


def torch_ln_bwd(cache, w, gy, grads, name):
    xhat, r = cache
    n = xhat.shape[-1]
    grads[name + ".weight"] = grads.get(name + ".weight", 0) + (gy * xhat).reshape(-1, n).sum(0)
    grads[name + ".bias"] = grads.get(name + ".bias", 0) + gy.reshape(-1, n).sum(0)
    g = gy * w
    return r * (g - g.mean(-1, keepdims=True) - xhat * (g * xhat).mean(-1, keepdims=True))


def attention_fwd(q, k, v, mask, pdrop=None):
    """transformer.py:137-156 on q,k,v [B,h,L,dk]; mask [B,L] True = padded KEY (masked_fill -inf); pdrop: the dropout
    multipliers of the probabilities (transformer.py:154-155), applied after the softmax.  Returns (p_dropped @ v, p)."""
    dk = q.shape[-1]
    sc = (q @ np.swapaxes(k, -1, -2)) / q.dtype.type(math.sqrt(dk))
    sc = np.where(mask[:, None, None, :], -np.inf, sc)
    m = sc.max(-1, keepdims=True)
    e = np.exp(sc - m)
    p = e / e.sum(-1, keepdims=True)
    pd = p if pdrop is None else p * pdrop
    return pd @ v, p


def attention_bwd(q, k, v, p, go, pdrop=None):
    dk = q.shape[-1]
    pd = p if pdrop is None else p * pdrop
    gv = np.swapaxes(pd, -1, -2) @ go
    gp = go @ np.swapaxes(v, -1, -2)
    if pdrop is not None:
        gp = gp * pdrop
    gs = p * (gp - (gp * p).sum(-1, keepdims=True))
    gs = gs / q.dtype.type(math.sqrt(dk))
    return gs @ k, np.swapaxes(gs, -1, -2) @ q, gv


def forward(p, cfg, x, mask, drop=None):
    """LTRModel.forward (model.py:72-80) for d_output == 1 (scores [B,L]); returns (scores, cache).
    drop: None (model.eval() / dropout 0) or {"fc": [mask per FC layer], "layers": [{"att", "ff", "s0", "s1"} per encoder layer]}
    -- multiplier arrays (0 or 1/(1-p)) of the nn.Dropout sites: after every FC activation (model.py:43), on the attention
    probabilities (transformer.py:155), after the feed-forward ReLU (:227), on each residual branch (:105)."""
    B, L, _ = x.shape
    cache = {"x": x, "mask": mask, "drop": drop}
    dt = x.dtype

    def dm(m_):
        return None if m_ is None else np.asarray(m_, dtype=dt)
    act_f, _ = ACTS[cfg.get("fc_activation")]
    h = x
    if cfg.get("fc_input_norm"):
        h, cache["in_ln"] = torch_ln_fwd(h, p["input_layer.input_norm.weight"], p["input_layer.input_norm.bias"])
    fc = []
    nfc = len(cfg.get("fc_sizes") or [])
    for i in range(nfc):                                           # model.py:42-43
        z = linear_fwd(h, p["input_layer.layers.%d.weight" % i], p["input_layer.layers.%d.bias" % i])
        y = act_f(z)
        fc.append((h, z, y))
        h = y if drop is None else y * dm(drop["fc"][i])
    cache["fc"] = fc
    d = h.shape[-1]
    H = cfg.get("h", 1)
    layers = []
    for n in range(cfg.get("N", 0)):
        pre = "encoder.layers.%d." % n
        lc = {"x0": h}
        xn, lc["ln0"] = custom_ln_fwd(h, p[pre + "sublayer.0.norm.a_2"], p[pre + "sublayer.0.norm.b_2"])
        lc["xn0"] = xn
        dk = d // H
        qkv = []
        for j in range(3):                                         # transformer.py:193-195
            t = linear_fwd(xn, p[pre + "self_attn.linears.%d.weight" % j], p[pre + "self_attn.linears.%d.bias" % j])
            qkv.append(t.reshape(B, L, H, dk).transpose(0, 2, 1, 3))
        ld = None if drop is None else drop["layers"][n]
        o, pa = attention_fwd(qkv[0], qkv[1], qkv[2], mask, None if ld is None else dm(ld["att"]))
        lc["qkv"], lc["p"] = qkv, pa
        oc = o.transpose(0, 2, 1, 3).reshape(B, L, d)              # :201-202
        lc["oc"] = oc
        br = linear_fwd(oc, p[pre + "self_attn.linears.3.weight"], p[pre + "self_attn.linears.3.bias"])      # :203
        h = h + (br if ld is None else br * dm(ld["s0"]))                                                    # :105
        lc["x1"] = h
        xn, lc["ln1"] = custom_ln_fwd(h, p[pre + "sublayer.1.norm.a_2"], p[pre + "sublayer.1.norm.b_2"])
        lc["xn1"] = xn
        z = linear_fwd(xn, p[pre + "feed_forward.w_1.weight"], p[pre + "feed_forward.w_1.bias"])
        r = np.maximum(z, 0)                                       # :227
        if ld is not None:
            r = r * dm(ld["ff"])
        lc["z"], lc["r"] = z, r
        br = linear_fwd(r, p[pre + "feed_forward.w_2.weight"], p[pre + "feed_forward.w_2.bias"])
        h = h + (br if ld is None else br * dm(ld["s1"]))
        layers.append(lc)
    cache["layers"] = layers
    if cfg.get("N", 0):
        cache["enc_in"] = h
        h, cache["enc_ln"] = custom_ln_fwd(h, p["encoder.norm.a_2"], p["encoder.norm.b_2"])    # :56
    cache["out_in"] = h
    z = linear_fwd(h, p["output_layer.w_1.weight"], p["output_layer.w_1.bias"])[..., 0]           # model.py:117
    oact_f, _ = ACTS[cfg.get("output_activation")]
    y = oact_f(z)
    cache["out_z"], cache["out_y"] = z, y
    return y, cache


def backward(p, cfg, cache, gscores, relu_masks=None, fc_relu_masks=None):
    """returns dict of parameter gradients (same keys as p).  relu_masks (optional, one bool [B, L, d_ff] array per encoder
    layer): the feed-forward ReLU derivative pattern to differentiate through instead of this forward's own (z > 0).  ReLU is
    not differentiable at 0, so two fp32-class evaluations of the same weights legitimately pick different branches for the
    few units whose pre-activation is within round-off of 0; a parity test that hands in the ENGINE's pattern compares the two
    gradients on the same branch (and counts the differing units separately).  fc_relu_masks: the same for the FCModel
    activations (one bool array per FC layer, ReLU stacks only)."""
    grads = {}
    B, L = gscores.shape
    drop = cache.get("drop")
    dt = gscores.dtype
    _, oact_b = ACTS[cfg.get("output_activation")]
    gz = oact_b(cache["out_z"], cache["out_y"], gscores)[..., None]
    g = linear_bwd(cache["out_in"], p["output_layer.w_1.weight"], gz, grads, "output_layer.w_1")
    if cfg.get("N", 0):
        g = custom_ln_bwd(cache["enc_ln"], p["encoder.norm.a_2"], g, grads, "encoder.norm")
    H = cfg.get("h", 1)
    for n in reversed(range(cfg.get("N", 0))):
        pre = "encoder.layers.%d." % n
        lc = cache["layers"][n]
        d = lc["x0"].shape[-1]
        dk = d // H
        ld = None if drop is None else {k_: np.asarray(v_, dtype=dt) for k_, v_ in drop["layers"][n].items()}
        gb = g if ld is None else g * ld["s1"]
        gr = linear_bwd(lc["r"], p[pre + "feed_forward.w_2.weight"], gb, grads, pre + "feed_forward.w_2")
        if ld is not None:
            gr = gr * ld["ff"]
        gzz = gr * ((lc["z"] > 0) if relu_masks is None else relu_masks[n])
        gxn = linear_bwd(lc["xn1"], p[pre + "feed_forward.w_1.weight"], gzz, grads, pre + "feed_forward.w_1")
        g = g + custom_ln_bwd(lc["ln1"], p[pre + "sublayer.1.norm.a_2"], gxn, grads, pre + "sublayer.1.norm")
        gb = g if ld is None else g * ld["s0"]
        goc = linear_bwd(lc["oc"], p[pre + "self_attn.linears.3.weight"], gb, grads, pre + "self_attn.linears.3")
        go = goc.reshape(B, L, H, dk).transpose(0, 2, 1, 3)
        q, k, v = lc["qkv"]
        gq, gk, gv = attention_bwd(q, k, v, lc["p"], go, None if ld is None else ld["att"])
        gxn = 0
        for j, gt in enumerate((gq, gk, gv)):
            gt2 = gt.transpose(0, 2, 1, 3).reshape(B, L, d)
            gxn = gxn + linear_bwd(lc["xn0"], p[pre + "self_attn.linears.%d.weight" % j], gt2, grads,
                                   pre + "self_attn.linears.%d" % j)
        g = g + custom_ln_bwd(lc["ln0"], p[pre + "sublayer.0.norm.a_2"], gxn, grads, pre + "sublayer.0.norm")
    _, act_b = ACTS[cfg.get("fc_activation")]
    nfc = len(cfg.get("fc_sizes") or [])
    for i in reversed(range(nfc)):
        hin, z, y = cache["fc"][i]
        if drop is not None:
            g = g * np.asarray(drop["fc"][i], dtype=dt)
        g = act_b(z, y, g) if fc_relu_masks is None else g * fc_relu_masks[i]
        g = linear_bwd(hin, p["input_layer.layers.%d.weight" % i], g, grads, "input_layer.layers.%d" % i)
    if cfg.get("fc_input_norm"):
        torch_ln_bwd(cache["in_ln"], p["input_layer.input_norm.weight"], g, grads, "input_layer.input_norm")
    return grads


class Adam(object):
    """torch.optim.Adam (betas .9/.999, eps 1e-8, amsgrad off) restated: ``weight_decay`` adds wd * p to the gradient (torch.optim.Adam's
    L2 term); ``decoupled=True`` is torch.optim.AdamW: p <- p (1 - lr wd) before the moment update, the gradient untouched."""
    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.wd, self.decoupled = weight_decay, decoupled
        self.m = {k: np.zeros_like(v) for k, v in params.items()}
        self.v = {k: np.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k in params:
            g = grads[k].astype(params[k].dtype)
            w = params[k]
            if self.wd and self.decoupled:
                w = w * (1 - self.lr * self.wd)
            elif self.wd:
                g = g + self.wd * w
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            denom = np.sqrt(self.v[k]) / math.sqrt(bc2) + self.eps
            params[k] = (w - (self.lr / bc1) * self.m[k] / denom).astype(params[k].dtype)


def train_step(p, cfg, opt, x, y, loss_fn):
    """One training step of train_utils.py:18-29: mask, forward, loss, backward, optimizer step."""
    mask = y == -1
    scores, cache = forward(p, cfg, x, mask)
    out = loss_fn(scores, y)
    loss, gs = out[0], out[1]
    grads = backward(p, cfg, cache, gs.astype(scores.dtype))
    opt.step(p, grads)
    return loss, scores
