"""CPU-torch restatement of the reference training step  --  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

bench.py's ``cpu_baseline`` leg needs "the reference CPU path timed on the same box's host cores".  The reference tree is
not on the GPU box, and the numpy oracle (oracle/model_oracle.py) is written for checkability, not speed.  This module
states the same computation with the torch CPU operators the reference itself calls (F.softmax,
torch.matmul, Tensor.sort, nn.Linear's addmm, autograd, torch.optim.Adam), so its timing on N host cores is what
allRank's own ``loss_batch`` (allrank/training/train_utils.py:18-29) costs there:

    FCModel (model.py:35-44) -> N x [x + attn(LN(x)); x + ffn(LN(x))] (transformer.py:98-134) -> LN -> Linear(d, 1)
    -> approxNDCGLoss (approxNDCG.py:7-53) / listNet (listNet.py:8-30) -> backward -> Adam

Pinned by tests/test_oracle_pinned.py::test_torch_port_matches_numpy_oracle (scores, loss, gradients vs the numpy
oracle, which is itself pinned to the reference's golden vectors).  Parameters use the reference's state_dict keys.
"""
import math

import torch
import torch.nn.functional as F


def make_params(np_params):
    return {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in np_params.items()}


def _ln(x, a, b, eps=1e-6):
    # transformer.py:73-81: unbiased std, eps added to std
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def forward(p, cfg, x, mask):
    h = x
    for i in range(len(cfg["fc_sizes"])):
        h = F.linear(h, p["input_layer.layers.%d.weight" % i], p["input_layer.layers.%d.bias" % i])
        if cfg.get("fc_activation") == "ReLU":
            h = torch.relu(h)
    B, L, d = h.shape
    H = cfg["h"]
    dk = d // H
    for n in range(cfg["N"]):
        pre = "encoder.layers.%d." % n
        xn = _ln(h, p[pre + "sublayer.0.norm.a_2"], p[pre + "sublayer.0.norm.b_2"])
        q, k, v = [F.linear(xn, p[pre + "self_attn.linears.%d.weight" % j], p[pre + "self_attn.linears.%d.bias" % j])
                   .view(B, L, H, dk).transpose(1, 2) for j in range(3)]
        sc = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)                      # transformer.py:148-149
        sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))                      # :150-151
        o = torch.matmul(F.softmax(sc, dim=-1), v).transpose(1, 2).contiguous().view(B, L, d)
        h = h + F.linear(o, p[pre + "self_attn.linears.3.weight"], p[pre + "self_attn.linears.3.bias"])
        xn = _ln(h, p[pre + "sublayer.1.norm.a_2"], p[pre + "sublayer.1.norm.b_2"])
        r = torch.relu(F.linear(xn, p[pre + "feed_forward.w_1.weight"], p[pre + "feed_forward.w_1.bias"]))
        h = h + F.linear(r, p[pre + "feed_forward.w_2.weight"], p[pre + "feed_forward.w_2.bias"])
    if cfg["N"]:
        h = _ln(h, p["encoder.norm.a_2"], p["encoder.norm.b_2"])
    return F.linear(h, p["output_layer.w_1.weight"], p["output_layer.w_1.bias"]).squeeze(2)


def approx_ndcg(s, y, eps=1e-10, alpha=1.0):
    """approxNDCG.py:7-53 with dense [B, L, L] intermediates, as the reference evaluates it"""
    pad = y == -1
    s = s.masked_fill(pad, float("-inf"))
    y = y.masked_fill(pad, float("-inf"))
    s_sorted, idx = s.sort(descending=True, dim=-1)                 # :27
    y_sorted, _ = y.sort(descending=True, dim=-1)                   # :28
    tsp = torch.gather(y, 1, idx)                                   # :31
    pair = torch.isfinite(tsp[:, :, None] - tsp[:, None, :])        # :32-33 (both items valid)
    L = s.shape[1]
    pair = pair & ~torch.eye(L, dtype=torch.bool)[None]             # :34
    tsp = tsp.clamp(min=0.0)                                        # :37
    y_sorted = y_sorted.clamp(min=0.0)                              # :38
    D = torch.log2(torch.arange(1, L + 1, dtype=torch.float32) + 1.0)[None, :]       # :42
    max_dcg = ((torch.pow(2.0, y_sorted) - 1) / D).sum(-1).clamp(min=eps)          # :43
    G = (torch.pow(2.0, tsp) - 1) / max_dcg[:, None]                                # :44
    sd = (s_sorted[:, :, None] - s_sorted[:, None, :]).masked_fill(~pair, 0.0)      # :47-48
    approx_pos = 1.0 + (pair.float() * torch.sigmoid(-alpha * sd).clamp(min=eps)).sum(-1)   # :49
    return -(G / torch.log2(1.0 + approx_pos)).sum(-1).mean()       # :50-53


def list_net(s, y, eps=1e-10):
    pad = y == -1
    s = s.masked_fill(pad, float("-inf"))
    y = y.masked_fill(pad, float("-inf"))
    ps, pt = F.softmax(s, dim=1), F.softmax(y, dim=1)
    return -(pt * torch.log(ps + eps)).sum(1).mean()


LOSSES = {"approxNDCGLoss": approx_ndcg, "listNet": list_net}


class Stepper(object):
    def __init__(self, np_params, cfg, loss="approxNDCGLoss", lr=1e-3):
        self.p, self.cfg = make_params(np_params), cfg
        self.loss = LOSSES[loss]
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr)

    def step(self, x, y):
        mask = y == -1
        loss = self.loss(forward(self.p, self.cfg, x, mask), y)
        loss.backward()
        self.opt.step()
        self.opt.zero_grad()
        return loss.item()
