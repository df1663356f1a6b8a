"""Training-step driver for the hot path: the body of ``loss_batch`` (allrank/training/train_utils.py:18-29) without
its per-step host syncs, on device-resident data, optionally slate-sharded across ranks.

    mask = (yb == -1); loss = loss_func(model(xb, mask, indices), yb); loss.backward(); [clip]; opt.step(); opt.zero_grad()

Differences from the reference driver (all outside the arithmetic): no ``loss.item()`` per step (the loss stays on
the device; call ``.item()`` when you want it), gradients live in one flat buffer (allrank_amd.parallel), and under
``world_size > 1`` the loss is normalised by the global batch and gradients are summed over RCCL.
"""
import ctypes

import torch
from torch.nn.utils import clip_grad_norm_

from . import parallel, sharding

PADDED_Y_VALUE = -1


class Trainer(object):
    def __init__(self, model, loss_func, optimizer, gradient_clipping_norm=None, world_size=1, group=None):
        self.model = model
        self.loss_func = loss_func
        self.opt = optimizer
        self.clip = gradient_clipping_norm
        self.world = world_size
        self.group = group
        self.flat = parallel.FlatGradients(model.parameters(), group)

    def step(self, xb, yb, indices=None, global_batch=None):
        """one training step on this rank's slates; returns the (device) loss tensor -- this rank's share of the
        global loss when sharded."""
        self.flat.check()         # .grad must still alias the flat buffer (an external zero_grad(set_to_none=True) detaches it)
        mask = (yb == PADDED_Y_VALUE)
        gb = global_batch if global_batch is not None else xb.shape[0] * self.world
        if self.world > 1:
            with sharding.shard_context(gb, self.group):
                out = self.model(xb, mask, indices)
                loss = self.loss_func(out, yb)
        else:
            out = self.model(xb, mask, indices)
            loss = self.loss_func(out, yb)
        # scores of THIS forward (model.py:82-92: d_output > 1 sums the last axis), for train metrics without a second pass
        self.last_scores = out.detach() if out.dim() == 2 else out.detach().sum(-1)
        loss.backward()
        if self.world > 1:
            self.flat.all_reduce()
        if self.clip:
            clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        self.flat.zero()          # == opt.zero_grad(set_to_none=False): grads stay views of the flat buffer
        return loss


# ------------------------------------------------------------------------------------------------------------------
# FusedTrainer: the same training step, written out explicitly (no autograd) on preallocated buffers
# ------------------------------------------------------------------------------------------------------------------
class FusedTrainer(object):
    """One training step of train_utils.py:18-29 as a fixed launch sequence on persistent HBM buffers:

        FC -> N x [LN -> QKV GEMM -> fused attention -> out GEMM(+res) -> LN -> FFN GEMMs(+ReLU, +res)] -> LN
           -> score head -> fused listwise loss (value + d/dscores) -> hand-written backward -> [RCCL all-reduce]
           -> fused Adam over ONE flat parameter buffer.

    * parameters / gradients / Adam moments are single flat fp32 buffers; the nn.Module's parameters are re-pointed
      at views of the flat buffer, so ``model.state_dict()`` / ``model.score()`` always see the trained weights, and
      the Q,K,V projection weights of a layer sit adjacently => ONE [3d, d] GEMM feeds the attention kernel in place.
    * every gradient is written exactly once per step straight into the flat gradient buffer (GEMM ``out=`` views):
      no zero_grad pass, no accumulation kernels, one collective for multi-GPU.
    * every launch of the sequence is a libltrx kernel: the dense projections are the split-bf16 MFMA GEMMs
      (ltrx_gemm_nt / ltrx_gemm_tn; ``gemm="hipblaslt"`` swaps in torch.mm/addmm as a comparison arithmetic).
    * the whole sequence is captured in a hipGraph after warm-up (``use_graph=True``) -- at 64 slates/GPU the step is
      ~100 launches of 10-300 us, so launch latency matters.
    * dropout (every shipped transformer config trains with 0.1-0.4) is counter-based: a mask element is a hash of
      (site seed ^ per-step device word, element index), generated inside the producing kernel's epilogue (GEMM bias+
      ReLU+dropout, the residual sum in the epilogue of the projection that closes a sublayer, the attention probabilities)
      and REGENERATED in the backward
      -- no mask tensors, and a replayed hipGraph draws fresh masks because the step word lives in device memory.
    Supported model family: FCModel (optional input_norm = nn.LayerNorm, activation None / ReLU, or Sigmoid / Tanh without FC
    dropout) -> optional encoder with
    optional fixed / learned positional encoding (positional.py:15-77) -> OutputLayer(any d_output, activation None / Sigmoid /
    Tanh; d_output > 1 feeds the ``ordinal`` loss, ``scores`` is then the sum over the output units, model.py:119-128).
    Other FC / output activations raise NotImplementedError (use ``Trainer``).
    """

    def __init__(self, model, loss_name, loss_args, B, L, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, world_size=1, group=None,
                 optimizer="Adam", weight_decay=0.0, momentum=0.0, nesterov=False,
                 use_graph=True, gemm="split_bf16", dropout=True, seed=None, gradient_clipping_norm=None, compact=False,
                 weight_images=True, fc_step=True, group_wgrad=True, relu_bits=True, pad_input=True, force_dist=False):
        """gemm: "split_bf16" -- libltrx fp32-accurate GEMMs on the bf16 MFMA (3 products), "split_bf16_strict" (6
        products), "hipblaslt" (torch.mm/addmm, exact-fp32 library GEMMs), or "bf16" -- the THROUGHPUT mode: one bf16
        product per contraction in the dense projections AND in attention (fp32 storage, accumulation, LayerNorm, softmax,
        loss and Adam; about 2^-9 relative error per product, outside the 1e-5 parity contract -- bench.py reports it
        on its own line with its measured loss error).  dropout=False trains with every
        nn.Dropout of the model disabled; seed keys the dropout masks (default: torch.initial_seed(), nothing is drawn from the global generator);
        gradient_clipping_norm: clip_grad_norm_ of train_utils.py:24-25 (the coefficient stays on the device).
        compact=True: variable-length execution -- the valid items of each padded batch (dataset.py:28-38) are packed into
        consecutive rows, every row-wise kernel runs over the packed rows only, attention reads per-slate extents from
        cu_seqlens, and scores / d loss/d scores move between the packed rows and the padded [B, L] grid the loss kernels
        work on.  Same loss and gradients as the padded step (padded rows carry no gradient and are masked as keys); the
        row count changes per batch, so this mode runs eagerly (no hipGraph).
        optimizer: "Adam" (betas, eps, weight_decay = L2 term), "AdamW" (decoupled weight decay) or "SGD" (momentum, nesterov,
        weight_decay; dampening 0) -- torch.optim's update rules (main.py:82) in one flat-buffer kernel.
        weight_images=False: the GEMMs split the weight operand on the fly in every tile instead of reading the per-step pre-split
        images (same results bit for bit; kept for A/B measurements).
        fc_step=True: a model that is FCModel([H]) -> OutputLayer(H, 1) with the listNet loss (BASELINE configs[1]) trains through
        the slate-resident step of ltrx_fc_listnet_step -- forward, loss, backward and Adam in two launches that read the batch
        from HBM once, straight from the caller's tensors (no staging copy, no hipGraph needed); False keeps the GEMM launch
        sequence for A/B runs; "collapse" (opt-in, FC activation None only): the linear scorer's two layers evaluated as ONE
        matrix-vector product per slate with the exact rank-1 gradients (ltrx_fc_linear_listnet_step: fp32 FMAs, the slate in registers,
        HBM-bound).  ``self.fcstep`` tells which one is active (False / True / "collapse").
        group_wgrad=True: the four weight gradients of an encoder layer run as one ltrx_gemm_tn_group launch (4x fewer partial slabs
        to write and reduce); False = one ltrx_gemm_tn per projection (A/B runs).
        relu_bits=True: the feed-forward ReLU(+dropout) mask travels from the forward GEMM to the input-gradient GEMM as one bit per
        element (ltrx_gemm_nt acts 4 / 5) instead of being re-read from the saved fp32 activation; same results bit for bit.
        pad_input=True: the static input buffer keeps the features in rows padded to 256 floats so that the first FC layer (F = 136 is
        no multiple of the GEMM's 32-column step) runs the large-tile forward and weight-gradient kernels; False = dense rows (A/B).
        (Round 6: the two opt-in NEGATIVE RESULTS of round 5 -- ``overlap_wgrad`` (fork / join of the weight-gradient launches on a
        second stream: 1-2 % slower, profiles/r05_wgrad_overlap_ab.md) and ``act_images`` (activations as pre-split operand images:
        VALU -6 ... -28 % in the consumers, cycles unchanged, step -1 %, profiles/r05_act_images_ab.md) -- no longer live in this class;
        tools/lab/patches/r06_engine_overlap_wgrad_act_images.patch re-adds them; the image-form kernel entry points they drove (ltrx_gemm_nt_img,
        ltrx_gemm_tn_group_img, ltrx_layernorm_fwd_image) stay in the library with their kernel-level bit-identity test.)
        force_dist=True: see ``self.sharded`` below."""
        import torch.nn as nn
        from . import _lib as LB
        from .losses import FusedLoss
        from .model import FCModel, Encoder, LTRModel, LearnedPositionalEncoding
        self.LB = LB
        self.lib = LB.lib()
        if gemm not in ("split_bf16", "split_bf16_strict", "hipblaslt", "bf16"):
            raise ValueError("gemm must be split_bf16, split_bf16_strict, hipblaslt or bf16")
        self.gemm = gemm
        self.weight_images = bool(weight_images)
        self.group_wgrad = bool(group_wgrad) and gemm != "split_bf16_strict"     # (the strict arithmetic has no large-tile kernel)
        self._wg_pending = []
        self.wgrad_group_log = []             # (problems, grouped kernel ran?) of the most recent _wgrad_flush calls
        self._wg_probe = (ctypes.c_ubyte * 65536)()
        self._red_pending = []                # (src ptr, partial rows, row stride, columns, dst ptr) entries of the next _reduce_flush
        self._ln_slot = 0
        self._prec = {"split_bf16_strict": 1, "bf16": 2}.get(gemm, 0)        # precision code of ltrx_gemm_nt / ltrx_gemm_tn
        # attention arithmetic of THIS trainer, passed with every ltrx_mha_fwd / ltrx_mha_bwd call (the library keeps no mode):
        # 1 = three bf16 products (fp32-class, parity), 2 = one product (the "bf16" throughput mode)
        self._mha_mode = 2 if gemm == "bf16" else 1
        self._wT = {}
        self.model = model
        self.B, self.L, self.M = B, L, B * L
        self.lr, self.betas, self.eps = lr, betas, eps
        if optimizer not in ("Adam", "AdamW", "SGD"):
            raise NotImplementedError("FusedTrainer: optimizer %r (Adam, AdamW and SGD are fused)" % (optimizer,))
        self.optimizer, self.weight_decay = optimizer, float(weight_decay)
        self.momentum, self.nesterov = float(momentum), bool(nesterov)
        self.world, self.group = world_size, group
        # the step takes its SHARDED form (global divisor through shard_context, bucketed all-reduce of the flat gradient behind the
        # backward, normaliser all-reduces, hipGraph segments cut at every collective) when there is more than one rank -- or when
        # ``force_dist`` asks for it on a one-rank group: the whole collective path, RCCL included, on the one GPU a build box has
        # (bench.py --force-dist, tests/test_gpu_rccl.py).  Needs an initialised process group.
        self.sharded = bool(world_size > 1 or force_dist)
        if not isinstance(model, LTRModel) or not isinstance(model.input_layer, FCModel):
            raise NotImplementedError("FusedTrainer needs an allrank_amd LTRModel with an FCModel input block")
        fc = model.input_layer
        self.in_norm = None if isinstance(fc.input_norm, nn.Identity) else fc.input_norm      # nn.LayerNorm (model.py:27)
        self.p_fc = float(fc.dropout.p) if dropout else 0.0
        if seed is None:
            # derived from the seed torch's global generator was given (main.py:36: torch.manual_seed(42)) WITHOUT drawing from it: the
            # loaders' shuffle order comes from that generator, and a draw here would shift every epoch's batches away from the
            # reference's (round 6: the trajectory fixtures of tests/golden/make_golden_trajectory.py)
            seed = int(torch.initial_seed() & 0x7FFFFFFF)
        self._seed = (int(seed) * 0x9E3779B1 + 0x7F4A7C15 * (1 + self._rank(group, world_size))) & 0xFFFFFFFF
        if isinstance(fc.activation, nn.Identity):
            self.fc_act = 0
        elif isinstance(fc.activation, nn.ReLU):
            self.fc_act = 1
        elif isinstance(fc.activation, (nn.Sigmoid, nn.Tanh)) and self.p_fc == 0.0:
            # (model.py:28-29 resolves any torch.nn name; Sigmoid / Tanh run as an elementwise pass after the GEMM whose derivative is
            #  taken from the stored output -- with dropout after them the stored output no longer determines it: autograd Trainer)
            self.fc_act = 3 if isinstance(fc.activation, nn.Sigmoid) else 4
        else:
            raise NotImplementedError("FusedTrainer: FC activation %r%s" % (fc.activation, " with dropout" if self.p_fc else ""))
        enc = model.encoder if isinstance(model.encoder, Encoder) else None
        self.pos = enc.position if (enc is not None and enc.position is not None) else None
        self.pos_learned = isinstance(self.pos, LearnedPositionalEncoding)
        out = model.output_layer
        self.n_out = int(out.d_output)                # > 1: ordinal configs (model.py:111-128: forward [B, L, d_output], score = sum)
        if isinstance(out.activation, nn.Identity):
            self.out_act = 0
        elif isinstance(out.activation, nn.Sigmoid):
            self.out_act = 1
        elif isinstance(out.activation, nn.Tanh):
            self.out_act = 2
        else:
            raise NotImplementedError("FusedTrainer: output activation %r" % (out.activation,))
        dev = next(model.parameters()).device
        self.dev = dev
        self.enc = enc
        self.nfc = len(fc.layers)
        self.fc_sizes = [fc.layers[0].in_features] + [l.out_features for l in fc.layers]
        self.d = self.fc_sizes[-1]
        self.N = len(enc.layers) if enc is not None else 0
        if enc is not None:
            l0 = enc.layers[0]
            self.h = l0.self_attn.h
            self.dff = l0.feed_forward.w_1.out_features
            self.ln_eps = enc.norm.eps

        # ---- flat parameter layout (16-byte aligned segments; q,k,v weights and biases adjacent) ----
        order = []
        if self.in_norm is not None:
            order += [self.in_norm.weight, self.in_norm.bias]
        for lyr in fc.layers:
            order += [lyr.weight, lyr.bias]
        if self.pos_learned:
            order += [self.pos.pe.weight]
        if enc is not None:
            for lay in enc.layers:
                lin = lay.self_attn.linears
                order += [lin[0].weight, lin[1].weight, lin[2].weight, lin[0].bias, lin[1].bias, lin[2].bias,
                          lin[3].weight, lin[3].bias, lay.feed_forward.w_1.weight, lay.feed_forward.w_1.bias,
                          lay.feed_forward.w_2.weight, lay.feed_forward.w_2.bias,
                          lay.sublayer[0].norm.a_2, lay.sublayer[0].norm.b_2, lay.sublayer[1].norm.a_2, lay.sublayer[1].norm.b_2]
            order += [enc.norm.a_2, enc.norm.b_2]
        order += [out.w_1.weight, out.w_1.bias]
        assert len(order) == len(list(model.parameters()))
        offs, n = [], 0
        for p in order:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.nflat = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.clip = float(gradient_clipping_norm) if gradient_clipping_norm else None
        self.clip_scale = torch.ones(1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.ws_clip = torch.empty(max(self.lib.ltrx_clip_workspace_bytes(n), 64), dtype=torch.uint8, device=dev)
        self.drop_step = torch.zeros(1, dtype=torch.int32, device=dev)       # u32 word folded into every dropout seed
        # slate-resident FC + ListNet step (csrc/ltrx_fcstep.hip): eligibility.  It reads the padded batch in place and masks padded
        # items itself, so a request for variable-length execution is moot for such a job (and would only route it to the slower
        # GEMM launch sequence): compact is dropped.
        fc_ok = bool(
            fc_step and self.N == 0 and self.nfc == 1 and self.in_norm is None and self.pos is None and self.fc_act in (0, 1)
            and self.p_fc == 0.0 and self.n_out == 1 and self.out_act == 0 and loss_name == "listNet"
            and gemm == "split_bf16" and optimizer in ("Adam", "AdamW")
            and self.lib.ltrx_fc_listnet_supported(L, self.fc_sizes[0], self.fc_sizes[1]))
        if fc_ok:
            if compact or not use_graph:
                import logging
                logging.getLogger("allrank_amd.engine").info(
                    "FusedTrainer: the slate-resident FC + ListNet step is taken (two launches per step, the padded batch read in place): "
                    "compact=%s / use_graph=%s do not apply to it; scores / labels of the last step alias the caller's tensors until the "
                    "next step()", compact, use_graph)
            compact = False
        self.compact = bool(compact)
        self.rows = B * L                                                     # rows the row-wise kernels run over
        self.n_valid = B * L
        # cu_seqlens of the packed batch and the attention launch order (longest slate first), one buffer / one copy
        self._cuord = torch.zeros(2 * B + 1, dtype=torch.int32, device=dev) if compact else None
        self.cu = self._cuord[:B + 1] if compact else None
        self.order = self._cuord[B + 1:] if compact else None
        self.idx = torch.zeros(B * L, dtype=torch.int32, device=dev) if compact else None  # packed row -> padded row
        if compact:
            self._cu_ring = [(torch.zeros(2 * B + 1, dtype=torch.int32).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self._pack_turn = 0
        # gradient buckets for the multi-GPU all-reduce, in the order the backward completes them: the tail of the flat
        # buffer (last encoder layer + final norm + head) first, then one bucket per earlier layer, the FC stack last
        offs_of = {id(p): o for p, o in zip(order, offs)}
        starts = [offs_of[id(lay.self_attn.linears[0].weight)] for lay in enc.layers] if enc is not None else []
        self._buckets = []
        prev_hi = n
        for st_ in reversed(starts):
            self._buckets.append((st_, prev_hi))
            prev_hi = st_
        self._buckets.append((0, prev_hi))
        self._works = []
        self.comm_enabled = True      # bench.py measures the exposed part of the collective by switching it off
        self._pv, self._wv, self._gv = {}, {}, {}
        self._order = order
        with torch.no_grad():
            for p, o in zip(order, offs):
                view = self.flat_p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view_as(p)
                self._pv[id(p)] = (o, p.shape)
                self._wv[id(p)] = view
                self._gv[id(p)] = p.grad

        # the kernels address the flat buffers through these views, never through p.data / p.grad: an external
        # zero_grad(set_to_none=True) or .to() cannot redirect the step (step() re-attaches the module's views)
        def W(p):
            return self._wv[id(p)]

        def G(p):
            return self._gv[id(p)]

        self.W, self.G = W, G

        def fused_view(buf, first, count_rows, cols=None):
            o, _ = self._pv[id(first)]
            if cols is None:
                return buf[o:o + count_rows]
            return buf[o:o + count_rows * cols].view(count_rows, cols)

        M, d = self.M, self.d
        f32 = dict(dtype=torch.float32, device=dev)
        # The input features: rows of F floats, or -- where the first FC layer can then run the large-tile GEMMs -- rows padded to a
        # multiple of 256 floats (zeros): the forward projection contracts over F rounded up to the kernel's 32-column step against a
        # row-padded copy of W_0 (refreshed with the weight images), the weight gradient reads the padded rows as operand tiles.
        F0 = self.fc_sizes[0]
        self._x_pad = bool(pad_input and self.in_norm is None and gemm in ("split_bf16", "bf16") and F0 % 4 == 0 and F0 % 256 != 0
                           and F0 <= 1024 and self.fc_sizes[1] % 256 == 0 and M >= 2048)
        if self._x_pad:
            self.x_in_buf = torch.zeros((M, (F0 + 255) // 256 * 256), **f32)
            self.x_in = self.x_in_buf[:, :F0]
            kp = (F0 + 31) // 32 * 32
            self.x_in_k = self.x_in_buf[:, :kp]
            self.w0_pad = torch.zeros((self.fc_sizes[1], kp), **f32)
            self.w0_pad_i = torch.zeros_like(self.w0_pad)
        else:
            self.x_in = torch.zeros((M, F0), **f32)
        if self.in_norm is not None:
            F_ = self.fc_sizes[0]
            self.x_norm = torch.zeros((M, F_), **f32)
            self.mean_in = torch.zeros(M, **f32)
            self.rstd_in = torch.zeros(M, **f32)
            self.d_in = torch.zeros((M, F_), **f32)
            self.ws_ln_in = torch.empty(max(self.lib.ltrx_layernorm_bwd_workspace_bytes(M, F_), 64), dtype=torch.uint8, device=dev)
        if self.pos is not None:
            self.x_pe = torch.zeros((M, d), **f32)
            self.idx_rows = torch.full((M,), -1, dtype=torch.int64, device=dev)      # original rank of every row (-1: none)
            if self.pos_learned:
                self.pos_pad = int(self.pos.pe.padding_idx)
            else:
                self.pos_pad = int(self.pos.padding_idx)
        self.y_in = torch.zeros((B, L), **f32)
        self.mask = torch.zeros((B, L), dtype=torch.uint8, device=dev)
        self.fc_out = [torch.zeros((M, s), **f32) for s in self.fc_sizes[1:]]
        self.layers = []
        for i in range(self.N):
            lay = enc.layers[i]
            lin = lay.self_attn.linears
            st = dict(
                wqkv=fused_view(self.flat_p, lin[0].weight, 3 * d, d), bqkv=fused_view(self.flat_p, lin[0].bias, 3 * d),
                gwqkv=fused_view(self.flat_g, lin[0].weight, 3 * d, d), gbqkv=fused_view(self.flat_g, lin[0].bias, 3 * d),
                xsum0=None if i == 0 else torch.zeros((M, d), **f32),      # residual stream entering the layer
                xn0=torch.zeros((M, d), **f32), mean0=torch.zeros(M, **f32), rstd0=torch.zeros(M, **f32),
                qkv=torch.zeros((M, 3 * d), **f32), o=torch.zeros((M, d), **f32), lse=torch.zeros((B, self.h, L), **f32),
                x1=torch.zeros((M, d), **f32), xn1=torch.zeros((M, d), **f32), mean1=torch.zeros(M, **f32),
                rstd1=torch.zeros(M, **f32), r=torch.zeros((M, self.dff), **f32), mod=lay,
                # dropout sites (transformer.py:105,155,227): probabilities and seeds
                p_att=float(lay.self_attn.dropout.p) if dropout else 0.0,
                p_ff=float(lay.feed_forward.dropout.p) if dropout else 0.0,
                p_s0=float(lay.sublayer[0].dropout.p) if dropout else 0.0,
                p_s1=float(lay.sublayer[1].dropout.p) if dropout else 0.0,
                s_att=self._site(4 * i), s_ff=self._site(4 * i + 1), s_s0=self._site(4 * i + 2), s_s1=self._site(4 * i + 3))
            self.layers.append(st)
        if relu_bits and self.N and self.dff % 256 == 0 and d % 32 == 0 and gemm not in ("hipblaslt", "split_bf16_strict"):
            # one bit per feed-forward activation (written by the forward GEMM, read by the input-gradient GEMM instead of r)
            for st in self.layers:
                st["rbits"] = torch.zeros(((M + 255) // 256) * (self.dff // 256) * 8192, dtype=torch.uint8, device=dev)
        # (no active dropout site -> no mask counter to advance: one launch less per step)
        self._any_dropout = bool(self.p_fc or any(st[k] for st in self.layers for k in ("p_att", "p_ff", "p_s0", "p_s1")))
        if self.N:
            self.xsum_f = torch.zeros((M, d), **f32)
            self.xf = torch.zeros((M, d), **f32)
            self.mean_f = torch.zeros(M, **f32)
            self.rstd_f = torch.zeros(M, **f32)
            self.d_r = torch.zeros((M, self.dff), **f32)
            self.dqkv = torch.zeros((M, 3 * d), **f32)
            self.d_o = torch.zeros((M, d), **f32)
            self.tmp_d = torch.zeros((M, d), **f32)
            self.d_br = torch.zeros((M, d), **f32)            # gradient of a dropped residual branch (ds * keep)
            self.ws_ln = torch.empty(max(self.lib.ltrx_layernorm_bwd_workspace_bytes(M, d), 64), dtype=torch.uint8, device=dev)
            # deferred parameter-gradient partials of up to three LayerNorm backwards per encoder layer (_reduce_flush)
            self.ws_ln_g = [self.ws_ln] + [torch.empty_like(self.ws_ln) for _ in range(2)]
            self.ws_mha = torch.empty(max(self.lib.ltrx_mha_bwd_workspace_bytes(B, L, self.h, self.d // self.h, self._mha_mode), 64), dtype=torch.uint8, device=dev)
        no = self.n_out
        self.scores_raw = torch.zeros((B, L) if no == 1 else (B, L, no), **f32)      # what the loss sees (model.forward)
        self.scores = self.scores_raw if no == 1 else torch.zeros((B, L), **f32)      # model.score (sum over the output units)
        if compact:
            self.scores_c = torch.zeros(M if no == 1 else (M, no), **f32)
            self.dsc_c = torch.zeros(M if no == 1 else (M, no), **f32)
        if no > 1:
            npad = (no + 3) // 4 * 4
            self.dz_pad = torch.zeros((M, npad), **f32)        # d loss / d pre-activation, K padded to a multiple of 4 for the dgrad GEMM
            self.woutT_pad = torch.zeros((d, npad), **f32)     # W_out^T zero-padded the same way (refreshed after every optimizer step)
            self.zb = torch.zeros(no, **f32)
        self.d_a = torch.zeros((M, d), **f32)                 # gradient w.r.t. the residual stream (ping)
        self.d_b = torch.zeros((M, d), **f32)                 # (pong)
        maxn = max([3 * d, self.dff if self.N else 0] + self.fc_sizes[1:])
        self.ws_col = torch.empty(max(self.lib.ltrx_colsum_workspace_bytes(M, maxn), 64), dtype=torch.uint8, device=dev)
        self.ws_head = torch.empty(max(self.lib.ltrx_score_head_bwd_workspace_bytes(M, d), 64), dtype=torch.uint8, device=dev)
        self.fc_dgrad = [torch.zeros((M, s), **f32) for s in self.fc_sizes[1:-1]]
        big = max([(3 * d) * d, (self.dff * d) if self.N else 0] + [a * b for a, b in zip(self.fc_sizes[:-1], self.fc_sizes[1:])])
        self._fused_images = False
        if self.gemm != "hipblaslt":
            nb = 0
            shapes = [(s1, s0) for s0, s1 in zip(self.fc_sizes[:-1], self.fc_sizes[1:])]
            if self.N:
                shapes += [(3 * d, d), (d, d), (self.dff, d), (d, self.dff)]
            if self.n_out > 1:
                shapes += [(self.n_out, d)]
            for (npp, kpp) in shapes:
                nb = max(nb, self.lib.ltrx_gemm_tn_workspace_bytes(M, npp, kpp))
            if self.N and self.group_wgrad:                       # the four projections of a layer in one launch (_wgrad_flush)
                # ... or, when both sublayer dropouts are on (the dropout buffer d_br is reused between the branches), as two groups of
                # two: fewer tiles per group -> more row splits per problem -> possibly MORE workspace than the group of four.  The
                # buffer covers every composition the step issues, in issue order
                w2, w1, wo, wq = (d, self.dff), (self.dff, d), (d, d), (3 * d, d)
                comps = [[w2, w1, wo, wq]]
                if any(st["p_s0"] and st["p_s1"] for st in self.layers):
                    comps += [[w2, w1], [wo, wq]]
                for comp in comps:
                    npa = (ctypes.c_int * len(comp))(*[c[0] for c in comp])
                    kpa = (ctypes.c_int * len(comp))(*[c[1] for c in comp])
                    nb = max(nb, self.lib.ltrx_gemm_tn_group_workspace_bytes(len(comp), M, npa, kpa))
            self.ws_tn = torch.empty(max(nb, 64), dtype=torch.uint8, device=dev)
            # transposed weight copies for the input-gradient GEMMs (refreshed after every optimizer step by ONE batched
            # transpose launch): all copies live in one flat buffer, the descriptor table is built once
            tw = [l.weight for l in (fc.layers if self.in_norm is not None else fc.layers[1:])]
            srcs = [(self._pv[id(p)][0], p.shape[0], p.shape[1], ("w", id(p))) for p in tw]
            if enc is not None:
                for li, st in enumerate(self.layers):
                    lay = st["mod"]
                    for p in (lay.self_attn.linears[3].weight, lay.feed_forward.w_1.weight, lay.feed_forward.w_2.weight):
                        srcs.append((self._pv[id(p)][0], p.shape[0], p.shape[1], ("w", id(p))))
                    srcs.append((self._pv[id(lay.self_attn.linears[0].weight)][0], 3 * d, d, ("qkv", li)))
            tot = sum((r * c + 3) // 4 * 4 for _, r, c, _ in srcs)
            self.flat_t = torch.zeros(max(tot, 4), **f32)
            desc, tstart, o = [], [0], 0
            for (so, r, c, key) in srcs:
                view = self.flat_t[o:o + r * c].view(c, r)
                if key[0] == "w":
                    self._wT[key[1]] = view
                else:
                    self.layers[key[1]]["wqkvT"] = view
                desc += [so, o, r, c]
                tstart.append(tstart[-1] + ((r + 31) // 32) * ((c + 31) // 32))
                o += (r * c + 3) // 4 * 4
            self._tdesc = torch.tensor(desc if desc else [0, 0, 0, 0], dtype=torch.int64, device=dev)
            self._tstart = torch.tensor(tstart, dtype=torch.int32, device=dev)
            self._tn, self._ttiles = len(srcs), tstart[-1]
            # pre-split bf16 hi / lo IMAGES of the weights and of their transposes (same offsets as flat_p / flat_t): what the
            # large-tile NT GEMMs stage as operand B without splitting it again in every tile (ltrx_split_image, include/ltrx.h)
            self.flat_pi = torch.empty_like(self.flat_p)
            self.flat_ti = torch.zeros_like(self.flat_t)
            # one-launch refresh (ltrx_weight_images) when every transposed matrix keeps image groups of 4 inside a row
            self._fused_images = self.nflat % 4 == 0 and all(r % 4 == 0 for _, r, _, _ in srcs)
            self._refresh_transposes()
        self.loss = FusedLoss(loss_name, B, L, dev, **(loss_args or {}))
        if (self.n_out > 1) != (loss_name == "ordinal") or (loss_name == "ordinal" and int(loss_args["n"]) != self.n_out):
            raise NotImplementedError("FusedTrainer: d_output > 1 goes with the ordinal loss of the same n (and only with it)")
        # ListMLE: the reference draws torch.randperm(L) on every call (listMLE.py:17).  shuffle_ties=True (default) does
        # the same with a device generator; tests that compare with the oracle set shuffle_ties=False and an explicit
        # permutation via ``trainer.loss.set_perm``.
        self.shuffle_ties = True
        self._perm_gen = torch.Generator(device=dev)
        self._perm_gen.manual_seed(int(seed) & 0x7FFFFFFF)
        self.use_graph = use_graph and not compact
        self.graph_fwd, self._warm_fwd = None, 0
        self.probe = None                                     # list collecting (start, end) events of the FFN1 GEMM (eager steps only)
        self.probe_wgrad = None                               # ... of the grouped weight-gradient launch (bench.py, eager steps only)
        # captured steps: {(batch divisor, collectives on?): [(hipGraph segment, collective to launch after it | None), ...]}
        import collections
        self._graphs = collections.OrderedDict()
        self.max_graphs = 4
        self._warned_evict = False
        self.capture_fallback = None                          # repr of the capture error if a sharded run fell back to eager steps
        self._seg_break = None                                # set while capturing: ends the current segment (see _capture)
        self._graph_pool = None
        self._cap_stream = None
        self._warm = 0
        self._wver = [p._version for p in self._order]
        self._images_stale = False
        self.y_cur = self.y_in                                # labels of the last step() (the caller's tensor in the fcstep path)
        self.keep_fc_out = False                              # tests: also write the FC activations to fc_out[0]
        self.keep_loss_grad = False                           # tests: also write d loss / d scores to self.loss.grad (4 B per item)
        self.fcstep = fc_ok
        if self.fcstep and fc_step == "collapse" and self.fc_act == 0:
            self.fcstep = "collapse"
        if self.fcstep:
            F_, H_ = self.fc_sizes[0], self.fc_sizes[1]
            H4 = (H_ + 3) // 4 * 4
            offs_fc = (0, H_ * F_, H_ * F_ + H4, H_ * F_ + 2 * H4)
            assert tuple(offs) == offs_fc and self.nflat == offs_fc[3] + 4, (offs, self.nflat)
            self._fc_ws = torch.empty(max(self.lib.ltrx_fc_listnet_workspace_bytes(B, L, F_, H_, self.nflat), 64), dtype=torch.uint8, device=dev)
            P = LB.ptr
            self._fc_a = (B, L, F_, H_, self.fc_act, P(self.flat_p), offs_fc[0], offs_fc[1], offs_fc[2], offs_fc[3], self.nflat,
                          float(self.loss.eps), float(self.loss.pad))
            if self.fcstep == "collapse":
                self._fc_ws = torch.empty(max(self.lib.ltrx_fc_linear_listnet_workspace_bytes(B, F_), 64), dtype=torch.uint8, device=dev)
                self._fc_a = self._fc_a[:4] + self._fc_a[5:]          # (no activation argument)
            self._fc_b = (P(self.scores_raw), P(self.loss.grad))

    # ---- thin launch helpers -----------------------------------------------------------------------------------
    def _st(self):
        return self.LB.launch_stream(self.dev)

    @staticmethod
    def _rank(group, world_size):
        if world_size <= 1:
            return 0
        import torch.distributed as dist
        return dist.get_rank(group)

    def _site(self, k):
        """seed of dropout site k (one per nn.Dropout instance of the model)"""
        x = (self._seed ^ ((k + 1) * 0x85EBCA6B)) & 0xFFFFFFFF
        x = ((x ^ (x >> 16)) * 0x7FEB352D) & 0xFFFFFFFF
        x = ((x ^ (x >> 15)) * 0x846CA68B) & 0xFFFFFFFF
        return x ^ (x >> 16)

    def _ln_fwd(self, x, res, a, b, xsum, y, mean, rstd, p=0.0, seed=0):
        """y = LN(x + drop_p(res)); xsum = x + drop_p(res)"""
        P = self.LB.ptr
        self.LB.check(self.lib.ltrx_layernorm_fwd(P(x), P(res), P(a), P(b), self.rows, self.d, float(self.ln_eps), P(xsum), P(y),
                                                  P(mean), P(rstd), float(p), seed, P(self.drop_step), self._st()), "layernorm_fwd")

    def _drop_apply(self, src, dst, p, seed):
        """dst = src * keep-mask/(1-p) of the site (the backward of a dropped branch)"""
        P = self.LB.ptr
        self.LB.check(self.lib.ltrx_dropout_apply(P(src), P(dst), self.rows * src.shape[1], float(p), seed, P(self.drop_step), self._st()),
                      "dropout_apply")

    def _branch_grad(self, ds, p, seed):
        if p == 0.0:
            return ds
        self._drop_apply(ds, self.d_br, p, seed)
        return self.d_br

    def _ln_bwd(self, dy, xsum, a, mean, rstd, dres, dx, da, db):
        P = self.LB.ptr
        if self.group_wgrad and self.gemm != "hipblaslt" and self._ln_slot < len(self.ws_ln_g):
            # dx now; the (da, db) partials join the layer's one reducing launch (_reduce_flush)
            import ctypes
            buf = self.ws_ln_g[self._ln_slot]
            self._ln_slot += 1
            rows_out = ctypes.c_int(0)
            self.LB.check(self.lib.ltrx_layernorm_bwd_partial(P(dy), P(xsum), P(a), P(mean), P(rstd), P(dres), self.rows, self.d,
                                                              float(self.ln_eps), P(dx), P(buf), ctypes.byref(rows_out), self._st()),
                          "layernorm_bwd_partial")
            self._red_pending.append((buf.data_ptr(), rows_out.value, 2 * self.d, self.d, da.data_ptr()))
            self._red_pending.append((buf.data_ptr() + 4 * self.d, rows_out.value, 2 * self.d, self.d, db.data_ptr()))
            return
        self.LB.check(self.lib.ltrx_layernorm_bwd(P(dy), P(xsum), P(a), P(mean), P(rstd), P(dres), self.rows, self.d,
                                                  float(self.ln_eps), P(dx), P(da), P(db), P(self.ws_ln), self._st()),
                      "layernorm_bwd")

    def _colsum(self, a, out):
        P = self.LB.ptr
        self.LB.check(self.lib.ltrx_colsum(P(a), self.rows, a.shape[1], a.stride(0), P(out), 0, P(self.ws_col), self._st()),
                      "colsum")

    def _relu_bwd(self, dr, r, p=0.0):
        """dr *= (r > 0) / (1 - p): backward of dropout(relu(z)) given the stored post-dropout activation r"""
        self.LB.check(self.lib.ltrx_relu_bwd(self.LB.ptr(dr), self.LB.ptr(r), self.rows * dr.shape[1], 1.0 / (1.0 - p), self._st()), "relu_bwd")

    def saved_activation(self, layer, key):
        """the saved activation ``key`` ("r", "xn0", "xn1", ...) of encoder layer ``layer`` of the LAST step (tests, diagnostics)"""
        return self.layers[layer][key]

    def _refresh_transposes(self):
        if self.n_out > 1:                                        # W_out^T [d, d_output] zero-padded to a multiple of 4 columns
            self.woutT_pad[:, :self.n_out].copy_(self.W(self.model.output_layer.w_1.weight).t())
        if self.gemm == "hipblaslt":
            return
        P = self.LB.ptr
        w0 = self.W(self.model.input_layer.layers[0].weight) if self._x_pad else None
        if self._fused_images:
            pad = (P(w0), w0.shape[0], w0.shape[1], self.w0_pad.shape[1], P(self.w0_pad), P(self.w0_pad_i)) if self._x_pad else \
                  (None, 0, 0, 0, None, None)
            self.LB.check(self.lib.ltrx_weight_images(P(self.flat_p), self.nflat, P(self.flat_pi), P(self.flat_t), P(self.flat_ti),
                                                      P(self._tdesc), P(self._tstart), self._tn, self._ttiles if self._tn else 0,
                                                      *pad, self._st()), "weight_images")
            return
        if self._x_pad:
            self.w0_pad[:, :w0.shape[1]].copy_(w0)
            self.LB.check(self.lib.ltrx_split_image(P(self.w0_pad), P(self.w0_pad_i), self.w0_pad.numel(), self._st()), "split_image(W0 padded)")
        if self._tn:
            self.LB.check(self.lib.ltrx_transpose_batch(P(self.flat_p), P(self.flat_t), P(self._tdesc), P(self._tstart), self._tn,
                                                        self._ttiles, self._st()), "transpose_batch")
            self.LB.check(self.lib.ltrx_split_image(P(self.flat_t), P(self.flat_ti), self.flat_t.numel(), self._st()), "split_image(W^T)")
        self.LB.check(self.lib.ltrx_split_image(P(self.flat_p), P(self.flat_pi), self.nflat, self._st()), "split_image(W)")

    def _sync_weights(self):
        """The forward / input-gradient GEMMs read the pre-split images and transposed copies of the weights, which the step
        refreshes after its own optimizer update.  A weight change made from OUTSIDE (``model.load_state_dict(ckpt)``, a manual
        re-initialisation, an external optimizer on the aliased parameters) bumps the parameters' autograd version counters --
        kernel writes through raw pointers do not -- so a changed counter (or a step of the slate-resident FC path, which never
        touches the images) means: refresh before the next forward."""
        ver = [p._version for p in self._order]
        if ver != self._wver or self._images_stale:
            self._refresh_transposes()
            self._wver = [p._version for p in self._order]
            self._images_stale = False

    def _img(self, w):
        """address of the pre-split image of a weight view (inside flat_p) or of a transposed copy (inside flat_t); None otherwise"""
        if w is None or self.gemm == "hipblaslt" or not self.weight_images:
            return None
        a = w.data_ptr()
        pairs = ((self.flat_p, self.flat_pi), (self.flat_t, self.flat_ti))
        if self._x_pad:
            pairs += ((self.w0_pad, self.w0_pad_i),)
        for base, img in pairs:
            lo = base.data_ptr()
            if lo <= a < lo + 4 * base.numel():
                return ctypes.c_void_p(img.data_ptr() + (a - lo))
        return None

    def _bucket_done(self, k):
        """gradient bucket k is final: start its all-reduce(SUM) now, behind the rest of the backward (the collective runs
        on the process group's own stream; ``_full`` waits for all of them before the optimizer step).  While a step is being
        captured the collective is not issued but handed to ``_seg_break``: it ends the hipGraph segment recorded so far and
        is launched between that segment's replay and the next one's."""
        if self.sharded and self.comm_enabled:
            import torch.distributed as dist
            lo, hi = self._buckets[k]
            if hi > lo:
                def launch(lo=lo, hi=hi):
                    self._works.append(dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if self._seg_break is not None:
                    self._seg_break(launch)
                else:
                    launch()

    def _wait_buckets(self):
        def wait():
            for w_ in self._works:                               # bucketed gradient all-reduces launched during the backward
                w_.wait()
            self._works = []
        if self._seg_break is not None and self.sharded and self.comm_enabled:
            self._seg_break(wait)
        else:
            wait()

    def _relu_bits(self, st):
        """the one-bit ReLU mask buffer of a layer's feed-forward activation for THIS step's row count, or None where the form does
        not apply (ltrx_gemm_nt_relu_bits_bytes: the large-tile GEMM must run both launches) -- then the input-gradient GEMM reads
        the saved fp32 activation instead (act 2)"""
        buf = st.get("rbits")
        if buf is None or self.gemm in ("hipblaslt", "split_bf16_strict"):
            return None
        # (both launches of a mask -- FFN-1 forward and FFN-2 input gradient -- contract over d_model: one K to ask about)
        need = self.lib.ltrx_gemm_nt_relu_bits_bytes(self.rows, self.dff, self.d)
        return buf if 0 < need <= buf.numel() else None

    def _lin_fwd(self, x, w, b, out, act=0, p=0.0, seed=0, res=None, bits=None):
        """out = drop_p(act(x w^T + b)) [+ res]   (nn.Linear forward, act 1 = ReLU; dropout in the epilogue; ``res`` = the
        residual stream of the SublayerConnection this projection closes, transformer.py:98-106: added in the epilogue, so the
        sum is written once by the GEMM instead of being re-read and re-written by the LayerNorm that follows)"""
        if self.gemm == "hipblaslt":
            n = self.rows
            torch.addmm(b, x[:n], w.t(), out=out[:n])
            if act == 1:
                torch.relu_(out[:n])
            if p:
                self._drop_apply(out, out, p, seed)
            if res is not None:
                out[:n].add_(res[:n])
            return
        P = self.LB.ptr
        if bits is not None and act == 1:                         # ReLU + its one-bit mask for the backward (act 4)
            self.LB.check(self.lib.ltrx_gemm_nt(P(x), x.stride(0), P(w), w.stride(0), self._img(w), P(out), out.stride(0), self.rows,
                                                w.shape[0], x.shape[1], P(b), 4, P(bits), 0, float(p), seed, P(self.drop_step), self._prec, 0,
                                                self._st()), "gemm_nt(fwd, relu bits)")
            return
        self.LB.check(self.lib.ltrx_gemm_nt(P(x), x.stride(0), P(w), w.stride(0), self._img(w), P(out), out.stride(0), self.rows, w.shape[0],
                                            x.shape[1], P(b), 3 if res is not None else act, P(res), res.stride(0) if res is not None else 0,
                                            float(p), seed, P(self.drop_step), self._prec, 0, self._st()), "gemm_nt(fwd)")

    def _lin_dgrad(self, dy, w, wT, out, relu_of=None, p=0.0, seed=0, bits=None):
        """out = dy w   (input gradient of nn.Linear); wT = w^T contiguous.  With ``relu_of`` (the saved post-ReLU,
        post-dropout activation that produced the layer input) the ReLU(+dropout p) backward mask is applied in the GEMM
        epilogue; without it, p > 0 re-applies the dropout mask of site ``seed`` (identity activation)."""
        if self.gemm == "hipblaslt":
            torch.mm(dy[:self.rows, :w.shape[0]], w, out=out[:self.rows])      # (dy may carry zero padding columns)
            if relu_of is not None:
                self._relu_bwd(out, relu_of, p)
            elif p:
                self._drop_apply(out, out, p, seed)
            return
        P = self.LB.ptr
        if bits is not None and relu_of is not None:              # the mask written by the forward launch (act 5): 1/32 of the bytes
            self.LB.check(self.lib.ltrx_gemm_nt(P(dy), dy.stride(0), P(wT), wT.stride(0), self._img(wT), P(out), out.stride(0), self.rows,
                                                wT.shape[0], dy.shape[1], None, 5, P(bits), 0, float(p), seed, P(self.drop_step),
                                                self._prec, 0, self._st()), "gemm_nt(dgrad, relu bits)")
            return
        self.LB.check(self.lib.ltrx_gemm_nt(P(dy), dy.stride(0), P(wT), wT.stride(0), self._img(wT), P(out), out.stride(0), self.rows,
                                            wT.shape[0], dy.shape[1], None, 2 if relu_of is not None else 0, P(relu_of),
                                            relu_of.stride(0) if relu_of is not None else 0, float(p), seed, P(self.drop_step),
                                            self._prec, 0, self._st()), "gemm_nt(dgrad)")

    def _lin_wgrad(self, dy, x, gw, gb, defer=False):
        """gw = dy^T x, gb = column sums of dy   (weight and bias gradients of nn.Linear).  ``defer``: an encoder-layer projection
        -- queued for the layer's one grouped launch (_wgrad_flush); dy and x must stay untouched until then."""
        if self.gemm == "hipblaslt":
            torch.mm(dy[:self.rows].t(), x[:self.rows], out=gw)
            self._colsum(dy, gb)
            return
        if defer and self.group_wgrad:
            self._wg_pending.append((dy, x, gw, gb))
            return
        P = self.LB.ptr
        # (tile 9: x is the padded input buffer -- its rows are readable up to the 256-column tile, see include/ltrx.h)
        padded = self._x_pad and x.data_ptr() == self.x_in_buf.data_ptr() and x.stride(0) == self.x_in_buf.stride(0)
        self.LB.check(self.lib.ltrx_gemm_tn(P(dy), dy.stride(0), P(x), x.stride(0), P(gw), P(gb), self.rows, dy.shape[1],
                                            x.shape[1], self._prec, 9 if padded else 0, P(self.ws_tn), self._st()),
                      "gemm_tn(wgrad)")

    def _wgrad_flush(self, defer_reduce=False):
        """the queued weight gradients of a layer as ONE ltrx_gemm_tn_group launch; their partial slabs are summed by the call
        (one launch per projection) or, with ``defer_reduce``, by the layer's _reduce_flush (the workspace must stay untouched
        until then)"""
        q = self._wg_pending
        if not q:
            return
        import ctypes
        n = len(q)
        vp, ci = ctypes.c_void_p * n, ctypes.c_int * n
        A = vp(*[t[0].data_ptr() for t in q])
        Bm = vp(*[t[1].data_ptr() for t in q])
        C = vp(*[t[2].data_ptr() for t in q])
        bo = vp(*[t[3].data_ptr() for t in q])
        lda = ci(*[t[0].stride(0) for t in q])
        ldb = ci(*[t[1].stride(0) for t in q])
        NP = ci(*[t[0].shape[1] for t in q])
        KP = ci(*[t[1].shape[1] for t in q])
        if defer_reduce:
            so, bso, sp = vp(), vp(), ctypes.c_int(0)
            outs = (so, bso, ctypes.byref(sp))
        else:
            outs = (None, None, None)
        # did the grouped kernel take this composition, or did the call fall back to one launch per problem (shapes outside the
        # large-tile kernel / not enough workspace)?  Same host-side predicate the library applies; tests assert it
        took = (self.lib.ltrx_debug_tn_group_map(n, self.rows, NP, KP, self._wg_probe, self._wg_probe) > 0
                and self.lib.ltrx_gemm_tn_group_workspace_bytes(n, self.rows, NP, KP) <= self.ws_tn.numel())
        self.wgrad_group_log.append((n, bool(took)))
        del self.wgrad_group_log[:-16]
        if self.probe_wgrad is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.LB.check(self.lib.ltrx_gemm_tn_group(n, A, lda, Bm, ldb, C, bo, self.rows, NP, KP, self._prec, self.LB.ptr(self.ws_tn),
                                                  self.ws_tn.numel(), *outs, self._st()), "gemm_tn_group(wgrad)")
        if self.probe_wgrad is not None:
            ev1.record()
            self.probe_wgrad.append((ev0, ev1, n, bool(took)))
        if defer_reduce and sp.value > 0:
            for i, t in enumerate(q):
                nw = t[0].shape[1] * t[1].shape[1]
                self._red_pending.append((so[i], sp.value, nw, nw, t[2].data_ptr()))
                if bso[i]:
                    self._red_pending.append((bso[i], sp.value, t[0].shape[1], t[0].shape[1], t[3].data_ptr()))
        q.clear()

    def _reduce_flush(self):
        """every pending fixed-order reduction of the layer (weight-gradient slabs, bias column sums, LayerNorm parameter-gradient
        partials) in ONE launch (ltrx_reduce_group)"""
        q = self._red_pending
        self._ln_slot = 0
        if not q:
            return
        import ctypes
        n = len(q)
        vp = ctypes.c_void_p * n
        self.LB.check(self.lib.ltrx_reduce_group(n, vp(*[t[0] for t in q]), (ctypes.c_int * n)(*[t[1] for t in q]),
                                                 (ctypes.c_size_t * n)(*[t[2] for t in q]), (ctypes.c_size_t * n)(*[t[3] for t in q]),
                                                 vp(*[t[4] for t in q]), self._st()), "reduce_group")
        q.clear()

    # ---- the step body (capturable) ----------------------------------------------------------------------------
    def _forward(self, train=True):
        """input buffers -> scores (self.scores_raw [B, L, n_out], self.scores [B, L]); ``train=False`` is model.eval(): every
        dropout rate is 0 (the saved activations are written all the same, nothing reads them).  Returns (feat, sc_rows)."""
        P = self.LB.ptr
        lib, M, d, B, L = self.lib, self.rows, self.d, self.B, self.L
        kpm = None if self.compact else self.mask                 # packed rows are all valid keys
        W = self.W
        fc = self.model.input_layer
        dp = (lambda p: p) if train else (lambda p: 0.0)          # dropout rate of a site in this pass
        h = self.x_in
        if self.in_norm is not None:                              # FCModel.input_norm (model.py:39)
            self.LB.check(lib.ltrx_layernorm_torch_fwd(P(h), P(W(self.in_norm.weight)), P(W(self.in_norm.bias)), M, self.fc_sizes[0],
                                                       float(self.in_norm.eps), P(self.x_norm), P(self.mean_in), P(self.rstd_in),
                                                       self._st()), "layernorm_torch_fwd")
            h = self.x_norm
        for i, lyr in enumerate(fc.layers):
            w_i = W(lyr.weight)
            if i == 0 and self._x_pad:                             # F rounded up to the GEMM's K step: padded rows x padded W_0
                h, w_i = self.x_in_k, self.w0_pad
            if self.fc_act >= 3:                                   # Sigmoid / Tanh: GEMM + bias, then the activation in place
                self._lin_fwd(h, w_i, W(lyr.bias), self.fc_out[i])
                self.LB.check(lib.ltrx_out_act_fwd(P(self.fc_out[i]), M * self.fc_out[i].shape[1], self.fc_act - 2, P(self.fc_out[i]),
                                                   self._st()), "fc_act_fwd")
            else:
                self._lin_fwd(h, w_i, W(lyr.bias), self.fc_out[i], self.fc_act, dp(self.p_fc), self._site(1000 + i))
            h = self.fc_out[i]
        if self.pos is not None:                                  # transformer.py:51-52: x = sqrt(d) x + pe[rank]
            self.LB.check(lib.ltrx_posenc_fwd(P(h), P(self._pos_table()), P(self.idx_rows), P(kpm), M, d, self.pos_pad, float(d) ** 0.5,
                                              P(self.x_pe), self._st()), "posenc_fwd")
            h = self.x_pe
        x = h                                                     # residual stream
        for i, st in enumerate(self.layers):
            lay = st["mod"]
            n0, n1 = lay.sublayer[0].norm, lay.sublayer[1].norm
            self._ln_fwd(x, None, W(n0.a_2), W(n0.b_2), None, st["xn0"], st["mean0"], st["rstd0"])
            st["xin"] = x
            self._lin_fwd(st["xn0"], st["wqkv"], st["bqkv"], st["qkv"])
            qkv = st["qkv"]
            self.LB.check(lib.ltrx_mha_fwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(kpm), B, L, self.h,
                                           d // self.h, 3 * d, P(st["o"]), d, P(st["lse"]), dp(st["p_att"]), st["s_att"],
                                           P(self.drop_step), P(self.cu), P(self.order), self._mha_mode, self._st()), "mha_fwd")
            lo = lay.self_attn.linears[3]
            # x1 = x + dropout(attention branch): the residual sum is the out-projection's epilogue (act 3)
            self._lin_fwd(st["o"], W(lo.weight), W(lo.bias), st["x1"], 0, dp(st["p_s0"]), st["s_s0"], res=x)
            self._ln_fwd(st["x1"], None, W(n1.a_2), W(n1.b_2), None, st["xn1"], st["mean1"], st["rstd1"])
            ff = lay.feed_forward
            if self.probe is not None and train:                  # bench.py: HIP events around the roofline kernel, in the step
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            self._lin_fwd(st["xn1"], W(ff.w_1.weight), W(ff.w_1.bias), st["r"], 1, dp(st["p_ff"]), st["s_ff"],
                          bits=self._relu_bits(st) if train else None)
            if self.probe is not None and train:
                ev1.record()
                self.probe.append((ev0, ev1))
            # x(next layer) = x1 + dropout(feed-forward branch), again in the epilogue of the projection that closes the sublayer
            nxt = self.layers[i + 1]["xsum0"] if i + 1 < len(self.layers) else self.xsum_f
            self._lin_fwd(st["r"], W(ff.w_2.weight), W(ff.w_2.bias), nxt, 0, dp(st["p_s1"]), st["s_s1"], res=st["x1"])
            x = nxt
        out = self.model.output_layer
        if self.N:
            nf = self.enc.norm
            self._ln_fwd(x, None, W(nf.a_2), W(nf.b_2), None, self.xf, self.mean_f, self.rstd_f)
            feat = self.xf
        else:
            feat = x
        no = self.n_out
        sc_rows = self.scores_c if self.compact else self.scores_raw
        if no == 1:
            self.LB.check(lib.ltrx_score_head_fwd(P(feat), P(W(out.w_1.weight)), P(W(out.w_1.bias)), M, d, P(sc_rows), self._st()),
                          "score_head_fwd")
        else:                                                     # Linear(d, d_output) as a GEMM (model.py:117)
            self._lin_fwd(feat, W(out.w_1.weight), W(out.w_1.bias), sc_rows.view(-1, no))
        if self.out_act:                                          # OutputLayer activation (model.py:117), in place
            self.LB.check(lib.ltrx_out_act_fwd(P(sc_rows), M * no, self.out_act, P(sc_rows), self._st()), "out_act_fwd")
        if self.compact:                                          # packed scores -> the padded [B, L] grid of the loss kernels
            self.scores_raw.zero_()
            self.LB.check(lib.ltrx_scatter_rows(P(self.scores_c), no, P(self.idx), self.n_valid, no, P(self.scores_raw), no, self._st()),
                          "scatter_rows")
        if no > 1:
            torch.sum(self.scores_raw, dim=-1, out=self.scores)   # model.score (model.py:127)
        return feat, sc_rows

    def _body(self):
        P = self.LB.ptr
        lib, M, d, B, L = self.lib, self.rows, self.d, self.B, self.L
        W, G = self.W, self.G
        kpm = None if self.compact else self.mask
        fc = self.model.input_layer
        out = self.model.output_layer
        no = self.n_out
        self._wg_pending.clear()                                  # (a step that raised half-way must not leave work queued)
        self._red_pending.clear()
        self._ln_slot = 0
        feat, sc_rows = self._forward(True)
        # ---------------- loss (value + d/dscores) ----------------
        loss, dsc = self.loss.run(self.scores_raw, self.y_in, self._divisor)
        # ---------------- backward ----------------
        ga, gb = self.d_a, self.d_b
        if self.compact:                                          # d loss / d scores of the packed rows (alignment rows: 0)
            self.LB.check(lib.ltrx_gather_rows(P(dsc), no, P(self.idx), self.n_valid, M, no, P(self.dsc_c), no, self._st()),
                          "gather_rows")
            dsc = self.dsc_c
        if self.out_act:                                          # d loss / d pre-activation = d loss / d score * act'(score)
            self.LB.check(lib.ltrx_out_act_bwd(P(dsc), P(sc_rows), M * no, self.out_act, P(dsc), self._st()), "out_act_bwd")
        if no == 1:
            self.LB.check(lib.ltrx_score_head_bwd(P(dsc), P(feat), P(W(out.w_1.weight)), M, d, P(ga), P(G(out.w_1.weight)),
                                                  P(G(out.w_1.bias)), P(self.ws_head), self._st()), "score_head_bwd")
        else:                                                     # the d_output-wide head: weight / bias / input gradients as GEMMs
            self.dz_pad[:M, :no].copy_(dsc.reshape(-1, no)[:M])
            self._lin_wgrad(self.dz_pad[:, :no], feat, G(out.w_1.weight), G(out.w_1.bias))
            self._lin_dgrad(self.dz_pad, W(out.w_1.weight), self.woutT_pad, ga)
        if self.N:
            nf = self.enc.norm
            self._ln_bwd(ga, self.xsum_f, W(nf.a_2), self.mean_f, self.rstd_f, None, gb, G(nf.a_2), G(nf.b_2))
            ds = gb                                               # d loss / d (x1_last + ffn_last)
            other = ga
            for i in range(self.N - 1, -1, -1):
                st = self.layers[i]
                lay = st["mod"]
                n0, n1 = lay.sublayer[0].norm, lay.sublayer[1].norm
                ff = lay.feed_forward
                d_r, dq = self.d_r, self.dqkv
                mid, outb = other, ds                              # where the two LayerNorm backwards of the layer write: ping-pong over (ga, gb)
                # FFN branch
                db = self._branch_grad(ds, st["p_s1"], st["s_s1"])
                # (the four weight gradients of the layer are queued and run as one grouped launch before the first kernel that
                #  overwrites one of their operands: the LN0 backward below, or the second use of the dropout buffer d_br)
                self._lin_wgrad(db, st["r"], G(ff.w_2.weight), G(ff.w_2.bias), defer=True)
                self._lin_dgrad(db, W(ff.w_2.weight), self._wT.get(id(ff.w_2.weight)), d_r, relu_of=st["r"], p=st["p_ff"],
                                bits=self._relu_bits(st))
                self._lin_wgrad(d_r, st["xn1"], G(ff.w_1.weight), G(ff.w_1.bias), defer=True)
                self._lin_dgrad(d_r, W(ff.w_1.weight), self._wT.get(id(ff.w_1.weight)), self.tmp_d)
                self._ln_bwd(self.tmp_d, st["x1"], W(n1.a_2), st["mean1"], st["rstd1"], ds, mid, G(n1.a_2), G(n1.b_2))
                ds = mid                                           # ds = d loss / d x1
                # attention branch
                lo = lay.self_attn.linears[3]
                if st["p_s0"] and st["p_s1"]:                      # d_br still holds the FFN branch's dY
                    self._wgrad_flush()
                db = self._branch_grad(ds, st["p_s0"], st["s_s0"])
                self._lin_wgrad(db, st["o"], G(lo.weight), G(lo.bias), defer=True)
                self._lin_dgrad(db, W(lo.weight), self._wT.get(id(lo.weight)), self.d_o)
                qkv = st["qkv"]
                self.LB.check(lib.ltrx_mha_bwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(kpm), P(st["o"]),
                                               P(self.d_o), P(st["lse"]), B, L, self.h, d // self.h, 3 * d, d, P(dq),
                                               dq.data_ptr() + 4 * d, dq.data_ptr() + 8 * d, 3 * d, st["p_att"], st["s_att"],
                                               P(self.drop_step), P(self.cu), P(self.order), self._mha_mode, P(self.ws_mha), self._st()),
                              "mha_bwd")
                if self.compact and M > self.n_valid:              # alignment rows belong to no slate: no gradient
                    dq[self.n_valid:M].zero_()
                self._lin_wgrad(dq, st["xn0"], st["gwqkv"], st["gbqkv"], defer=True)
                self._lin_dgrad(dq, st["wqkv"], st.get("wqkvT"), self.tmp_d)
                self._wgrad_flush(defer_reduce=True)
                self._ln_bwd(self.tmp_d, st["xin"], W(n0.a_2), st["mean0"], st["rstd0"], ds, outb, G(n0.a_2), G(n0.b_2))
                ds, other = outb, mid                              # ds = d loss / d (layer input)
                self._reduce_flush()                               # the layer's parameter gradients are final from here
                self._bucket_done(self.N - 1 - i)
        else:
            ds, other = ga, gb
        if self.pos is not None:                                  # backward of x = sqrt(d) fc_out + pe[rank]
            if self.pos_learned:
                self.LB.check(lib.ltrx_posenc_table_bwd(P(ds), P(self.idx_rows), P(kpm), M, d, self.pos_pad, P(G(self.pos.pe.weight)),
                                                        self._st()), "posenc_table_bwd")
            self.LB.check(lib.ltrx_scale_inplace(P(ds), M * d, float(d) ** 0.5, self._st()), "scale_inplace")
        # FC stack
        for i in range(self.nfc - 1, -1, -1):
            lyr = fc.layers[i]
            if i == self.nfc - 1:                                # the last FC activation(+dropout) feeds the encoder / head
                if self.fc_act >= 3:
                    self.LB.check(lib.ltrx_out_act_bwd(P(ds), P(self.fc_out[i]), M * ds.shape[1], self.fc_act - 2, P(ds), self._st()),
                                  "fc_act_bwd")
                elif self.fc_act == 1:
                    self._relu_bwd(ds, self.fc_out[i], self.p_fc)
                elif self.p_fc:
                    self._drop_apply(ds, ds, self.p_fc, self._site(1000 + i))
            inp = (self.x_norm if self.in_norm is not None else self.x_in) if i == 0 else self.fc_out[i - 1]
            self._lin_wgrad(ds, inp, G(lyr.weight), G(lyr.bias))
            if i == 0 and self.in_norm is not None:
                # nn.LayerNorm parameter gradients: dw = sum dy * xhat, db = sum dy with dy = ds W_0 (the input itself needs no
                # gradient; ltrx_layernorm_bwd's da / db formulas only use the saved mean and rstd, its dx output is scratch)
                self._lin_dgrad(ds, W(lyr.weight), self._wT.get(id(lyr.weight)), self.d_in)
                self.LB.check(lib.ltrx_layernorm_bwd(P(self.d_in), P(self.x_in), P(W(self.in_norm.weight)), P(self.mean_in), P(self.rstd_in),
                                                     None, M, self.fc_sizes[0], float(self.in_norm.eps), P(self.d_in),
                                                     P(G(self.in_norm.weight)), P(G(self.in_norm.bias)), P(self.ws_ln_in), self._st()),
                              "layernorm_bwd(input_norm)")
            if i > 0:
                self._lin_dgrad(ds, W(lyr.weight), self._wT.get(id(lyr.weight)), self.fc_dgrad[i - 1],
                                relu_of=self.fc_out[i - 1] if self.fc_act == 1 else None, p=self.p_fc,
                                seed=self._site(1000 + i - 1))
                ds = self.fc_dgrad[i - 1]
                if self.fc_act >= 3:
                    self.LB.check(lib.ltrx_out_act_bwd(P(ds), P(self.fc_out[i - 1]), M * ds.shape[1], self.fc_act - 2, P(ds), self._st()),
                                  "fc_act_bwd")
        self._bucket_done(len(self._buckets) - 1)
        return loss

    def _pos_table(self):
        return self.W(self.pos.pe.weight) if self.pos_learned else self.pos.pe

    def _adam(self):
        P = self.LB.ptr
        if self.clip:
            self.LB.check(self.lib.ltrx_clip_grad_norm_scale(P(self.flat_g), self.nflat, self.clip, P(self.clip_scale),
                                                             P(self.grad_norm), P(self.ws_clip), self._st()), "clip_grad_norm")
        gsc = P(self.clip_scale) if self.clip else None
        if self.optimizer == "SGD":
            self.LB.check(self.lib.ltrx_sgd_step(P(self.flat_p), P(self.flat_g), P(self.flat_m) if self.momentum else None, self.nflat,
                                                 float(self.lr), self.momentum, 1 if self.nesterov else 0, self.weight_decay, 1.0, gsc,
                                                 self._st()), "sgd_step")
            return
        self.LB.check(self.lib.ltrx_adam_step(P(self.flat_p), P(self.flat_g), P(self.flat_m), P(self.flat_v), self.nflat,
                                              float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                              self.weight_decay, 1 if self.optimizer == "AdamW" else 0,
                                              P(self.step_count), 1.0, gsc, self._st()),
                      "adam_step")

    def _full(self):
        if self._any_dropout:
            self.LB.check(self.lib.ltrx_bump_u32(self.LB.ptr(self.drop_step), self._st()), "bump_u32")   # fresh masks every step
        loss = self._body()
        self._wait_buckets()
        self._adam()
        self._refresh_transposes()
        return loss

    def first_nonfinite(self):
        """(name of the first parameter tensor -- flat-buffer order -- whose gradient of the LAST step holds a NaN / Inf, number of
        non-finite gradient elements), or (None, 0): the finiteness check that stands in for torch.autograd.detect_anomaly() on a
        step without an autograd graph (main.py:89).  One launch over the flat gradient buffer, one host sync."""
        if getattr(self, "_nf_seg", None) is None:
            names = {id(p): n for n, p in self.model.named_parameters()}
            self._nf_names = [names.get(id(p), "?") for p in self._order]
            self._nf_seg = torch.tensor([self._pv[id(p)][0] for p in self._order], dtype=torch.int64, device=self.dev)
            self._nf_out = torch.zeros(2, dtype=torch.int32, device=self.dev)
        P = self.LB.ptr
        self.LB.check(self.lib.ltrx_first_nonfinite(P(self.flat_g), self.nflat, P(self._nf_seg), len(self._order), P(self._nf_out),
                                                    self._st()), "first_nonfinite")
        first, count = (int(v) for v in self._nf_out.cpu())
        return (self._nf_names[first], count) if count else (None, 0)

    def set_lr(self, lr):
        """change the learning rate (per-epoch scheduler step): the rate is a launch argument of the Adam kernel, so a
        captured graph is dropped and re-captured on the next step"""
        if float(lr) != float(self.lr):
            self.lr = float(lr)
            self._graphs.clear()
            self._graph_pool = None      # (a pool whose graphs are all gone cannot be captured into again: fresh handle)

    def _pack(self, xb, lengths):
        """compact mode: build idx / cu_seqlens for this batch and gather the valid rows of xb into x_in.  ``lengths`` (host
        ints, one per slate; valid items first, as dataset.py:28-38 pads) avoids the one host sync that counting the
        valid items on the device costs."""
        B, L = self.B, self.L
        if lengths is not None:
            # only the B+1 prefix sums cross PCIe, from a ring of pinned staging buffers (a slot is reused after its copy
            # has completed); the packed-row index is derived on the device
            k = self._pack_turn % len(self._cu_ring)
            self._pack_turn += 1
            host, ev = self._cu_ring[k]
            ev.synchronize()
            lens = torch.as_tensor(lengths, dtype=torch.int32)
            if lens.is_cuda:                                      # (a device tensor costs the sync the host lengths are meant to avoid)
                lens = lens.cpu()
            lens = lens.reshape(B).clamp(min=0, max=L)
            host[0] = 0
            torch.cumsum(lens, 0, dtype=torch.int32, out=host[1:B + 1])
            host[B + 1:] = torch.argsort(lens, descending=True, stable=True)
            n = int(host[B])
            self._cuord.copy_(host, non_blocking=True)
            ev.record()
            self.LB.check(self.lib.ltrx_packed_row_index(self.LB.ptr(self.cu), B, L, n, self.LB.ptr(self.idx), self._st()),
                          "packed_row_index")
        else:
            valid = (self.mask == 0)
            cnt = valid.sum(1)
            self.cu[1:] = torch.cumsum(cnt, 0)
            self.order.copy_(torch.argsort(cnt, descending=True, stable=True))
            idx = torch.nonzero(valid.reshape(-1)).reshape(-1)    # (host sync: the row count sizes every launch)
            n = int(idx.numel())
            self.idx[:n] = idx
        if n == 0 and not self.sharded:
            raise ValueError("FusedTrainer(compact=True): the batch has no valid item")
        # (sharded: a rank whose block of a short last batch is empty -- 1 slate on 2 ranks -- still runs the step on 32 all-zero
        #  alignment rows: zero loss, zero gradient, and every collective of the step is entered by every rank)
        self.n_valid = n
        # alignment rows (zero input, zero gradient) keep the row count a multiple of 32; never more rows than the buffers hold
        # (a tiny batch, B * L < 32: the old max(32, ...) ran 32 rows over M-row buffers -- ADVICE r4)
        self.rows = min(self.M, max(32, (n + 31) // 32 * 32))
        F = self.x_in.shape[1]
        self.LB.check(self.lib.ltrx_gather_rows(self.LB.ptr(xb), F, self.LB.ptr(self.idx), n, self.rows, F, self.LB.ptr(self.x_in),
                                                self.x_in.stride(0),
                                                self._st()), "gather_rows")

    def _reattach(self):
        """the module's parameters / gradients must alias the flat buffers (state_dict(), score() and external readers of
        .grad see what the kernels wrote): re-point anything an external zero_grad(set_to_none=True) / .to() detached."""
        for p in self._order:
            w, g = self._wv[id(p)], self._gv[id(p)]
            if p.data.data_ptr() != w.data_ptr():
                p.data = w
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def step(self, xb, yb, indices=None, global_batch=None, lengths=None):
        """copy the batch into the static input buffers and run (or replay) the step; returns the device loss [1]."""
        self._reattach()
        if self.fcstep:
            return self._fc_step(xb, yb, global_batch)
        self._sync_weights()
        self.y_cur = self.y_in
        if self.loss.name == "listMLE" and self.shuffle_ties:
            # listMLE.py:17: a fresh random column order per call breaks ties among equal labels at random; the
            # permutation lives in a persistent device buffer, so the refresh is safe under hipGraph replay
            self.loss.perm.copy_(torch.randperm(self.L, device=self.dev, generator=self._perm_gen))
        self._divisor = float(global_batch if global_batch is not None else self.B * self.world)
        direct = (yb.dtype == torch.float32 and yb.is_contiguous() and yb.is_cuda and yb.numel() == self.M
                  and (self.compact or (xb.dtype == torch.float32 and xb.is_contiguous() and xb.is_cuda
                                        and xb.numel() == self.x_in.numel())))
        if direct:                                                # x, y and the padding mask in one launch
            P = self.LB.ptr
            self.LB.check(self.lib.ltrx_ingest_batch(None if self.compact else P(xb), P(yb), 0 if self.compact else xb.numel(), self.M,
                                                     self.x_in.shape[1], self.x_in.stride(0), float(PADDED_Y_VALUE),
                                                     None if self.compact else P(self.x_in), P(self.y_in), P(self.mask), self._st()),
                          "ingest_batch")
        else:
            self.y_in.copy_(yb)
            self.mask.copy_(yb == PADDED_Y_VALUE)
        if self.compact:
            self._pack(xb.reshape(self.M, -1).contiguous(), lengths)
        elif not direct:
            self.x_in.copy_(xb.reshape(self.M, -1))
        if self.pos is not None:
            if indices is None:
                raise ValueError("FusedTrainer: the model has a positional encoding, step() needs `indices`")
            if self.compact:                                      # rank of every packed row; alignment rows -> padding row
                self.idx_rows.fill_(-1)
                self.idx_rows[:self.n_valid] = indices.reshape(-1)[self.idx[:self.n_valid].long()]
            else:
                self.idx_rows.copy_(indices.reshape(-1))
        if not self.use_graph:
            return self._eager()
        # The batch divisor is a by-value launch argument of the loss kernels (include/ltrx.h: `batch_divisor`), i.e. a captured
        # step keeps the divisor it was captured with.  Captured steps are therefore keyed by the divisor: the short last batch of
        # an epoch (DataLoader drop_last=False, dataset_loading.py:245; it arrives topped up with padded slates and global_batch =
        # its real slate count) gets its own capture.  The captures form an LRU of ``max_graphs`` (4) entries: a fifth distinct
        # divisor evicts the least recently used one (and says so once) instead of silently running eagerly.
        key = (self._divisor, bool(self.comm_enabled))
        segs = self._graphs.get(key)
        if segs is None:
            if self._warm < 2:
                self._warm += 1                       # warm-up outside capture (lazy module loads, kernel attributes, workspaces)
                return self._eager()
            if len(self._graphs) >= self.max_graphs:
                old_key = next(iter(self._graphs))
                del self._graphs[old_key]
                if not self._warned_evict:
                    import warnings
                    self._warned_evict = True
                    warnings.warn("allrank_amd: more than %d distinct batch divisors in flight -- evicting the least recently used captured "
                                  "step (divisor %g); every new divisor costs one re-capture" % (self.max_graphs, old_key[0]))
            try:
                segs = self._graphs[key] = self._capture()
            except RuntimeError as exc:
                # sharded only, and only for errors of the capture mechanism itself (a collective backend that cannot live next to a
                # stream capture): the eager step is the same arithmetic (tests/dist_equiv_worker.py), only with launch overhead.  Any
                # other error -- a failed LB.check, a shape error, out of memory -- is a bug in the captured path and is raised.
                msg = str(exc)
                if not self.sharded or not any(t in msg for t in ("capture", "Capture", "hipErrorStreamCapture", "cudaErrorStreamCapture")):
                    raise
                import warnings
                warnings.warn("allrank_amd: hipGraph capture of the sharded step failed (%r); running eagerly" % (exc,))
                self.capture_fallback = repr(exc)     # queryable: tests/dist_equiv_worker.py asserts it stays None
                self.use_graph = False
                self._graphs.clear()
                return self._eager()
        else:
            self._graphs.move_to_end(key)
        for g, after in segs:                         # (capture only records: the replay executes this step)
            g.replay()
            if after is not None:
                after()
        return self._graph_loss

    def ensure_captured(self, global_batch=None):
        """Capture the step for this batch divisor NOW -- recording only: nothing executes and no collective is issued -- and say
        whether that worked.  For callers that must agree ACROSS RANKS on captured-vs-eager before any rank runs a step in either form
        (bench.py): each rank calls this after its eager warm-up steps, the ranks all-reduce the answers, and a rank whose
        peers could not capture switches to eager with them (``use_graph = False``).  A failed capture is closed before returning
        (``_capture``) and its error kept in ``capture_fallback``.  (``step()`` itself also survives a capture failure of ONE rank:
        the eager step issues the same collectives in the same order as a captured one.)"""
        if not self.use_graph or self.fcstep:
            return True
        self._divisor = float(global_batch if global_batch is not None else self.B * self.world)
        key = (self._divisor, bool(self.comm_enabled))
        if key in self._graphs:
            return True
        try:
            self._graphs[key] = self._capture()
            return True
        except RuntimeError as exc:
            self.capture_fallback = repr(exc)
            self._graphs.pop(key, None)
            return False

    def _fc_step(self, xb, yb, global_batch):
        """the slate-resident step (ltrx_fc_listnet_step): reads the caller's batch in place -- x once -- and leaves scores in
        ``self.scores``, d loss / d scores in ``self.loss.grad`` (with ``keep_loss_grad``), gradients in the flat buffer, the loss in ``self.loss.loss``.  One
        GPU without clipping: the reducing launch also applies Adam (two launches per step); sharded or clipped: gradients only,
        then the all-reduce / clip and the flat-buffer Adam as in the general step."""
        LB, P = self.LB, self.LB.ptr
        # (the general path's static-buffer copy accepts host batches; so does this one: a host batch is copied to the device first)
        if not xb.is_cuda:
            xb = xb.to(self.dev, non_blocking=True)
        if not yb.is_cuda:
            yb = yb.to(self.dev, non_blocking=True)
        xb = xb.reshape(self.M, -1)
        if xb.dtype != torch.float32 or not xb.is_contiguous():
            xb = xb.float().contiguous()
        if xb.data_ptr() & 15:                                # (the kernel reads 16-byte pieces: a misaligned view is copied once)
            xb = xb.clone()
        yb = yb.reshape(self.B, self.L)
        if yb.dtype != torch.float32 or not yb.is_contiguous():
            yb = yb.float().contiguous()
        LB.require_device(xb, yb)
        self.y_cur = yb
        div = float(global_batch if global_batch is not None else self.B * self.world)
        fused_adam = not self.sharded and not self.clip
        hid = P(self.fc_out[0]) if self.keep_fc_out else None
        if fused_adam:
            opt = (P(self.flat_m), P(self.flat_v), P(self.step_count), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                   float(self.eps), self.weight_decay, 1 if self.optimizer == "AdamW" else 0)
        else:
            opt = (None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0)
        dsc = self._fc_b[1] if self.keep_loss_grad else None
        if self.fcstep == "collapse":
            LB.check(self.lib.ltrx_fc_linear_listnet_step(P(xb), P(yb), *self._fc_a, div, self._fc_b[0], dsc, P(self.loss.loss), P(self.flat_g),
                                                          *opt, P(self._fc_ws), self._st()), "fc_linear_listnet_step")
        else:
            LB.check(self.lib.ltrx_fc_listnet_step(P(xb), P(yb), *self._fc_a, div, self._fc_b[0], dsc, hid, P(self.loss.loss), P(self.flat_g),
                                                   *opt, P(self._fc_ws), self._st()), "fc_listnet_step")
        if not fused_adam:
            if self.sharded and self.comm_enabled:
                import torch.distributed as dist
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)
            self._adam()
        self._images_stale = True                             # (score()'s GEMM forward reads the weight images: refresh them first)
        return self.loss.loss

    def _eager(self):
        with sharding.shard_context(int(self._divisor), self.group) if self.sharded else _null():
            return self._full()

    @property
    def graph(self):
        """the captured full-batch step (None before capture)"""
        return self._graphs.get((float(self.B * self.world), bool(self.comm_enabled)))

    def _capture(self):
        """Record the step as a chain of hipGraph segments.  On one GPU that is a single graph.  Sharded (world > 1), every
        collective of the step -- the bucketed gradient all-reduces, the wait before Adam, and the one-float all-reduce of a
        batch-global loss normaliser (sharding.allreduce_sum_) -- ends the segment being recorded and is kept as the host
        action to run between two replays: the compute of a step is ~75 launches of 10-300 us at 64 slates per GPU (launch
        latency matters exactly where 8-GPU runs sit), while the collectives stay ordinary torch.distributed calls on the
        process group's stream, whatever the backend (RCCL on a node, gloo in the one-GPU tests)."""
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
            self._cap_stream = torch.cuda.Stream(device=self.dev)
        segs = []
        torch.cuda.synchronize(self.dev)
        self._cap_stream.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(self._cap_stream):
            # "thread_local": only this thread's calls are checked against the capture -- a process group's watchdog thread or a
            # data-loader thread touching the runtime must not invalidate it
            cur = [torch.cuda.CUDAGraph()]
            cur[0].capture_begin(pool=self._graph_pool, capture_error_mode="thread_local")

            def seg_break(after):
                cur[0].capture_end()
                segs.append((cur[0], after))
                cur[0] = torch.cuda.CUDAGraph()
                cur[0].capture_begin(pool=self._graph_pool, capture_error_mode="thread_local")

            self._seg_break = seg_break
            ok = False
            try:
                with sharding.shard_context(int(self._divisor), self.group, deferred=seg_break) if self.sharded else _null():
                    self._graph_loss = self._full()
                ok = True
            finally:
                self._seg_break = None
                try:
                    cur[0].capture_end()
                except Exception:                     # noqa: BLE001 -- the original error (if any) is the one to report
                    if ok:
                        raise
            segs.append((cur[0], None))
        torch.cuda.current_stream(self.dev).wait_stream(self._cap_stream)
        return segs


    def _fwd_only(self):
        self._forward(False)

    def score(self, xb, yb, indices=None, lengths=None):
        """``model.score(xb, yb == PADDED_Y_VALUE, indices)`` in eval mode (model.py:82-92) through the kernels of the training
        step -- the forward half only, every dropout off, replayed from its own hipGraph -- for the validation / metric passes of
        an epoch (train_utils.py:32-56, 101-107).  The nn.Module forward computes the same scores with fp32 library GEMMs at
        less than half the rate.  ``yb`` only provides the padding mask; the batch must have the trainer's [B, L] shape (top up
        a short last batch with all-padded slates).  Returns the trainer's score buffer [B, L] (valid until the next call)."""
        self._reattach()
        self._sync_weights()
        self.y_in.copy_(yb)
        self.y_cur = self.y_in
        self.mask.copy_(yb == PADDED_Y_VALUE)
        if self.compact:
            self._pack(xb.reshape(self.M, -1).contiguous(), lengths)
        else:
            self.x_in.copy_(xb.reshape(self.M, -1))
        if self.pos is not None:
            if indices is None:
                raise ValueError("FusedTrainer: the model has a positional encoding, score() needs `indices`")
            if self.compact:
                self.idx_rows.fill_(-1)
                self.idx_rows[:self.n_valid] = indices.reshape(-1)[self.idx[:self.n_valid].long()]
            else:
                self.idx_rows.copy_(indices.reshape(-1))
        if not self.use_graph:
            self._fwd_only()
            return self.scores
        if self.graph_fwd is None:
            if self._warm_fwd < 2:
                self._warm_fwd += 1
                self._fwd_only()
                return self.scores
            self.graph_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_fwd):
                self._fwd_only()
        self.graph_fwd.replay()
        return self.scores


class _null(object):
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------------------------------------------------------
# epoch loop on device-resident data  (the body of fit(), allrank/training/train_utils.py:78-147, without its host work)
# ------------------------------------------------------------------------------------------------------------------
def fit_device(model, loss_name, loss_args, train_ds, val_ds, epochs, batch_size, slate_length, metrics=None, lr=1e-3,
               val_metric=None, early_stopping_patience=None, generator=None, use_fused=True, log=None,
               gradient_clipping_norm=None, lr_schedule=None, compact=False):
    """Train ``model`` on a DeviceSlates dataset; returns {"epochs", "train_loss", "val_metrics", "history"}.

    Per epoch: shuffled batches produced on the device (DeviceSlates.batches), one training step each (FusedTrainer when
    the model family / dropout allow it, else the autograd Trainer), the running loss is accumulated ON THE DEVICE (one
    host sync per epoch instead of the reference's ``loss.item()`` per step, train_utils.py:29), then one no-grad
    metrics pass over the validation set (train_utils.py:101-107).  The reference's second full pass over the TRAIN set
    for train metrics (train_utils.py:99, in train() mode, i.e. dropout active) is replaced by metrics taken from the
    scores of the TRAINING forward itself (same mode, no extra model pass): ``train_<metric>_<k>`` in
    the history is the mean over the epoch's training batches, each evaluated with the weights it was scored with.
    ``gradient_clipping_norm``: train_utils.py:24-25; ``lr_schedule(epoch) -> lr`` plays the role of the per-epoch
    ``scheduler.step()`` (train_utils.py:117-118), e.g. ``lambda e: 1e-3 * 0.1 ** (e // 50)`` for StepLR(50, 0.1).
    ``compact=True`` runs the fused step over the valid items only (FusedTrainer(compact=True))."""
    from . import losses as E
    from . import metrics as EMx
    from .data import evaluate
    metrics = metrics or {"ndcg": [5]}
    trainer, fused = None, False
    if use_fused:
        try:
            trainer = FusedTrainer(model, loss_name, loss_args, batch_size, slate_length, lr=lr, use_graph=True,
                                   gradient_clipping_norm=gradient_clipping_norm, compact=compact)
            fused = True
        except NotImplementedError:
            trainer = None
    if trainer is None:
        lossfn = (lambda s, t: getattr(E, loss_name)(s, t, **(loss_args or {})))
        trainer = Trainer(model, lossfn, torch.optim.Adam(model.parameters(), lr=lr), gradient_clipping_norm)
    history, best, best_epoch = [], -1.0, 0
    if epochs <= 0:
        return dict(epochs=0, train_loss=float("nan"), val_metrics=evaluate(model, val_ds, metrics), history=[], fused=fused)
    for epoch in range(epochs):
        if lr_schedule is not None:
            new_lr = float(lr_schedule(epoch))
            if fused:
                trainer.set_lr(new_lr)
            else:
                for g_ in trainer.opt.param_groups:
                    g_["lr"] = new_lr
        model.train()
        tot = torch.zeros(1, device=train_ds.device)
        nb = 0
        tm = {name: None for name in metrics} if fused else {}
        for xb, yb, idx in train_ds.batches(batch_size, slate_length, shuffle=True, generator=generator, drop_last=False):
            real = xb.shape[0]
            if fused and real < batch_size:
                # the last batch of an epoch (DataLoader drop_last=False, dataset_loading.py:245): the fused step has static
                # shapes, so the batch is topped up with fully padded slates (label -1, features 0: no loss, no gradient) and
                # the loss is normalised by the REAL slate count, exactly what the reference computes on the short batch
                padn = batch_size - real
                xb = torch.cat([xb, xb.new_zeros((padn,) + tuple(xb.shape[1:]))])
                yb = torch.cat([yb, yb.new_full((padn, yb.shape[1]), float(PADDED_Y_VALUE))])
                idx = torch.cat([idx, idx.new_full((padn, idx.shape[1]), -1)])
                loss = trainer.step(xb, yb, idx, global_batch=real)
            else:
                loss = trainer.step(xb, yb, idx)
            tot += loss.detach().view(1) * real
            nb += real
            for name in tm:                                   # metrics of the training forward (scores of this very step)
                v = getattr(EMx, name)(trainer.scores[:real], trainer.y_cur[:real], ats=metrics[name]).sum(0)
                tm[name] = v if tm[name] is None else tm[name] + v
        train_loss = float(tot.item()) / max(nb, 1)
        val = evaluate(model, val_ds, metrics)
        rec = dict(epoch=epoch, train_loss=train_loss, **val)
        for name, v in tm.items():
            if v is not None:
                for at, x_ in zip(metrics[name], (v / max(nb, 1)).cpu().numpy()):
                    rec["train_%s_%d" % (name, at)] = float(x_)
        history.append(rec)
        if log:
            log(history[-1])
        if val_metric is not None and early_stopping_patience is not None:
            cur = val[val_metric]
            if cur > best:
                best, best_epoch = cur, epoch
            if epoch - best_epoch > early_stopping_patience:       # early_stop.py:17-19
                break
    return dict(epochs=epoch, train_loss=history[-1]["train_loss"], val_metrics={k: v for k, v in history[-1].items()
                                                                                 if k not in ("epoch", "train_loss")
                                                                                 and not k.startswith("train_")},
                history=history, fused=fused)
