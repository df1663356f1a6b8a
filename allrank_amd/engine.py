"""Training-step driver for the hot path: the body of ``loss_batch`` (allrank/training/train_utils.py:18-29) without
its per-step host syncs, on device-resident data, optionally slate-sharded across ranks.

    mask = (yb == -1); loss = loss_func(model(xb, mask, indices), yb); loss.backward(); [clip]; opt.step(); opt.zero_grad()

Differences from the reference driver (all outside the arithmetic): no ``loss.item()`` per step (the loss stays on
the device; call ``.item()`` when you want it), gradients live in one flat buffer (allrank_amd.parallel), and under
``world_size > 1`` the loss is normalised by the global batch and gradients are summed over RCCL.
"""
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
        mask = (yb == PADDED_Y_VALUE)
        gb = global_batch if global_batch is not None else xb.shape[0] * self.world
        if self.world > 1:
            with sharding.shard_context(gb, self.group):
                loss = self.loss_func(self.model(xb, mask, indices), yb)
        else:
            loss = self.loss_func(self.model(xb, mask, indices), yb)
        loss.backward()
        if self.world > 1:
            self.flat.all_reduce()
        if self.clip:
            clip_grad_norm_(self.model.parameters(), self.clip)
        self.opt.step()
        self.flat.zero()          # == opt.zero_grad(set_to_none=False): grads stay views of the flat buffer
        return loss
