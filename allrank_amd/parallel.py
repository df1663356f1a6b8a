"""Slate-sharded data parallelism over RCCL/xGMI: one process per GPU, one flat fp32 gradient buffer, one
all-reduce(SUM) per step.

Replaces the reference's single-process nn.DataParallel (allrank/main.py:76-78, model_utils.py:40-53), which
broadcasts all parameters from GPU0, gathers scores to GPU0 and evaluates the loss there every step.  Here each
rank scores and differentiates its own contiguous block of slates; the losses are already divided by the GLOBAL
batch (allrank_amd.sharding), so gradients are summed, never averaged -- that reproduces the reference's loss on
the gathered batch also for uneven last batches and for lambdaLoss(reduction="sum").
All parameter gradients are views into ONE contiguous buffer (set up once), so the exchange is a single
25.5 MB collective at config (3) instead of one per tensor.
"""
import torch
import torch.distributed as dist


class FlatGradients(object):
    """Re-points every parameter's .grad at a slice of one flat buffer; all_reduce() sums it across ranks."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self._offs = []
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            self._offs.append(off)
            off += p.numel()

    def check(self):
        """``.grad`` must still alias the flat buffer: an external ``optimizer.zero_grad()`` (set_to_none=True is torch's
        default), ``model.zero_grad()`` or ``model.to()`` silently detaches the views -- after that the collective and the
        optimizer would work on a stale buffer while autograd fills fresh tensors.  Detached views are re-pointed (their
        content, if any, is kept); callers should use ``zero()`` instead of zero_grad()."""
        base = self.flat.data_ptr()
        for p, off in zip(self.params, self._offs):
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                view = self.flat[off:off + p.numel()].view_as(p)
                if g is not None and g.device == view.device:
                    view.copy_(g)
                else:
                    view.zero_()
                p.grad = view

    def zero(self):
        self.flat.zero_()

    def all_reduce(self):
        self.check()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)


def shard_slates(n_slates, rank, world):
    """contiguous block of slates of rank `rank` (mirrors DataParallel.scatter on dim 0; the last ranks may get one
    slate fewer when n_slates % world != 0)."""
    base, rem = divmod(n_slates, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
