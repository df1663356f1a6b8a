"""Slate-sharded data parallelism: the reduction algebra that makes per-rank losses add up to the reference's
loss on the gathered global batch (the reference runs the loss on GPU0 over the whole batch after
nn.DataParallel.gather, allrank/main.py:76-78, allrank/training/train_utils.py:20).

The losses consult this module for (a) the batch divisor -- the GLOBAL number of slates for the mean-type losses
-- and (b) an all-reduce(sum) of the batch-global normalisers that live on the device (neuralNDCG's count of
slates with idcg != 0, neuralNDCG.py:69; lambdaLoss(reduction="mean")'s pair count, lambdaLoss.py:77).
With no context active everything degenerates to the single-GPU case.  Gradients are summed (never averaged)
across ranks by allrank_amd.parallel -- each rank already divides by the global divisor.
"""
import contextlib

_state = {"global_batch": None, "group": None, "active": False, "deferred": None}


def active():
    return _state["active"]


def group():
    return _state["group"]


def batch_divisor(local_batch):
    gb = _state["global_batch"]
    return float(gb if (_state["active"] and gb is not None) else local_batch)


def allreduce_sum_(t):
    """in-place all-reduce(sum) of a small device tensor across the shard group (no-op when not sharded)."""
    if _state["active"]:
        import torch.distributed as dist
        grp = _state["group"]

        def launch():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp)
        if _state["deferred"] is not None:     # a step is being captured: the collective runs between two hipGraph segments
            _state["deferred"](launch)
        else:
            launch()
    return t


@contextlib.contextmanager
def shard_context(global_batch, group=None, deferred=None):
    """with shard_context(global_batch=G): loss = approxNDCGLoss(scores_local, y_local)  # -> this rank's share
    ``deferred(launch)``: instead of issuing a normaliser all-reduce, hand it to this callback (FusedTrainer._capture: the
    collective becomes the host action between two captured segments of the step)."""
    prev = dict(_state)
    _state.update(global_batch=int(global_batch), group=group, active=True, deferred=deferred)
    try:
        yield
    finally:
        _state.update(prev)
