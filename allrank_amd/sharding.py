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
import threading

_DEFAULT = {"global_batch": None, "group": None, "active": False, "deferred": None}


class _Local(threading.local):
    """the shard context of THIS thread (like ops.arithmetic): two replica threads of one process -- the reference's
    nn.DataParallel layout, allrank/models/model_utils.py:40-53 -- must not see each other's divisor / group / deferral hook"""

    def __init__(self):
        self.state = dict(_DEFAULT)


_tls = _Local()


def _st():
    return _tls.state


def active():
    return _st()["active"]


def group():
    return _st()["group"]


def batch_divisor(local_batch):
    st = _st()
    gb = st["global_batch"]
    return float(gb if (st["active"] and gb is not None) else local_batch)


def allreduce_sum_(t):
    """in-place all-reduce(sum) of a small device tensor across the shard group (no-op when not sharded)."""
    st = _st()
    if st["active"]:
        import torch.distributed as dist
        grp = st["group"]

        def launch():
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp)
        if st["deferred"] is not None:         # a step is being captured: the collective runs between two hipGraph segments
            st["deferred"](launch)
        else:
            launch()
    return t


@contextlib.contextmanager
def shard_context(global_batch, group=None, deferred=None):
    """with shard_context(global_batch=G): loss = approxNDCGLoss(scores_local, y_local)  # -> this rank's share
    ``deferred(launch)``: instead of issuing a normaliser all-reduce, hand it to this callback (FusedTrainer._capture: the
    collective becomes the host action between two captured segments of the step)."""
    st = _st()
    prev = dict(st)
    st.update(global_batch=int(global_batch), group=group, active=True, deferred=deferred)
    try:
        yield
    finally:
        st.update(prev)
