"""The epoch loop of allRank with the reference's own signature -- ``fit`` of allrank/training/train_utils.py:78-147 -- so that
``allrank_amd.install(fit=True)`` puts the explicit MI355X training step (engine.FusedTrainer) behind an UNMODIFIED
``allrank/main.py`` (main.py:90 calls ``fit(model=..., loss_func=..., optimizer=..., scheduler=..., train_dl=..., valid_dl=...,
config=..., device=..., output_dir=..., tensorboard_output_path=..., **asdict(config.training))``).

What is kept from the reference loop: the arguments and their meaning, per-epoch order (train pass, validation pass, scheduler
step incl. ReduceLROnPlateau on ``config.val_metric``, early stopping with EarlyStop's rule, early_stop.py:7-19), gradient clipping,
the ``model.pkl`` state_dict written to ``output_dir`` (loadable by the reference: same keys and shapes), tensorboard scalars when the
reference's writer is importable, and the returned dict {"epochs", "train_metrics", "val_metrics", "num_params"}.

What changes (all outside the arithmetic of a step):
  * a training step is ``FusedTrainer.step`` (hand-written HIP forward/backward/Adam, hipGraph replay) when the model family, the
    loss and the optimizer allow it -- an allrank_amd LTRModel, a loss from allrank_amd.losses bound with functools.partial (what
    main.py:83 builds once install() has rebound the names), torch.optim.Adam / AdamW / SGD (one parameter group, any betas / eps /
    weight decay / momentum; no amsgrad, maximize or dampening) -- and the
    autograd ``Trainer`` (same kernels for attention / LayerNorm / loss, torch autograd + the given optimizer) otherwise;
  * host batches are copied on a separate stream one batch ahead (``_Prefetcher``) instead of ``.to(device)`` on the compute
    stream in front of every step (train_utils.py:95);
  * no ``loss.item()`` per step: the running loss is accumulated on the device, one host sync per epoch (train_utils.py:29);
  * the validation pass runs through ``FusedTrainer.score`` (the forward half of the explicit step, dropout off, its own hipGraph)
    instead of the nn.Module forward;
  * train metrics come from the scores of the training forward itself instead of a second full pass over ``train_dl`` in train()
    mode (train_utils.py:99; same mode, same data).  ``train_metrics="reference"`` (or the environment variable
    ALLRANK_AMD_TRAIN_METRICS=reference -- main.py passes only config.training's keys) runs the reference's second pass instead: one
    pass over ``train_dl`` per metric name with the END-of-epoch weights, so ``experiment_result.json``'s ``train_metrics/*`` are the
    reference's numbers (dropout is off in that pass; the reference leaves it on, so with dropout > 0 its numbers carry that noise);
  * the loaders' draws from torch's global generator stay in step with the reference's loop: the reference iterates ``train_dl``
    1 + (number of metric names) times and ``valid_dl`` 1 + (number of metric names) times per epoch (train_utils.py:95-107, :36-43),
    each iteration drawing a worker base seed and -- shuffled -- a sampler seed; the passes skipped here are replaced by ``_burn``, so
    under main.py:36-38's seeds epoch e trains on the reference's batches in the reference's order;
  * batches that are already on the device pass straight through (``allrank_amd.data.DeviceLoader``: the training set lives in HBM,
    install() binds it behind main.py:57-68); under a process group such a loader yields only this rank's block of each global
    batch (``ShardBatch``) and nothing is sliced here;
  * ``config.detect_anomaly`` (main.py:89; the explicit step has no autograd graph for torch's anomaly mode to watch) or
    LTRX_CHECK_FINITE=1: every step's loss and flat gradient buffer are checked for non-finite values (one fused reduction, one
    host sync per step) and the first offending parameter tensor is named in the error;
  * the last, short batch of an epoch (DataLoader drop_last=False) is topped up with fully padded slates for the static-shape step
    and its loss is normalised by the real slate count;
  * under an initialised ``torch.distributed`` group every rank takes its contiguous block of each global batch -- training AND
    validation batches (``_evaluate``) -- (the reference wraps the model in nn.DataParallel instead, main.py:76-78; a DataParallel
    wrapper passed in is unwrapped).
"""
import functools
import logging
import os
import time

import numpy as np
import torch

from . import losses as E
from . import metrics as EM
from .data import ShardBatch
from .engine import FusedTrainer, Trainer, PADDED_Y_VALUE
from .parallel import shard_slates

log = logging.getLogger("allrank_amd.fit")

last_run = {}          # what the most recent fit() used: {"engine": "fused" | "autograd", "compact": bool, "reason": str}


class _EarlyStop(object):
    """allrank/training/early_stop.py:7-19"""

    def __init__(self, patience):
        self.patience, self.best_value, self.best_epoch = patience, 0.0, 0

    def step(self, current_value, current_epoch):
        if current_value > self.best_value:
            self.best_value, self.best_epoch = current_value, current_epoch

    def stop_training(self, current_epoch):
        return current_epoch - self.best_epoch > self.patience


def _tensorboard(path):
    try:
        from allrank.utils.tensorboard_utils import TensorboardSummaryWriter
        return TensorboardSummaryWriter(path)
    except Exception:
        return None


def _fused_spec(model, loss_func, optimizer):
    """(loss_name, loss_args, lr, optimizer kwargs of FusedTrainer) if the explicit step can run this job, else (None, reason)"""
    from .model import LTRModel
    if not isinstance(model, LTRModel):
        return None, "model is not an allrank_amd LTRModel"
    if not isinstance(loss_func, functools.partial) or getattr(E, getattr(loss_func.func, "__name__", ""), None) is not loss_func.func:
        return None, "loss is not a functools.partial of an allrank_amd loss"
    if loss_func.args:
        return None, "positional loss arguments"
    if len(optimizer.param_groups) != 1:
        return None, "optimizer has several parameter groups"
    g = optimizer.param_groups[0]
    if g.get("maximize", False) or g.get("amsgrad", False):
        return None, "maximize / amsgrad"
    if type(optimizer) in (torch.optim.Adam, torch.optim.AdamW):
        opt = dict(optimizer=type(optimizer).__name__, betas=tuple(g["betas"]), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]))
    elif type(optimizer) is torch.optim.SGD:
        if g.get("dampening", 0) != 0:
            return None, "SGD with dampening"
        opt = dict(optimizer="SGD", momentum=float(g["momentum"]), nesterov=bool(g["nesterov"]), weight_decay=float(g["weight_decay"]))
    else:
        return None, "optimizer %s is not one of the fused Adam / AdamW / SGD" % type(optimizer).__name__
    return (loss_func.func.__name__, dict(loss_func.keywords or {}), float(g["lr"]), opt), ""


def _pad_batch(xb, yb, idx, B):
    n = B - xb.shape[0]
    if n <= 0:
        return xb, yb, idx
    return (torch.cat([xb, xb.new_zeros((n,) + tuple(xb.shape[1:]))]),
            torch.cat([yb, yb.new_full((n, yb.shape[1]), float(PADDED_Y_VALUE))]),
            torch.cat([idx, idx.new_full((n, idx.shape[1]), -1)]))


class _Prefetcher(object):
    """Host batches -> device batches with one batch of look-ahead.  The reference copies every batch on the compute stream in
    front of the step (``xb.to(device)``, train_utils.py:95).  Here batch t+1 is copied on a SEPARATE stream into one of two
    persistent device sets while the GPU runs step t (as soon as the step that last read that set has finished), and the compute
    stream waits for the copy's event before step t+1.  Measured at config 3 (tools/fit_h2d_timing.py, 34 MB per batch): batch
    resident in HBM 9.9 ms per step, ``.to(device)`` per step 10.8 ms, this class 10.65-10.7 ms -- the host-to-device copy executes as
    a blit kernel that shares the CUs with the step, so most of its 0.8 ms stays; an extra hop through pinned staging buffers
    costs 5+ ms of host time per batch on this platform (15-17 ms per step) and is not used.  Keeping the training set in HBM
    (allrank_amd.data.DeviceSlates) is what removes the copy.  Batches that are already on the device pass through."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.stage = [None, None]          # per set: (device tensors, event "copy done")
        self.free = [None, None]           # per set: event on the compute stream "the step that read this device set has finished"
        self.turn = 0

    def _send(self, batch):
        if all((not torch.is_tensor(t)) or t.is_cuda for t in batch):
            return batch, None, None
        k = self.turn & 1
        self.turn += 1
        slot = self.stage[k]
        if slot is None or any(d.shape[1:] != t.shape[1:] or d.shape[0] < t.shape[0] or d.dtype != t.dtype for d, t in zip(slot[0], batch)):
            torch.cuda.current_stream(self.device).synchronize()      # (re)allocation: nothing of the old set may still be in use
            slot = self.stage[k] = ([torch.empty(t.shape, dtype=t.dtype, device=self.device) for t in batch], torch.cuda.Event())
            self.free[k] = None
        dev, ev = slot
        n = batch[0].shape[0]
        if self.free[k] is not None:
            self.stream.wait_event(self.free[k])
        with torch.cuda.stream(self.stream):
            for t, d in zip(batch, dev):
                d[:n].copy_(t, non_blocking=True)
            ev.record(self.stream)
        return [d[:n] for d in dev], ev, k

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._send(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev, k = nxt
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
            yield cur
            if k is not None:                             # the consumer has enqueued its work on the compute stream
                e = torch.cuda.Event()
                e.record(torch.cuda.current_stream(self.device))
                self.free[k] = e
            try:
                nxt = self._send(next(it))
            except StopIteration:
                nxt = None


def make_result(epoch, train_metrics, val_metrics, num_params):
    """the dict main.py:90 receives, with the VALUE TYPES the reference's fit returns: main.py:104 hands it to
    dump_experiment_result (utils/experiments.py:20-24), which calls ``.item()`` on every metric and on num_params -- numpy scalars in
    the reference (np.mean in train_utils.py:38-43, np.sum in model_utils.py:21-28), so numpy scalars here"""
    return {"epochs": epoch,
            "train_metrics": {k: np.float64(v) for k, v in train_metrics.items()},
            "val_metrics": {k: np.float64(v) for k, v in val_metrics.items()},
            "num_params": np.int64(num_params)}


def _batch_shape(dl):
    """(slates per batch, slate length) of a loader without consuming a batch: DataLoader.batch_size and the shape of one dataset
    item ([L, F] features, dataset_loading.py:19-29); loaders that do not expose them are asked for their first batch instead.
    Also returns the fraction of valid (non-padded) slots over a sample of the set (the auto choice of variable-length execution)."""
    if hasattr(dl, "batch_shape"):                  # allrank_amd.data.DeviceLoader knows all three without touching a batch
        return dl.batch_shape()
    try:
        bsz, ds = int(dl.batch_size), dl.dataset
        # The probe indexes dataset items; the reference's FixLength transform draws from the GLOBAL numpy generator when it
        # subsamples a long slate (dataset_loading.py:70), so the generator state is saved and restored around it -- a seeded run draws
        # the same subsamples with and without the probe.
        st = np.random.get_state()
        try:
            items = [ds[i] for i in range(0, len(ds), max(1, len(ds) // 32))][:32]
        finally:
            np.random.set_state(st)
        ys = torch.stack([torch.as_tensor(it[1]).float() for it in items])
        return bsz, int(items[0][0].shape[0]), float((ys != PADDED_Y_VALUE).float().mean())
    except (AttributeError, TypeError, IndexError, KeyError):          # loaders without batch_size / an indexable dataset
        first = next(iter(dl))
        return int(first[0].shape[0]), int(first[0].shape[1]), float((first[1] != PADDED_Y_VALUE).float().mean())


def _check_same_batch(xb, yb, world, batch=None):
    """sharded runs: every rank must see the SAME global batch for shard_slates to partition it (the reference's DataLoader
    shuffles per process; give the ranks one sampler seed) -- checked on the first batch with an all-gathered checksum.  A
    ``ShardBatch`` that is already this rank's block carries the checksum of the global batch's slate ids instead."""
    import torch.distributed as dist
    if batch is not None:
        chk = torch.tensor([float(batch.order_tag & 0xFFFFFF), float(batch.order_tag >> 24), float(batch.global_slates)], device=xb.device,
                           dtype=torch.float64)
    else:
        chk = torch.stack([xb.double().sum(), yb.double().sum(), torch.tensor(float(xb.shape[0]), device=xb.device, dtype=torch.float64)])
    allc = [torch.empty_like(chk) for _ in range(world)]
    dist.all_gather(allc, chk)
    if any(not torch.equal(c, allc[0]) for c in allc):
        raise RuntimeError("allrank_amd.fit: the ranks received different first batches -- under torch.distributed every rank must "
                           "iterate the same global batches (same DataLoader sampler seed / shuffle order); rank r trains on its "
                           "contiguous block of each of them")


def _my_block(batch, loader, rank, world):
    """(xb, yb, idx, slates in the global batch, the ShardBatch if the loader has already cut this rank's block) of one loader item"""
    xb, yb, idx = batch
    pre = isinstance(batch, ShardBatch) and world > 1 and getattr(loader, "world", 1) == world
    n_glob = batch.global_slates if isinstance(batch, ShardBatch) else int(xb.shape[0])
    if world > 1 and not pre:
        a, b = shard_slates(n_glob, rank, world)
        xb, yb, idx = xb[a:b], yb[a:b], idx[a:b]
    return xb, yb, idx, n_glob, (batch if pre else None)


def _host_lengths(batch, trainer, real, pre, world):
    """valid items per slate of this rank's block as HOST integers, topped up to the trainer's batch -- what variable-length
    execution needs to size its launches without a device round trip (FusedTrainer._pack).  A DeviceLoader knows them (the slate
    lengths of the resident set); any other loader: None (the step counts the valid items on the device, one host sync)."""
    lens = getattr(batch, "lengths", None)
    if lens is None or not trainer.compact or not (pre is not None or world == 1) or int(lens.numel()) != real:
        return None
    return torch.cat([lens, lens.new_zeros(trainer.B - real)]) if real < trainer.B else lens


def _burn(dl, times=1):
    """Consume what ``times`` iterations over ``dl`` draw from torch's global generator without loading a batch: the worker base
    seed every ``iter(DataLoader)`` draws and the seed a RandomSampler draws at its first index (torch/utils/data/dataloader.py,
    sampler.py) -- the passes of the reference's epoch loop that this loop does not make (module docstring)."""
    from torch.utils.data import DataLoader
    for _ in range(times):
        if hasattr(dl, "burn"):
            dl.burn()
        elif isinstance(dl, DataLoader) and dl.batch_sampler is not None:
            iter(DataLoader(dl.dataset, batch_sampler=dl.batch_sampler, num_workers=0, generator=dl.generator))   # the base seed (no
            #                                                                                   worker is started, no item is loaded)
            next(iter(dl.sampler), None)                                                     # the sampler's seed / permutation draw
        else:
            return False
    return True


def _evaluate(model, loss_func, dl, device, metrics, trainer=None, world=1, rank=0):
    """validation pass of train_utils.py:101-107: mean loss (weighted by batch size) and metric means, no autograd.  With a
    FusedTrainer the scores come from its forward-only pass (``FusedTrainer.score``: the kernels of the training step, dropout
    off, hipGraph replay) instead of the nn.Module forward (fp32 library GEMMs).
    Sharded (``world`` > 1, round 5): every rank scores its contiguous block of each validation batch -- the same partition as the
    training step -- under ``shard_context(global batch)``, so its loss is its share of the reference's loss on the whole batch
    (global divisors / normalisers); shares, slate counts and per-slate metric sums are all-reduced once at the end, and
    every rank returns the same numbers (scheduler and early stopping stay in step).  Before, every rank evaluated the whole set."""
    import torch.distributed as dist
    from . import sharding
    tot, num = torch.zeros((), device=device), 0
    acc = {name: None for name in metrics}
    pf = dl if isinstance(dl, _Prefetcher) else _Prefetcher(dl, device)
    with torch.no_grad():
        for batch in pf:
            xb, yb, idx, n_glob, pre = _my_block(batch, pf.loader, rank, world)
            n = int(xb.shape[0])
            if trainer is not None and n <= trainer.B and tuple(xb.shape[1:2]) == (trainer.L,):
                xs, ys, ids = _pad_batch(xb, yb, idx, trainer.B)
                sc = trainer.score(xs, ys, ids, lengths=_host_lengths(batch, trainer, n, pre, world))[:n]
                # (an empty shard of a short last batch still enters the loss -- its normaliser all-reduce is a collective -- with
                #  one fully padded slate: value 0, count 0)
                out, yl = trainer.scores_raw[:max(n, 1)], ys[:max(n, 1)]
            else:
                if n == 0:
                    xb, yb, idx = _pad_batch(xb, yb, idx, 1)
                mask = yb == PADDED_Y_VALUE
                out = model(xb, mask, idx)
                sc = (out if out.dim() == 2 else model.score(xb, mask, idx))[:n]
                yl = yb
            if loss_func is None:                                     # (metrics only: the reference's compute_metrics pass)
                share = tot.new_zeros(())
            elif world > 1:
                with sharding.shard_context(n_glob):
                    share = loss_func(out, yl).detach().float()
            else:
                share = loss_func(out, yl).detach().float()
            tot += share.reshape(()) * n_glob
            num += n
            if n > 0:
                for name, ats in metrics.items():
                    v = getattr(EM, name)(sc, yb[:n], ats=ats).sum(0)
                    acc[name] = v if acc[name] is None else acc[name] + v
    sums = [acc[name].float() if acc[name] is not None else torch.zeros(len(metrics[name]), device=device) for name in metrics]
    stats = torch.cat([tot.reshape(1), torch.tensor([float(num)], device=device)] + sums)
    if world > 1:
        dist.all_reduce(stats)
    stats = stats.cpu().numpy()
    n_all = max(float(stats[1]), 1.0)
    out, off = {}, 2
    for name, ats in metrics.items():
        for at in ats:
            out["%s_%d" % (name, at)] = np.float32(stats[off] / n_all)
            off += 1
    return float(stats[0]) / n_all, out


def _assert_finite(trainer, fused, loss, epoch, step, model):
    """the anomaly switch of the explicit step (main.py:89): raise on the first step whose loss or gradients are not finite and
    name the first offending parameter tensor.  The fused step has already applied the update when this runs (the check reads the
    gradient buffer the step left behind) -- the run stops here either way."""
    if fused:
        name, count = trainer.first_nonfinite()
    else:
        name, count = None, 0
        for n, p in model.named_parameters():                      # (autograd step: flat.zero() has run; check the weights it left)
            bad = int((~torch.isfinite(p.detach())).sum().item())
            if bad and name is None:
                name = n
            count += bad
    loss_ok = bool(torch.isfinite(loss.detach()).all().item())
    if name is not None or not loss_ok:
        raise FloatingPointError(
            "allrank_amd.fit (detect_anomaly / LTRX_CHECK_FINITE): step %d of epoch %d produced %s%s -- first offending parameter "
            "tensor: %s (%d non-finite %s element(s) in all).  Usual causes: non-finite input features, a learning rate that "
            "diverged, a slate with no valid item in a loss that divides by its normaliser."
            % (step, epoch, "a non-finite loss" if not loss_ok else "a finite loss", "" if name is None else " and non-finite values",
               name, count, "gradient" if fused else "weight"))


def fit(epochs, model, loss_func, optimizer, scheduler, train_dl, valid_dl, config, gradient_clipping_norm, early_stopping_patience,
        device, output_dir, tensorboard_output_path, use_fused=True, compact=None, gemm="split_bf16", train_metrics=None,
        check_finite=None):
    """Same positional / keyword arguments as the reference ``fit``; ``use_fused`` / ``compact`` / ``gemm`` / ``train_metrics`` /
    ``check_finite`` are extensions (compact=None: variable-length execution when less than 80 % of the first batch's slots are valid
    items; gemm: the arithmetic of the fused step, "split_bf16" = fp32-class parity arithmetic, "bf16" = the one-product throughput
    mode, see FusedTrainer; train_metrics: "fused" (default) = from the training forward, "reference" = the reference's second pass,
    train_utils.py:99; check_finite: None = ``config.detect_anomaly`` or LTRX_CHECK_FINITE=1, see the module docstring)."""
    import torch.distributed as dist
    device = torch.device(device)
    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if isinstance(model, torch.nn.DataParallel):
        # main.py:76-78 wrapped the model because several GPUs are visible.  The engine does not replicate inside one process: under a
        # process group every rank already holds its replica (allrank_amd.launch rebinds the wrapper away); without one, say so --
        # the whole (gpu_count x batch_size) batch of dataset_loading.py:240-241 is about to train on ONE device.
        n_vis = len(getattr(model, "device_ids", None) or [])
        model = model.module
        if world == 1:
            log.warning("allrank_amd.fit: the model arrived wrapped in nn.DataParallel over %d GPUs but no torch.distributed process "
                        "group is up -- training runs on %s ALONE, the other GPUs stay idle.  Start the job as "
                        "`python -m allrank_amd.launch --nproc %d -- <main.py arguments>` (one process per GPU, slate-sharded).",
                        n_vis, device, max(n_vis, 2))
    metrics = dict(config.metrics)
    val_metric = getattr(config, "val_metric", None)
    train_metrics_mode = (train_metrics or os.environ.get("ALLRANK_AMD_TRAIN_METRICS") or "fused").lower()
    if train_metrics_mode not in ("fused", "reference"):
        raise ValueError("train_metrics must be 'fused' or 'reference', got %r" % (train_metrics_mode,))
    if check_finite is None:
        check_finite = bool(getattr(config, "detect_anomaly", False)) or os.environ.get("LTRX_CHECK_FINITE", "0") not in ("", "0")
    writer = _tensorboard(tensorboard_output_path) if tensorboard_output_path else None
    num_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    early_stop = _EarlyStop(early_stopping_patience)
    rank = dist.get_rank() if world > 1 else 0

    spec, reason = _fused_spec(model, loss_func, optimizer) if use_fused else (None, "use_fused=False")
    trainer, fused = None, False
    B_glob, L, valid_frac = _batch_shape(train_dl)
    lo, hi = shard_slates(B_glob, rank, world)
    if spec is not None:
        if compact is None:
            compact = valid_frac < 0.8
        try:
            trainer = FusedTrainer(model, spec[0], spec[1], hi - lo, L, lr=spec[2], world_size=world, use_graph=True,
                                   gradient_clipping_norm=gradient_clipping_norm, compact=bool(compact), gemm=gemm, **spec[3])
            fused = True
        except (NotImplementedError, KeyError) as e:
            reason = "FusedTrainer: %s" % (e,)
    if trainer is None:
        trainer = Trainer(model, loss_func, optimizer, gradient_clipping_norm, world, None)
    last_run.clear()
    last_run.update(engine="fused" if fused else "autograd", compact=bool(trainer.compact) if fused else False, reason=reason,
                    fcstep=getattr(trainer, "fcstep", False))
    log.info("allrank_amd.fit: %s step%s", last_run["engine"], (" (" + reason + ")") if reason else "")

    epoch, train_metrics, val_metrics = -1, {}, {}
    train_pf, valid_pf = _Prefetcher(train_dl, device), _Prefetcher(valid_dl, device)     # one per loader: stream and staging sets persist
    checked = world == 1
    last_run["epoch_log"] = []            # per epoch: wall seconds of the training / validation pass, slates and slots trained on
    n_steps = 0
    for epoch in range(epochs):
        model.train()
        t_epoch = time.perf_counter()
        tot, num, slots = torch.zeros((), device=device), 0, 0
        tm = {name: None for name in metrics}
        for batch in train_pf:
            xb, yb, idx, real_glob, pre = _my_block(batch, train_dl, rank, world)
            if not checked:
                _check_same_batch(batch[0], batch[1], world, pre)
                checked = True
            real = int(xb.shape[0])
            if fused:
                xs, ys, ids = _pad_batch(xb, yb, idx, trainer.B)
                loss = trainer.step(xs, ys, ids, global_batch=real_glob, lengths=_host_lengths(batch, trainer, real, pre, world))
                scores, labels = trainer.scores[:real], trainer.y_cur[:real]
            else:
                if real == 0:
                    # (sharded: this rank's block of a short last batch is empty -- the step still has to run, its gradient all-reduce
                    #  is a collective: one fully padded slate, i.e. zero loss and zero gradients.  Found by the 2-rank autograd job of
                    #  tests/dist_fit_worker.py in round 5: the nn.Module forward raised on 0 rows while the peer sat in the all-reduce.)
                    xb, yb, idx = _pad_batch(xb, yb, idx, 1)
                loss = trainer.step(xb, yb, idx, global_batch=real_glob)
                scores, labels = trainer.last_scores[:real], yb[:real]
            n_steps += 1
            if check_finite:
                _assert_finite(trainer, fused, loss, epoch, n_steps, model)
            tot += loss.detach().float().reshape(()) * real_glob          # (sharded: this rank's share of the global-batch loss)
            num += real
            slots += real * int(yb.shape[1])
            if train_metrics_mode == "fused":
                for name, ats in metrics.items():
                    # (a rank whose shard of a short last batch is empty contributes zeros -- every rank keeps the same stats layout)
                    v = getattr(EM, name)(scores, labels, ats=ats).sum(0) if real > 0 else torch.zeros(len(ats), device=device)
                    tm[name] = v if tm[name] is None else tm[name] + v
        stats = torch.cat([tot.reshape(1), torch.tensor([float(num)], device=device)] + [tm[n].float() for n in metrics if tm[n] is not None])
        if world > 1:
            dist.all_reduce(stats)
        stats = stats.cpu().numpy()
        t_train = time.perf_counter() - t_epoch
        n_all = max(stats[1], 1.0)
        train_loss = float(stats[0]) / n_all
        train_metrics, off = {}, 2
        for name, ats in metrics.items():
            if tm[name] is None:
                continue
            for at in ats:
                train_metrics["%s_%d" % (name, at)] = float(stats[off]) / n_all
                off += 1
        if train_metrics_mode == "reference":
            # train_utils.py:99: compute_metrics(config.metrics, model, train_dl, device) -- one more pass over train_dl per metric
            # name (:46-54), with the end-of-epoch weights, the model still in train() mode
            train_metrics = {}
            for name, ats in metrics.items():
                _, one = _evaluate(model, None, train_pf, device, {name: ats}, trainer if fused else None, world, rank)
                train_metrics.update({k: float(v) for k, v in one.items()})
        else:
            _burn(train_dl, len(metrics))                          # the generator draws of the passes not made (module docstring)

        model.eval()
        t_val = time.perf_counter()
        val_loss, val_metrics = _evaluate(model, loss_func, valid_pf, device, metrics, trainer if fused else None, world, rank)
        _burn(valid_dl, len(metrics))                              # train_utils.py:107 iterates valid_dl once more per metric name
        last_run["epoch_log"].append({"train_s": t_train, "val_s": time.perf_counter() - t_val, "slates": int(stats[1]),
                                      "slots": int(slots)})

        lr_now = optimizer.param_groups[0]["lr"]
        if writer is not None:
            tb = {("train", "loss"): train_loss, ("val", "loss"): val_loss, ("train", "lr"): lr_now}
            tb.update({("train", k): v for k, v in train_metrics.items()})
            tb.update({("val", k): v for k, v in val_metrics.items()})
            writer.save_to_tensorboard(tb, epoch)
        log.info("Epoch : %d Train loss: %s Val loss: %s %s %s", epoch, train_loss, val_loss,
                 " ".join("Train %s %s" % kv for kv in train_metrics.items()), " ".join("Val %s %s" % kv for kv in val_metrics.items()))

        current = val_metrics.get(val_metric)
        if scheduler:                                              # train_utils.py:117-122
            if fused:
                optimizer._opt_called = True                       # (torch's "scheduler.step() before optimizer.step()" check: in
                #                                                     the fused run the optimizer object only carries the learning rate)
            if type(scheduler) is torch.optim.lr_scheduler.ReduceLROnPlateau:
                scheduler.step(val_metrics[val_metric])
            else:
                scheduler.step()
            if fused:                                              # the scheduler edits optimizer.param_groups; the fused Adam follows
                trainer.set_lr(float(optimizer.param_groups[0]["lr"]))
        early_stop.step(current, epoch)
        if early_stop.stop_training(epoch):
            log.info("early stopping at epoch %d since %s didn't improve from epoch no %d. Best value %s, current value %s",
                     epoch, val_metric, early_stop.best_epoch, early_stop.best_value, current)
            break

    if rank == 0:
        torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, os.path.join(output_dir, "model.pkl"))
    if writer is not None:
        writer.close_all_writers()
    return make_result(epoch, train_metrics, val_metrics, num_params)
