"""Device-resident learning-to-rank dataset with on-device FixLength -- SURVEY.md §8(f) row 1.

The reference loader (allrank/data/dataset_loading.py) keeps one numpy array per query on the host, pads / samples each
slate in Python inside DataLoader workers (FixLength, :32-93) and ships every batch over PCIe from pageable memory
(train_utils.py:95).  That tops out orders of magnitude below what the training step consumes.  Here the whole
dataset lives in HBM in CSR form (WEB30K fold 1: 2.27 M items x 136 fp32 = 1.2 GB of the 288 GB):

    x_items [n_items, F] f32,  y_items [n_items] f32,  offsets [n_slates + 1] i64,  item_of [n_slates, max_len] i64 (-1 = none)

and a batch is produced on the device by two HIP kernels (``ltrx_fixlength_positions`` + ``ltrx_assemble_batch``,
allrank_amd/csrc/ltrx_data.hip; no host round trip, no per-slate Python; ``_positions_torch`` is the same transform in torch
device ops, kept as the independent implementation the tests compare the kernel with):
  * slates shorter than ``slate_length`` are padded: features 0, label -1, index -1 (FixLength._pad, :81-93);
  * longer slates are subsampled WITHOUT replacement in random order (FixLength._sample, :61-79) by drawing one
    uniform key per item and keeping the top ``slate_length`` keys;
  * the reference's relevance rule is kept: if the sample contains no relevant item but the slate has some, then --
    exactly one relevant document in the slate: it replaces the last sampled slot (:72-74); otherwise the slate is
    re-sampled until a relevant item is in (:75-76).
Output per batch: ``(xb f32[B, L, F], yb f32[B, L], indices i64[B, L])`` exactly like ToTensor (:19-29), on the device.
The libsvm text itself is parsed on the device too (``parse_svm_file_on_device`` -> ``ltrx_libsvm_parse``: the file's bytes are
uploaded once, one thread per line; scikit-learn's load_svmlight_file, the reference's parser (:130), stays available as
``from_svm_file(..., parser="sklearn")`` and is what the parity test compares with).

Round 6 -- behind the plugin boundary: ``load_libsvm_dataset`` / ``load_libsvm_dataset_role`` / ``create_data_loaders`` below have the
signatures of allrank/data/dataset_loading.py:197-248 and are what ``allrank_amd.install()`` binds to those names (and to
``allrank.main``'s copies, main.py:8), so an unmodified main.py trains from HBM: ``DeviceLibSVMDataset`` (a ``DeviceSlates`` plus the
slate length of its role) and ``DeviceLoader`` (batch order drawn from torch's global generator exactly as the reference's
``DataLoader(shuffle=True)`` draws it -- same seeds, same slates per batch; under a process group a rank assembles ONLY its
contiguous block of each global batch: no full-batch collation, no host-to-device copy at all).
"""
import logging

import numpy as np
import torch

log = logging.getLogger("allrank_amd.data")

PADDED_Y_VALUE = -1
PADDED_INDEX_VALUE = -1


def parse_svm_file_on_device(path, device="cuda", n_features=None):
    """(X f32[n, F], y f32[n], qid i64[n]) as device tensors, parsed on the GPU.  Feature indices follow scikit-learn's
    zero_based="auto": one-based unless the smallest index in the file is 0."""
    from . import _lib as LB
    lib = LB.lib()
    dev = torch.device(device)
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        raise ValueError("empty file: %s" % path)
    text = torch.from_numpy(raw).to(dev)
    nl = torch.nonzero(text == 10).flatten()
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), nl + 1])
    starts = starts[starts < text.numel()]
    # drop blank / comment-only lines like sklearn does
    first = text[starts]
    starts = starts[(first != 10) & (first != 13) & (first != 35)].contiguous()
    n = int(starts.numel())
    y = torch.empty(n, dtype=torch.float32, device=dev)
    qid = torch.empty(n, dtype=torch.int64, device=dev)
    mm = torch.tensor([2 ** 31 - 1, -1], dtype=torch.int32, device=dev)
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    st = LB.stream_of(text)
    LB.check(lib.ltrx_libsvm_parse(LB.ptr(text), LB.ptr(starts), n, int(text.numel()), LB.ptr(y), LB.ptr(qid), None, 0, 0, LB.ptr(mm),
                                   LB.ptr(bad), st), "libsvm_parse(scan)")
    lo, hi = (int(v) for v in mm.cpu())
    if int(bad.item()):
        raise ValueError("%d malformed line(s) in %s" % (int(bad.item()), path))
    base = 0 if lo == 0 else 1
    F = int(n_features) if n_features is not None else hi - base + 1
    X = torch.zeros((n, F), dtype=torch.float32, device=dev)
    LB.check(lib.ltrx_libsvm_parse(LB.ptr(text), LB.ptr(starts), n, int(text.numel()), None, None, LB.ptr(X), F, base, None, LB.ptr(bad),
                                   st), "libsvm_parse(fill)")
    return X, y, qid


class DeviceSlates(object):
    def __init__(self, X, y, query_ids, device="cuda"):
        """X: [n_items, F] array (dense or scipy sparse), y: [n_items], query_ids: [n_items]; items of one query must be
        contiguous (as in libsvm LTR files); queries keep their order of first appearance (dataset_loading.py:109-113)."""
        if hasattr(X, "toarray"):
            X = X.toarray()
        self.device = torch.device(device)
        if torch.is_tensor(X):                       # already on the device (parse_svm_file_on_device): only the query ids go to the host
            q = query_ids.cpu().numpy() if torch.is_tensor(query_ids) else np.asarray(query_ids)
            Xt, yt = X.to(self.device, torch.float32).contiguous(), y.to(self.device, torch.float32).contiguous()
        else:
            X = np.ascontiguousarray(X, dtype=np.float32)
            y = np.asarray(y, dtype=np.float32)
            q = np.asarray(query_ids)
            Xt, yt = torch.from_numpy(X).to(self.device), torch.from_numpy(y).to(self.device)
        change = np.flatnonzero(q[1:] != q[:-1]) + 1
        starts = np.concatenate([[0], change]).astype(np.int64)
        offsets = np.concatenate([starts, [len(q)]]).astype(np.int64)
        lens = np.diff(offsets)
        self.n_slates = int(len(lens))
        self.n_features = int(X.shape[1])
        self.longest_query_length = int(lens.max())
        self.lengths_host = torch.from_numpy(lens.astype(np.int32))
        self.x_items = Xt
        self.y_items = yt
        self.offsets = torch.from_numpy(offsets).to(self.device)
        self.lengths = torch.from_numpy(lens).to(self.device)
        pos = torch.arange(self.longest_query_length, device=self.device)[None, :]
        item = self.offsets[:-1, None] + pos
        self.item_of = torch.where(pos < self.lengths[:, None], item, torch.full_like(item, -1))
        # per-slate relevance totals for the sampling rule
        ypad = torch.where(self.item_of >= 0, self.y_items[self.item_of.clamp(min=0)], torch.zeros((), device=self.device))
        self.label_sum = ypad.sum(1)
        self.argmax_pos = ypad.argmax(1)

    @classmethod
    def from_svm_file(cls, path, device="cuda", parser="device"):
        """libsvm / SVMlight text -> DeviceSlates.  parser="device" (default on a GPU): the file's bytes are uploaded and parsed
        by ``ltrx_libsvm_parse`` (one thread per line); parser="sklearn": the reference's host parser (dataset_loading.py:130)."""
        if parser == "device" and torch.device(device).type == "cuda":
            X, y, qid = parse_svm_file_on_device(path, device)
            return cls(X, y, qid, device)
        from sklearn.datasets import load_svmlight_file
        X, y, qid = load_svmlight_file(path, query_id=True)
        return cls(X, y, qid, device)

    def __len__(self):
        return self.n_slates

    @property
    def shape(self):
        return [self.n_slates, self.longest_query_length, self.n_features]

    # ------------------------------------------------------------------------------------------------------------
    def _positions_torch(self, slates, L, generator):
        """[B, L] positions inside each slate (-1 = padding) following FixLength -- torch device ops (reference implementation
        of the kernel for the tests)."""
        lens = self.lengths[slates]
        B = slates.numel()
        maxlen = self.longest_query_length
        pos = torch.arange(L, device=self.device)[None, :].expand(B, L)
        out = torch.where(pos < lens[:, None], pos, torch.full_like(pos, -1))          # the padding branch (len < L)
        # reference: sample_size < dim -> pad, else sample (a slate of exactly L items is "sampled" = randomly permuted)
        long_rows = torch.nonzero(lens >= L, as_tuple=False).flatten()
        if long_rows.numel():
            todo = long_rows
            for _ in range(64):
                ls = lens[todo]
                keys = torch.rand((todo.numel(), maxlen), device=self.device, generator=generator)
                keys = torch.where(torch.arange(maxlen, device=self.device)[None, :] < ls[:, None], keys, torch.full_like(keys, -1.0))
                samp = keys.topk(L, dim=1).indices                                          # random subset, random order
                ysamp = self.y_items[self.item_of[slates[todo]].gather(1, samp)]
                none_rel = ysamp.sum(1) == 0
                tot = self.label_sum[slates[todo]]
                one = none_rel & (tot == 1)                                                  # dataset_loading.py:72-74
                samp[one, L - 1] = self.argmax_pos[slates[todo]][one]
                retry = none_rel & (tot != 1) & (tot > 0)                                    # :75-76
                out[todo[~retry]] = samp[~retry]
                todo = todo[retry]
                if todo.numel() == 0:
                    break
            if todo.numel():                    # astronomically unlikely: keep the last draw
                out[todo] = samp[retry]
        return out

    def positions(self, slates, L, seed):
        """[B, L] positions inside each slate (-1 = padding): FixLength on the device (ltrx_fixlength_positions)."""
        from . import _lib as LB
        LB.require_device(self.y_items, slates)
        slates = slates.to(torch.int64).contiguous()
        B = int(slates.numel())
        pos = torch.empty((B, int(L)), dtype=torch.int64, device=self.device)
        LB.check(LB.lib().ltrx_fixlength_positions(LB.ptr(self.offsets), LB.ptr(self.y_items), LB.ptr(slates), B, int(L),
                                                   self.longest_query_length, int(seed) & 0xFFFFFFFFFFFFFFFF, LB.ptr(pos),
                                                   LB.stream_of(self.y_items)), "fixlength_positions")
        return pos

    def batch(self, slates, slate_length, generator=None, seed=None):
        """slates: i64[B] slate ids -> (xb [B,L,F], yb [B,L], indices [B,L]) on the device.  ``seed`` keys the sampling of the
        slates longer than ``slate_length`` (default: drawn from ``generator`` -- one host sync; ``batches`` draws one seed per
        epoch and counts batches instead)."""
        from . import _lib as LB
        LB.require_device(self.x_items)
        L = int(slate_length)
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,), device=self.device, generator=generator).item())
        slates = slates.to(device=self.device, dtype=torch.int64).contiguous()
        B = int(slates.numel())
        if B == 0:                                   # (an empty block of a short last batch under a process group)
            return (torch.empty((0, L, self.n_features), dtype=torch.float32, device=self.device),
                    torch.empty((0, L), dtype=torch.float32, device=self.device), torch.empty((0, L), dtype=torch.int64, device=self.device))
        pos = self.positions(slates, L, seed)
        xb = torch.empty((B, L, self.n_features), dtype=torch.float32, device=self.device)
        yb = torch.empty((B, L), dtype=torch.float32, device=self.device)
        idx = torch.empty((B, L), dtype=torch.int64, device=self.device)
        LB.check(LB.lib().ltrx_assemble_batch(LB.ptr(self.x_items), LB.ptr(self.y_items), LB.ptr(self.offsets), LB.ptr(slates), LB.ptr(pos),
                                              B, L, self.n_features, LB.ptr(xb), LB.ptr(yb), LB.ptr(idx), LB.stream_of(xb)),
                 "assemble_batch")
        return xb, yb, idx

    def batch_torch(self, slates, slate_length, generator=None):
        """the same batch through torch device ops (tests)"""
        L = int(slate_length)
        pos = self._positions_torch(slates, L, generator)
        valid = pos >= 0
        item = self.item_of[slates].gather(1, pos.clamp(min=0))
        item = torch.where(valid, item, torch.zeros_like(item))
        xb = self.x_items[item] * valid[:, :, None]
        yb = torch.where(valid, self.y_items[item], torch.full((), float(PADDED_Y_VALUE), device=self.device))
        idx = torch.where(valid, pos, torch.full_like(pos, PADDED_INDEX_VALUE))
        return xb, yb, idx

    def batches(self, batch_size, slate_length=None, shuffle=False, generator=None, drop_last=False):
        """epoch iterator (DataLoader(batch_size, shuffle) semantics, dataset_loading.py:245-246); ``slate_length=None``
        pads to the longest slate like the validation transform (:185-194)."""
        L = self.longest_query_length if slate_length is None else int(slate_length)
        order = (torch.randperm(self.n_slates, device=self.device, generator=generator) if shuffle
                 else torch.arange(self.n_slates, device=self.device))
        # one seed per epoch (one host sync), a counter per batch: sampling of long slates never syncs inside the epoch
        base = int(torch.randint(0, 2 ** 40, (1,), device=self.device, generator=generator).item()) if L < self.longest_query_length else 0
        for k, s in enumerate(range(0, self.n_slates, batch_size)):
            ids = order[s:s + batch_size]
            if drop_last and ids.numel() < batch_size:
                break
            yield self.batch(ids, L, seed=(base << 22) + k)


def evaluate(model, dataset, metrics, batch_size=512, slate_length=None):
    """epoch-level metrics (compute_metrics / metric_on_epoch, train_utils.py:32-56) in one no-grad pass over a
    DeviceSlates dataset: {"ndcg_5": value, ...}; ``metrics`` = {"ndcg": [5, 10, ...]} like config.metrics."""
    from . import metrics as EM
    out = {}
    was_training = model.training
    model.eval()
    with torch.no_grad():
        acc = {name: [] for name in metrics}
        for xb, yb, idx in dataset.batches(batch_size, slate_length):
            sc = model.score(xb, yb == PADDED_Y_VALUE, idx)
            for name, ats in metrics.items():
                acc[name].append(getattr(EM, name)(sc, yb, ats=ats))
        for name, ats in metrics.items():
            vals = torch.cat(acc[name]).mean(0).cpu().numpy()
            for at, v in zip(ats, vals):
                out["%s_%d" % (name, at)] = float(v)
    model.train(was_training)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The reference's loader interface on top of DeviceSlates (allrank/data/dataset_loading.py:96-248) -- what install() binds
# ---------------------------------------------------------------------------------------------------------------------
class ShardBatch(tuple):
    """``(xb, yb, indices)`` as a DeviceLoader yields it: a plain 3-tuple for every consumer that unpacks it (the reference's own
    ``fit`` included), plus what a sharded consumer needs to know about the GLOBAL batch it is a block of:
      ``global_slates``  slates in the whole global batch (the loss divisor of SURVEY 8e),
      ``offset``         first row of this block inside the global batch,
      ``order_tag``      a checksum of the global batch's slate ids (host integer; every rank must hold the same one),
      ``lengths``        valid items per slate of THIS block as a host int32 tensor (variable-length execution sizes its launches
                         from it without a device round trip)."""

    def __new__(cls, tensors, global_slates, offset, order_tag, lengths=None):
        self = super(ShardBatch, cls).__new__(cls, tensors)
        self.global_slates, self.offset, self.order_tag, self.lengths = int(global_slates), int(offset), int(order_tag), lengths
        return self


class _SlateIds(torch.utils.data.Dataset):
    """dataset of the numbers 0 .. n-1: a torch DataLoader over it yields the slate ids of every batch, i.e. the sampler /
    batch-sampler behaviour of the reference's loader (dataset_loading.py:245-246) without touching any data"""

    def __init__(self, n):
        self.n = int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return int(i)


class DeviceLibSVMDataset(object):
    """What ``load_libsvm_dataset`` returns instead of a ``LibSVMDataset`` (dataset_loading.py:96-165): the slates of one role in
    HBM (``slates``: DeviceSlates) plus the length its batches are fixed to -- ``slate_length`` for the training role, the longest
    slate for every other role (:212-227).  ``shape`` / ``longest_query_length`` / ``len()`` / indexing keep the reference's meaning
    (main.py:63-64 reads ``shape[-1]``); an item is the transformed slate ``(x [L, F], y [L], indices [L])`` -- on the device."""

    def __init__(self, slates, slate_length=None):
        self.slates = slates
        self.slate_length = int(slates.longest_query_length if slate_length is None else slate_length)

    @classmethod
    def from_svm_file(cls, svm_file_path, slate_length=None, device="cuda", parser="device"):
        return cls(DeviceSlates.from_svm_file(svm_file_path, device=device, parser=parser), slate_length)

    def __len__(self):
        return self.slates.n_slates

    @property
    def longest_query_length(self):
        return self.slates.longest_query_length

    @property
    def shape(self):
        return self.slates.shape

    @property
    def samples(self):
        """True if FixLength's sampling branch can occur (some slate has at least ``slate_length`` items, :55-58)"""
        return self.slates.longest_query_length >= self.slate_length

    def __getitem__(self, idx):
        xb, yb, ib = self.slates.batch(torch.tensor([int(idx)], device=self.slates.device), self.slate_length,
                                       seed=int(np.random.randint(0, 2 ** 31 - 1)) if self.samples else 0)
        return xb[0], yb[0], ib[0]


class DeviceLoader(object):
    """The iteration contract of the reference's ``DataLoader(ds, batch_size=total, shuffle=..., drop_last=False)``
    (dataset_loading.py:245-246) over a DeviceLibSVMDataset: yields ``(xb f32[B, L, F], yb f32[B, L], indices i64[B, L])`` --
    already on the device, assembled there by ``ltrx_fixlength_positions`` + ``ltrx_assemble_batch``.

    * Batch composition: the slate ids of every batch come from a torch ``DataLoader`` over the ids 0..n-1 with the same
      ``batch_size`` / ``shuffle``, so the draws from torch's GLOBAL generator (worker base seed, RandomSampler seed) and the
      resulting permutation are the reference loader's: under main.py:36-38's seeds epoch e visits the same slates in the same
      batches in the same order.  ``burn()`` consumes the draws of one iteration without running it (``allrank_amd.fit`` skips the
      reference's extra passes over the loaders, train_utils.py:99,107, and keeps the generator in step this way).
    * FixLength: the padding branch is exact.  The sampling branch (slates of >= L items, :61-79) is keyed by one seed per
      iteration -- drawn from numpy's global generator, the generator the reference samples from (:70) -- and the slate id.
    * ``world`` > 1: this rank yields its contiguous block (``parallel.shard_slates``, the rule of DataParallel.scatter) of each
      global batch as a ``ShardBatch``; nothing of the other ranks' blocks is assembled.  ``batch_size`` stays the GLOBAL size."""

    def __init__(self, dataset, batch_size, shuffle=False, rank=0, world=1):
        from torch.utils.data import DataLoader
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.rank, self.world = int(rank), max(1, int(world))
        self.drop_last = False
        self.num_workers = 0
        self._ids = DataLoader(_SlateIds(len(dataset)), batch_size=self.batch_size, shuffle=self.shuffle, num_workers=0)
        self.sampler = self._ids.sampler

    slate_length = property(lambda self: self.dataset.slate_length)

    def __len__(self):
        return len(self._ids)

    def burn(self):
        """consume what ONE iteration over the reference's loader draws from torch's global generator, without assembling a batch"""
        next(iter(self._ids), None)

    def batch_shape(self):
        """(global slates per batch, slate length, fraction of valid slots over the whole set) -- no batch is consumed"""
        s, L = self.dataset.slates, self.dataset.slate_length
        valid = float(s.lengths.clamp(max=L).double().sum().item()) / max(1.0, float(s.n_slates) * L)
        return self.batch_size, L, valid

    def __iter__(self):
        from .parallel import shard_slates
        ds, L = self.dataset, self.dataset.slate_length
        chunks = [c.to(torch.int64) for c in self._ids]                # host tensors; all of the epoch's generator draws happen here
        seed = int(np.random.randint(0, 2 ** 31 - 1)) if ds.samples else 0
        if not chunks:
            return
        order = torch.cat(chunks).to(ds.slates.device)                 # ONE small upload per epoch (8 B per slate)
        start = 0
        weights = torch.arange(1, self.batch_size + 1, dtype=torch.int64)
        for c in chunks:
            n = int(c.numel())
            a, b = shard_slates(n, self.rank, self.world)
            xb, yb, ib = ds.slates.batch(order[start + a:start + b], L, seed=seed)
            tag = int((c * weights[:n]).sum().item()) & 0x7FFFFFFFFFFF
            start += n
            yield ShardBatch((xb, yb, ib), n, a, tag, ds.slates.lengths_host[c[a:b]].clamp(max=L))


def _local_path(input_path, role):
    import os
    return os.path.join(input_path, "{}.txt".format(role))


def load_libsvm_role(input_path, role, device=None):
    """dataset_loading.py:168-182 (``{input_path}/{role}.txt`` -> dataset, no transform yet = padded to its longest slate): the
    file's bytes go to the GPU once and are parsed there."""
    from .launch import get_torch_device
    dev = torch.device(device) if device is not None else get_torch_device()
    path = _local_path(input_path, role)
    if str(input_path).startswith("gs://"):
        raise NotImplementedError("allrank_amd.data: %s is a GCS path -- copy the files to local storage (the device-resident loader "
                                  "reads local files; the reference's GCS helper, utils/file_utils.py, is out of scope)" % path)
    log.info("will load %s data from %s", role, path)
    ds = DeviceLibSVMDataset.from_svm_file(path, None, device=dev)
    log.info("%s DS shape: %s (resident on %s)", role, ds.shape, dev)
    return ds


def load_libsvm_dataset_role(role, input_path, slate_length, device=None):
    """dataset_loading.py:212-227: the training role is fixed to ``slate_length``, every other role to its longest slate"""
    ds = load_libsvm_role(input_path, role, device)
    if role == "train":
        ds.slate_length = int(slate_length)
    else:
        log.info("Will pad to the longest slate: %d", ds.longest_query_length)
        ds.slate_length = int(ds.longest_query_length)
    return ds


def load_libsvm_dataset(input_path, slate_length, validation_ds_role, device=None):
    """dataset_loading.py:197-209: (train, validation) datasets -- resident in HBM"""
    return (load_libsvm_dataset_role("train", input_path, slate_length, device),
            load_libsvm_dataset_role(validation_ds_role, input_path, slate_length, device))


def processing_units():
    """(rank, world) of the slate sharding: the process group's when one is up (or the launcher's record of it), else one unit"""
    import torch.distributed as dist
    from . import launch
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    if launch.world_size() > 1:
        return launch.rank(), launch.world_size()
    return 0, 1


def create_data_loaders(train_ds, val_ds, num_workers, batch_size):
    """dataset_loading.py:230-248: train loader shuffled, validation loader not, drop_last False, both with ``units x batch_size``
    slates per GLOBAL batch (":240-241: multiplying the batch size by the processing units count").  The processing units are the
    ranks of the process group when one is up (one process per GPU, ``allrank_amd.launch``), else the visible GPUs as in the
    reference.  DeviceLibSVMDataset -> DeviceLoader (``num_workers`` is meaningless there: no host work per batch); any other
    dataset -> the reference's torch DataLoader."""
    from torch.utils.data import DataLoader
    rank, world = processing_units()
    units = world if world > 1 else max(1, torch.cuda.device_count())
    total = units * int(batch_size)
    log.info("total batch size is %d (%d processing unit(s) x %d)", total, units, batch_size)
    out = []
    for ds, shuffle in ((train_ds, True), (val_ds, False)):
        if isinstance(ds, DeviceLibSVMDataset):
            out.append(DeviceLoader(ds, total, shuffle=shuffle, rank=rank, world=world))
        else:
            out.append(DataLoader(ds, batch_size=total, num_workers=num_workers, shuffle=shuffle))
    return out[0], out[1]
