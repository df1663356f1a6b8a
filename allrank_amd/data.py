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
"""
import numpy as np
import torch

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
        slates = slates.to(torch.int64).contiguous()
        B = int(slates.numel())
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
