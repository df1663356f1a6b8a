"""Device-resident learning-to-rank dataset with on-device FixLength -- SURVEY.md §8(f) row 1.

The reference loader (allrank/data/dataset_loading.py) keeps one numpy array per query on the host, pads / samples each
slate in Python inside DataLoader workers (FixLength, :32-93) and ships every batch over PCIe from pageable memory
(train_utils.py:95).  That tops out orders of magnitude below what the training step consumes.  Here the whole
dataset lives in HBM in CSR form (WEB30K fold 1: 2.27 M items x 136 fp32 = 1.2 GB of the 288 GB):

    x_items [n_items, F] f32,  y_items [n_items] f32,  offsets [n_slates + 1] i64,  item_of [n_slates, max_len] i64 (-1 = none)

and a batch is produced by torch device ops only (no host round trip, no per-slate Python):
  * slates shorter than ``slate_length`` are padded: features 0, label -1, index -1 (FixLength._pad, :81-93);
  * longer slates are subsampled WITHOUT replacement in random order (FixLength._sample, :61-79) by drawing one
    uniform key per item and keeping the top ``slate_length`` keys;
  * the reference's relevance rule is kept: if the sample contains no relevant item but the slate has some, then --
    exactly one relevant document in the slate: it replaces the last sampled slot (:72-74); otherwise the slate is
    re-sampled until a relevant item is in (:75-76).
Output per batch: ``(xb f32[B, L, F], yb f32[B, L], indices i64[B, L])`` exactly like ToTensor (:19-29), on the device.
Parsing libsvm text stays on the host (scikit-learn's load_svmlight_file, like the reference, :130).
"""
import numpy as np
import torch

PADDED_Y_VALUE = -1
PADDED_INDEX_VALUE = -1


class DeviceSlates(object):
    def __init__(self, X, y, query_ids, device="cuda"):
        """X: [n_items, F] array (dense or scipy sparse), y: [n_items], query_ids: [n_items]; items of one query must be
        contiguous (as in libsvm LTR files); queries keep their order of first appearance (dataset_loading.py:109-113)."""
        if hasattr(X, "toarray"):
            X = X.toarray()
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = np.asarray(y, dtype=np.float32)
        q = np.asarray(query_ids)
        change = np.flatnonzero(q[1:] != q[:-1]) + 1
        starts = np.concatenate([[0], change]).astype(np.int64)
        offsets = np.concatenate([starts, [len(q)]]).astype(np.int64)
        lens = np.diff(offsets)
        self.n_slates = int(len(lens))
        self.n_features = int(X.shape[1])
        self.longest_query_length = int(lens.max())
        self.device = torch.device(device)
        self.x_items = torch.from_numpy(X).to(self.device)
        self.y_items = torch.from_numpy(y).to(self.device)
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
    def from_svm_file(cls, path, device="cuda"):
        from sklearn.datasets import load_svmlight_file
        X, y, qid = load_svmlight_file(path, query_id=True)
        return cls(X, y, qid, device)

    def __len__(self):
        return self.n_slates

    @property
    def shape(self):
        return [self.n_slates, self.longest_query_length, self.n_features]

    # ------------------------------------------------------------------------------------------------------------
    def _positions(self, slates, L, generator):
        """[B, L] positions inside each slate (-1 = padding) following FixLength."""
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

    def batch(self, slates, slate_length, generator=None):
        """slates: i64[B] slate ids -> (xb [B,L,F], yb [B,L], indices [B,L]) on the device."""
        L = int(slate_length)
        pos = self._positions(slates, L, generator)
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
        for s in range(0, self.n_slates, batch_size):
            ids = order[s:s + batch_size]
            if drop_last and ids.numel() < batch_size:
                break
            yield self.batch(ids, L, generator)


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
