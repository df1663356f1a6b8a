"""CPU restatement of the reference's host data path -- TEST INFRASTRUCTURE (only tests/, smoke() and bench.py's cpu_baseline leg
may import this; the product's loader is allrank_amd/data.py and never touches it).

What is restated, with the numpy / torch calls in the reference's order so that the same seeds give the same bits:
    allrank/data/dataset_loading.py:96-127   LibSVMDataset.__init__   queries in order of first appearance, one array per query
    allrank/data/dataset_loading.py:32-93    FixLength                pad (features 0, label -1, index -1) or sample without
                                                                       replacement with the relevance rule (:70-77)
    allrank/data/dataset_loading.py:19-29    ToTensor                 float32 / float32 / int64 tensors
    allrank/data/dataset_loading.py:197-248  load_libsvm_dataset, create_data_loaders (torch DataLoader: train shuffled, validation
                                             not, drop_last False, batch = processing units x batch_size)
Pinned: tests/test_loader_cpu.py compares every batch of two epochs with the reference's own loaders (imported from /root/reference
in the build container) bit for bit, padding AND sampling branch, under main.py:36-38's seeds.
"""
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

PAD_Y = -1
PAD_INDEX = -1


def fix_length(x, y, target, rng=np.random):
    """FixLength.__call__ (:46-59): fewer than ``target`` items -> _pad (:81-93), otherwise _sample (:61-79; a slate of exactly
    ``target`` items is 'sampled', i.e. permuted).  ``rng``: the numpy module (= its global generator, as the reference) or a
    RandomState."""
    n = len(y)
    if n < target:
        gap = target - n
        return (np.pad(x, ((0, gap), (0, 0)), "constant"), np.pad(y, (0, gap), "constant", constant_values=PAD_Y),
                np.pad(np.arange(0, n), (0, gap), "constant", constant_values=PAD_INDEX))
    while True:
        pick = rng.choice(n, target, replace=False)                                        # :70
        if y[pick].sum() == 0:
            if y.sum() == 1:                                                               # :72-74 keep the only relevant item
                pick = np.concatenate([rng.choice(pick, target - 1, replace=False), [np.argmax(y)]])
            elif y.sum() > 0:                                                              # :75-76 draw again
                continue
        return x[pick], y[pick], pick


class HostSlates(Dataset):
    """LibSVMDataset (:96-165): ``X`` scipy-sparse or dense [n_items, F], ``y`` [n_items], ``qid`` [n_items]"""

    def __init__(self, X, y, qid, slate_length=None):
        X = X.toarray() if hasattr(X, "toarray") else np.asarray(X)
        _, first, counts = np.unique(qid, return_index=True, return_counts=True)           # :109
        cuts = np.cumsum(counts[np.argsort(first)])                                        # :110 (order of first appearance)
        self.xs = np.split(X, cuts)[:-1]
        self.ys = np.split(y, cuts)[:-1]
        self.longest_query_length = max(len(a) for a in self.xs)
        self.slate_length = slate_length

    @classmethod
    def from_svm_file(cls, path, slate_length=None):
        from sklearn.datasets import load_svmlight_file
        X, y, qid = load_svmlight_file(path, query_id=True)                                # :130
        return cls(X, y, qid, slate_length)

    def __len__(self):
        return len(self.xs)

    def __getitem__(self, i):
        x, y, idx = fix_length(self.xs[i], self.ys[i], int(self.slate_length))
        return (torch.from_numpy(x).type(torch.float32), torch.from_numpy(y).type(torch.float32),       # ToTensor :28
                torch.from_numpy(idx).type(torch.long))

    @property
    def shape(self):
        return [len(self), self.longest_query_length, self.xs[0].shape[-1]]


def load_libsvm_dataset(input_path, slate_length, validation_ds_role):
    """:197-227: the training role fixed to ``slate_length``, the validation role to its own longest slate"""
    train = HostSlates.from_svm_file(os.path.join(input_path, "train.txt"), int(slate_length))
    val = HostSlates.from_svm_file(os.path.join(input_path, "%s.txt" % validation_ds_role))
    val.slate_length = int(val.longest_query_length)
    return train, val


def create_data_loaders(train_ds, val_ds, num_workers, batch_size, units=1):
    """:230-248 with ``units`` processing units (the reference: max(1, torch.cuda.device_count()))"""
    total = max(1, units) * batch_size
    return (DataLoader(train_ds, batch_size=total, num_workers=num_workers, shuffle=True),
            DataLoader(val_ds, batch_size=total, num_workers=num_workers, shuffle=False))
