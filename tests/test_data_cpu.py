"""The torch restatement of FixLength / ToTensor inside allrank_amd/data.py (``DeviceSlates.batch_torch``; device-agnostic torch
ops) == allrank/data/dataset_loading.py:19-93, checked on CPU tensors.  The product path (``DeviceSlates.batch`` -> the HIP kernels of
ltrx_data.hip) is compared with this restatement on the GPU (tests/test_gpu_data.py) and refuses CPU tensors."""
import numpy as np
import pytest
import torch

from allrank_amd.data import DeviceSlates
from oracle.ref_loader import reference_available


def _batches(ds, batch_size, slate_length=None, shuffle=False, generator=None):
    """DeviceSlates.batches with the torch implementation of the transform"""
    L = ds.longest_query_length if slate_length is None else int(slate_length)
    order = torch.randperm(ds.n_slates, generator=generator) if shuffle else torch.arange(ds.n_slates)
    for s in range(0, ds.n_slates, batch_size):
        yield ds.batch_torch(order[s:s + batch_size], L, generator)


def _toy(seed=0, n_q=40, F=6):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 30, n_q)
    lens[3] = 50
    lens[7] = 12
    X = rng.standard_normal((lens.sum(), F)).astype(np.float32)
    y = rng.choice(5, size=lens.sum(), p=[0.7, 0.15, 0.1, 0.03, 0.02]).astype(np.float32)
    qid = np.repeat(np.arange(100, 100 + n_q), lens)
    # slate 3 (50 items): exactly one relevant document
    o3 = lens[:3].sum()
    y[o3:o3 + 50] = 0
    y[o3 + 41] = 1
    # slate 5: no relevant document at all
    o5 = lens[:5].sum()
    y[o5:o5 + lens[5]] = 0
    return X, y, qid, lens


def test_padding_branch_matches_fixlength_pad():
    X, y, qid, lens = _toy()
    ds = DeviceSlates(X, y, qid, device="cpu")
    assert len(ds) == 40 and ds.longest_query_length == 50 and ds.shape == [40, 50, 6]
    L = 60
    xb, yb, idx = next(_batches(ds, 40, L))
    off = np.concatenate([[0], np.cumsum(lens)])
    for s in range(40):
        n = lens[s]
        assert np.array_equal(xb[s, :n].numpy(), X[off[s]:off[s] + n]) and torch.all(xb[s, n:] == 0)
        assert np.array_equal(yb[s, :n].numpy(), y[off[s]:off[s] + n]) and torch.all(yb[s, n:] == -1)
        assert idx[s, :n].tolist() == list(range(n)) and torch.all(idx[s, n:] == -1)
    assert xb.dtype == torch.float32 and yb.dtype == torch.float32 and idx.dtype == torch.int64


def test_sampling_branch_without_replacement_and_relevance_rules():
    X, y, qid, lens = _toy()
    ds = DeviceSlates(X, y, qid, device="cpu")
    g = torch.Generator().manual_seed(1)
    L = 10
    off = np.concatenate([[0], np.cumsum(lens)])
    seen_orders = set()
    for rep in range(30):
        xb, yb, idx = ds.batch_torch(torch.arange(40), L, g)
        for s in range(40):
            n = lens[s]
            if n < L:
                assert idx[s, :n].tolist() == list(range(n)) and torch.all(idx[s, n:] == -1)
                continue
            ii = idx[s].numpy()
            assert len(set(ii.tolist())) == L and ii.min() >= 0 and ii.max() < n          # a subset, no repeats
            assert np.array_equal(xb[s].numpy(), X[off[s] + ii]) and np.array_equal(yb[s].numpy(), y[off[s] + ii])
            tot = y[off[s]:off[s] + n].sum()
            if tot > 0:
                assert yb[s].sum() > 0                                                    # dataset_loading.py:71-76
        assert idx[3, L - 1] == 41 or 41 in idx[3].tolist()                                # the only relevant doc is kept
        seen_orders.add(tuple(idx[7].tolist()))
    assert len(seen_orders) > 5                                                            # random order / subset


def test_epoch_iterator_covers_every_slate_once():
    X, y, qid, lens = _toy()
    ds = DeviceSlates(X, y, qid, device="cpu")
    g = torch.Generator().manual_seed(0)
    nb = 0
    firsts = []
    for xb, yb, idx in _batches(ds, 16, 50, shuffle=True, generator=g):
        nb += xb.shape[0]
        firsts += [tuple(np.round(r, 5)) for r in xb[:, 0, :2].numpy().tolist()]
    assert nb == 40 and len(set(firsts)) == 40
    xv, yv, iv = next(_batches(ds, 64))                      # validation: pad to the longest slate
    assert xv.shape == (40, 50, 6)


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
def test_against_reference_fixlength_pad_and_libsvm_roundtrip(tmp_path):
    from oracle.ref_loader import load_reference
    load_reference()
    from allrank.data.dataset_loading import FixLength, ToTensor
    from sklearn.datasets import dump_svmlight_file
    X, y, qid, lens = _toy(seed=4, n_q=12)
    path = str(tmp_path / "train.txt")
    dump_svmlight_file(X, y, path, query_id=qid)
    ds = DeviceSlates.from_svm_file(path, device="cpu")
    L = 64
    xb, yb, idx = next(_batches(ds, 12, L))
    off = np.concatenate([[0], np.cumsum(lens)])
    fl, tt = FixLength(L), ToTensor()
    for s in range(12):
        rx, ry, ri = tt(fl((X[off[s]:off[s + 1]].astype(np.float64), y[off[s]:off[s + 1]], None)))
        assert torch.allclose(xb[s], rx, atol=1e-6) and torch.equal(yb[s], ry) and torch.equal(idx[s], ri)


def test_product_batch_path_refuses_cpu_tensors():
    X, y, qid, lens = _toy()
    ds = DeviceSlates(X, y, qid, device="cpu")
    with pytest.raises(RuntimeError):
        ds.batch(torch.arange(4), 16, seed=1)
