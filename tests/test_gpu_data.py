"""-m gpu: the on-device FixLength / batch-assembly kernels (allrank_amd/csrc/ltrx_data.hip) against
FixLength's definition (allrank/data/dataset_loading.py:32-93): the padding branch exactly, the sampling branch through its
invariants (a subset without repeats, the gathered features / labels, the relevance rules of :72-76) and distributionally
(uniform inclusion, uniform order) -- and against the independent torch restatement in allrank_amd/data.py, which the CPU suite
pins to the reference's own FixLength."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _toy(seed=0, n_q=40, F=8):
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, 30, n_q)
    lens[3] = 50
    lens[7] = 12
    lens[11] = 300
    X = rng.standard_normal((lens.sum(), F)).astype(np.float32)
    y = rng.choice(5, size=lens.sum(), p=[0.7, 0.15, 0.1, 0.03, 0.02]).astype(np.float32)
    qid = np.repeat(np.arange(100, 100 + n_q), lens)
    off = np.concatenate([[0], np.cumsum(lens)])
    y[off[3]:off[4]] = 0
    y[off[3] + 41] = 1                     # slate 3: exactly one relevant document
    y[off[5]:off[6]] = 0                   # slate 5: none
    y[off[11]:off[12]] = 0
    y[off[11] + 7] = 2
    y[off[11] + 250] = 1                   # slate 11 (300 items): two relevant ones -> resampling rule
    return X, y, qid, lens, off


def test_padding_branch_is_exact():
    from allrank_amd.data import DeviceSlates
    X, y, qid, lens, off = _toy()
    ds = DeviceSlates(X, y, qid, device=DEV)
    L = 320
    xb, yb, idx = next(ds.batches(40, L))
    xb, yb, idx = xb.cpu(), yb.cpu(), idx.cpu()
    for s in range(40):
        n = lens[s]
        assert np.array_equal(xb[s, :n].numpy(), X[off[s]:off[s] + n]) and torch.all(xb[s, n:] == 0)
        assert np.array_equal(yb[s, :n].numpy(), y[off[s]:off[s] + n]) and torch.all(yb[s, n:] == -1)
        assert idx[s, :n].tolist() == list(range(n)) and torch.all(idx[s, n:] == -1)
    assert xb.dtype == torch.float32 and yb.dtype == torch.float32 and idx.dtype == torch.int64
    # and identical to the torch restatement (same transform, no randomness on this branch)
    xt, yt, it = ds.batch_torch(torch.arange(40, device=DEV), L)
    assert torch.equal(xt.cpu(), xb) and torch.equal(yt.cpu(), yb) and torch.equal(it.cpu(), idx)


def test_sampling_branch_invariants_and_relevance_rules():
    from allrank_amd.data import DeviceSlates
    X, y, qid, lens, off = _toy()
    ds = DeviceSlates(X, y, qid, device=DEV)
    L = 10
    slates = torch.arange(40, device=DEV)
    orders = set()
    for rep in range(40):
        xb, yb, idx = (t.cpu() for t in ds.batch(slates, L, seed=1000 + rep))
        for s in range(40):
            n = lens[s]
            if n < L:
                assert idx[s, :n].tolist() == list(range(n)) and torch.all(idx[s, n:] == -1)
                continue
            ii = idx[s].numpy()
            assert len(set(ii.tolist())) == L and ii.min() >= 0 and ii.max() < n
            assert np.array_equal(xb[s].numpy(), X[off[s] + ii]) and np.array_equal(yb[s].numpy(), y[off[s] + ii])
            if y[off[s]:off[s] + n].sum() > 0:
                assert yb[s].sum() > 0                                           # dataset_loading.py:71-76
        assert 41 in idx[3].tolist()                                             # the only relevant document is kept (:72-74)
        orders.add(tuple(idx[7].tolist()))
    assert len(orders) > 30                                                      # random subset / order, new every batch
    # the same seed reproduces the batch, another seed does not
    a = ds.batch(slates, L, seed=5)
    b = ds.batch(slates, L, seed=5)
    c = ds.batch(slates, L, seed=6)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and not torch.equal(a[2], c[2])


def test_sampling_is_uniform_without_replacement_in_random_order():
    """slate of 60 items without any relevance (no resampling): inclusion probability L/n for every item, every slot uniform
    over the items -- compared with the torch restatement's frequencies and with the exact values (5-sigma bands)."""
    from allrank_amd.data import DeviceSlates
    n, L, reps = 60, 20, 3000
    X = np.random.default_rng(0).standard_normal((n * 64, 4)).astype(np.float32)
    y = np.zeros(n * 64, np.float32)
    qid = np.repeat(np.arange(64), n)
    ds = DeviceSlates(X, y, qid, device=DEV)
    slates = torch.arange(64, device=DEV)
    incl = torch.zeros(n, device=DEV)
    first = torch.zeros(n, device=DEV)
    pair = torch.zeros(n, n, device=DEV)
    for rep in range(reps // 64 + 1):
        idx = ds.positions(slates, L, seed=77 + rep)
        incl += torch.bincount(idx.reshape(-1), minlength=n).float()
        first += torch.bincount(idx[:, 0], minlength=n).float()
        pair[idx[:, 0], idx[:, 1]] += 1
    tot = float((reps // 64 + 1) * 64)
    p = L / n
    sd = (tot * p * (1 - p)) ** 0.5
    assert float((incl - tot * p).abs().max()) < 5 * sd, (incl.min().item(), incl.max().item(), tot * p)
    p1 = 1.0 / n
    sd1 = (tot * p1 * (1 - p1)) ** 0.5
    assert float((first - tot * p1).abs().max()) < 5 * sd1 + 1
    assert float(torch.diagonal(pair).sum()) == 0                                # without replacement
    # the torch restatement (pinned to the reference's FixLength on CPU) has the same inclusion frequencies
    g = torch.Generator(device=DEV).manual_seed(3)
    incl_t = torch.zeros(n, device=DEV)
    for rep in range(reps // 64 + 1):
        incl_t += torch.bincount(ds._positions_torch(slates, L, g).reshape(-1), minlength=n).float()
    assert float((incl_t - incl).abs().max()) < 7 * sd


def test_epoch_iterator_and_end_to_end_shapes():
    from allrank_amd.data import DeviceSlates
    X, y, qid, lens, off = _toy()
    ds = DeviceSlates(X, y, qid, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    seen = 0
    for xb, yb, idx in ds.batches(16, 24, shuffle=True, generator=g):
        assert xb.shape[1:] == (24, 8) and yb.shape == idx.shape == xb.shape[:2]
        seen += xb.shape[0]
        assert torch.all((yb == -1) == (idx == -1))
    assert seen == 40


def test_device_libsvm_parser_is_bit_identical_to_sklearn(tmp_path):
    """the device parser (ltrx_libsvm_parse) against scikit-learn's load_svmlight_file -- what the reference calls at
    dataset_loading.py:130 -- on the same text: features, labels and query ids bit for bit (float64 parse narrowed to fp32, as the
    reference's ToTensor does), including sparse rows, negative / exponent / long-mantissa numbers, comments and a zero-based file."""
    from sklearn.datasets import dump_svmlight_file, load_svmlight_file
    from allrank_amd.data import DeviceSlates, parse_svm_file_on_device
    rng = np.random.default_rng(5)
    n, F = 700, 33
    X = (rng.standard_normal((n, F)) * np.exp(rng.uniform(-12, 12, (n, F)))).astype(np.float64)
    X[rng.random((n, F)) < 0.3] = 0.0                         # sparse entries are omitted from the text
    X[:, F - 1] = rng.standard_normal(n)                      # (keep the last column populated: n_features is inferred)
    y = rng.integers(0, 5, n).astype(np.float64)
    qid = np.repeat(np.arange(10, 10 + n // 35), 35)
    path = str(tmp_path / "train.txt")
    dump_svmlight_file(X, y, path, query_id=qid, zero_based=False)
    with open(path, "a") as fh:                               # hand-written lines: comment, exponents, many digits, tabs, CRLF
        fh.write("# a comment line\n")
        fh.write("3 qid:99 1:1e-3 2:-2.5E+2 5:0.12345678901234567 33:7\t# trailing comment\r\n")
        fh.write("0 qid:99 4:123456789012345678 33:-0.0\n")
    Xs, ys, qs = load_svmlight_file(path, query_id=True)
    Xd, yd, qd = parse_svm_file_on_device(path, DEV)
    assert Xd.shape == Xs.shape
    assert np.array_equal(Xd.cpu().numpy(), Xs.toarray().astype(np.float32))
    assert np.array_equal(yd.cpu().numpy(), ys.astype(np.float32)) and np.array_equal(qd.cpu().numpy(), qs)
    ds = DeviceSlates.from_svm_file(path, device=DEV)
    ref = DeviceSlates.from_svm_file(path, device=DEV, parser="sklearn")
    assert ds.n_slates == ref.n_slates and torch.equal(ds.x_items, ref.x_items) and torch.equal(ds.offsets, ref.offsets)
    # zero-based file (smallest index 0): scikit-learn's zero_based="auto" rule
    path0 = str(tmp_path / "zero.txt")
    dump_svmlight_file(X[:50], y[:50], path0, query_id=qid[:50], zero_based=True)
    X0, _, _ = load_svmlight_file(path0, query_id=True)
    Xd0, _, _ = parse_svm_file_on_device(path0, DEV)
    assert np.array_equal(Xd0.cpu().numpy(), X0.toarray().astype(np.float32))
