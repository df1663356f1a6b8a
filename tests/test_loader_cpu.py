"""The loader boundary (round 6): ``load_libsvm_dataset`` / ``create_data_loaders`` of allrank_amd/data.py behind the
names of allrank/data/dataset_loading.py:197-248 / main.py:8,57-68.

CPU half (no GPU here):
  * oracle/loader_oracle.py (the restated host loader the GPU tests and the bench compare with) == the reference's own loaders, every
    batch of two epochs bit for bit, padding and sampling branch, under main.py:36-38's seeds            [needs /root/reference]
  * DeviceLoader's batch ORDER, its draws from torch's global generator, ``burn()`` and the rank blocks == the reference loader's
    behaviour.  The batch assembly itself is the HIP kernels' job (tests/test_gpu_loader.py); here the torch restatement
    ``DeviceSlates.batch_torch`` (pinned to the reference's FixLength by tests/test_data_cpu.py) is injected in their place.
  * install(): the loader names are rebound and restored; a bare install() never creates a process group; uninstall() restores the
    TRUE originals also when allrank.main was imported between two install() calls
  * launch._device_of under a per-rank HIP_VISIBLE_DEVICES.
"""
import os

import numpy as np
import pytest
import torch

from oracle.ref_loader import reference_available
from oracle import loader_oracle as LO


def _write(tmp_path, seed=0, n_q=37, F=9, max_len=30, long=None):
    """train.txt / vali.txt under tmp_path; ``long``: {query: length} overrides"""
    from sklearn.datasets import dump_svmlight_file
    rng = np.random.default_rng(seed)
    for role in ("train", "vali"):
        lens = rng.integers(1, max_len, n_q)
        for q, n in (long or {}).items():
            lens[q] = n
        X = np.round(rng.standard_normal((lens.sum(), F)), 4)
        X[:, F - 1] = np.round(rng.uniform(0.5, 1.5, lens.sum()), 4)          # (last column populated: n_features is inferred)
        y = rng.choice(5, size=lens.sum(), p=[0.6, 0.2, 0.1, 0.06, 0.04]).astype(np.float64)
        qid = np.repeat(np.arange(500, 500 + n_q), lens)
        dump_svmlight_file(X, y, str(tmp_path / ("%s.txt" % role)), query_id=qid)
    return str(tmp_path)


def _seed():
    torch.manual_seed(42)                      # main.py:36-38
    np.random.seed(42)


def _epochs(train_dl, val_dl, n=2, extra_train=1, extra_val=1):
    """the loader traffic of the reference's fit (train_utils.py:95-107): per epoch train_dl 1 + extra_train times, valid_dl
    1 + extra_val times; returns every batch"""
    out = []
    for _ in range(n):
        for _ in range(1 + extra_train):
            out += [tuple(t.clone() for t in b) for b in train_dl]
        for _ in range(1 + extra_val):
            out += [tuple(t.clone() for t in b) for b in val_dl]
    return out


def _canon(batch):
    """rows without padding went through FixLength's SAMPLING branch (a slate of exactly L items is randomly permuted,
    dataset_loading.py:55-58 -- every longest slate of a validation set): compared as sets, i.e. re-ordered by original index"""
    x, y, i = (t.cpu().clone() for t in batch)
    for r in range(i.shape[0]):
        if bool((i[r] >= 0).all()):
            o = torch.argsort(i[r])
            x[r], y[r], i[r] = x[r][o], y[r][o], i[r][o]
    return x, y, i


def _same(a, b, canon=False):
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert len(u) == len(v) == 3
        if canon:
            u, v = _canon(u), _canon(v)
        for s, t in zip(u, v):
            assert s.dtype == t.dtype and s.shape == t.shape and torch.equal(s.cpu(), t.cpu())


@pytest.mark.skipif(not reference_available(), reason="needs a checkout of allegro/allRank (ALLRANK_REFERENCE)")
@pytest.mark.parametrize("slate_length", [40, 12])          # 40: padding only; 12: sampling branch incl. the relevance rules
def test_oracle_loader_equals_the_reference_loader(tmp_path, slate_length):
    from oracle.ref_loader import load_reference
    load_reference(stable_sort=False)
    import allrank.data.dataset_loading as RD
    path = _write(tmp_path, long={3: 33, 8: 12})
    _seed()
    ref = _epochs(*RD.create_data_loaders(*RD.load_libsvm_dataset(path, slate_length, "vali"), num_workers=0, batch_size=8))
    _seed()
    mine = _epochs(*LO.create_data_loaders(*LO.load_libsvm_dataset(path, slate_length, "vali"), num_workers=0, batch_size=8))
    _same(ref, mine)


@pytest.fixture
def torch_assembly(monkeypatch):
    """DeviceSlates.batch -> its torch restatement (no GPU in this container; the kernels are compared with it in test_gpu_data.py)"""
    from allrank_amd.data import DeviceSlates

    def batch(self, slates, slate_length, generator=None, seed=None):
        slates = slates.to(self.device, torch.int64)
        if slates.numel() == 0:
            L = int(slate_length)
            return (torch.empty((0, L, self.n_features)), torch.empty((0, L)), torch.empty((0, L), dtype=torch.int64))
        # (a private generator: the product path draws nothing from torch's global generator when it assembles a batch)
        return self.batch_torch(slates, slate_length, torch.Generator().manual_seed(int(seed or 0) + 1))
    monkeypatch.setattr(DeviceSlates, "batch", batch)


def _device_loaders(path, slate_length, batch_size, rank=0, world=1):
    from allrank_amd import data as ED
    tr, va = ED.load_libsvm_dataset(path, slate_length, "vali", device="cpu")
    return (ED.DeviceLoader(tr, world * batch_size, shuffle=True, rank=rank, world=world),
            ED.DeviceLoader(va, world * batch_size, shuffle=False, rank=rank, world=world))


def test_device_loader_batches_equal_the_host_loader_padded_only(tmp_path, torch_assembly):
    """same seeds -> same slates per batch, same order, same tensors (the padding branch is exact), for the loader traffic of two
    epochs of the reference's fit -- the torch generator ends in the same state"""
    path = _write(tmp_path)
    _seed()
    ref = _epochs(*LO.create_data_loaders(*LO.load_libsvm_dataset(path, 40, "vali"), num_workers=0, batch_size=8))
    state_ref = torch.get_rng_state()
    _seed()
    tr, va = _device_loaders(path, 40, 8)
    mine = _epochs(tr, va)
    assert torch.equal(torch.get_rng_state(), state_ref)
    _same(ref, mine, canon=True)
    assert tr.batch_size == 8 and len(tr) == 5 and tr.batch_shape()[:2] == (8, 40) and 0.2 < tr.batch_shape()[2] < 0.5
    assert va.slate_length == va.dataset.longest_query_length and va.dataset.samples and not tr.dataset.samples


def test_burn_consumes_what_one_iteration_draws(tmp_path, torch_assembly):
    """fit() skips the reference's extra passes (train metrics, validation metrics) and burns their generator draws instead: the
    batches of the passes it DOES make must stay the reference's"""
    from allrank_amd import fit as EF
    path = _write(tmp_path)
    _seed()
    tr_h, va_h = LO.create_data_loaders(*LO.load_libsvm_dataset(path, 40, "vali"), num_workers=0, batch_size=8)
    ref = []
    for _ in range(3):                                      # the reference's traffic; only the first train / first val pass is kept
        ref += [tuple(t.clone() for t in b) for b in tr_h]
        list(tr_h)
        ref += [tuple(t.clone() for t in b) for b in va_h]
        list(va_h)
    _seed()
    tr, va = _device_loaders(path, 40, 8)
    mine = []
    for _ in range(3):
        mine += list(tr)
        assert EF._burn(tr)
        mine += list(va)
        assert EF._burn(va)
    _same(ref, mine, canon=True)
    # ... and the generic form for torch DataLoaders (what fit() does when it is handed the reference's own loaders)
    _seed()
    tr_b, va_b = LO.create_data_loaders(*LO.load_libsvm_dataset(path, 40, "vali"), num_workers=0, batch_size=8)
    again = []
    for _ in range(3):
        again += [tuple(t.clone() for t in b) for b in tr_b]
        assert EF._burn(tr_b)
        again += [tuple(t.clone() for t in b) for b in va_b]
        assert EF._burn(va_b)
    _same(ref, again, canon=True)


def test_rank_blocks_partition_every_global_batch(tmp_path, torch_assembly):
    """world 2 (and 3: uneven blocks, an empty block on the short last batch): concatenating the ranks' blocks gives the one-rank
    batch; every ShardBatch carries the global slate count, its offset, the same order tag on every rank, host lengths"""
    from allrank_amd.data import ShardBatch
    from allrank_amd.parallel import shard_slates
    path = _write(tmp_path, n_q=33)
    for world in (2, 3):
        _seed()
        one = _epochs(*_device_loaders(path, 40, 4 * world), n=1, extra_train=0, extra_val=0)
        per_rank = []
        for r in range(world):
            _seed()                                         # every rank seeds identically (main.py:36-38)
            tr, va = _device_loaders(path, 40, 4, rank=r, world=world)
            assert tr.batch_size == 4 * world
            per_rank.append(list(tr) + list(va))
        assert all(len(p) == len(one) for p in per_rank)
        for k, whole in enumerate(one):
            blocks = [p[k] for p in per_rank]
            assert all(isinstance(b, ShardBatch) for b in blocks)
            n = whole[0].shape[0]
            assert all(b.global_slates == n for b in blocks) and len({b.order_tag for b in blocks}) == 1
            for r, b in enumerate(blocks):
                lo, hi = shard_slates(n, r, world)
                assert b.offset == lo and b[0].shape[0] == hi - lo
                assert torch.equal(b.lengths, (b[1] != -1).sum(1).to(torch.int32))
            cat = _canon(tuple(torch.cat([b[j] for b in blocks]) for j in range(3)))
            for j, t in enumerate(_canon(whole)):
                assert torch.equal(cat[j], t)
        # 33 slates, global batch 12: the last batch has 9 slates -> world 3 blocks of 3; world 2 (batch 8): last batch 1 slate -> rank 1 empty
        if world == 2:
            assert per_rank[1][4][0].shape[0] == 0 and per_rank[0][4][0].shape[0] == 1


def test_create_data_loaders_rule_and_fallback(tmp_path, torch_assembly):
    """DeviceLibSVMDataset -> DeviceLoader; any other dataset -> the reference's torch DataLoader; total = units x batch_size"""
    from torch.utils.data import DataLoader, TensorDataset
    from allrank_amd import data as ED, launch
    path = _write(tmp_path)
    tr, va = ED.load_libsvm_dataset(path, 16, "vali", device="cpu")
    assert tr.slate_length == 16 and va.slate_length == va.longest_query_length and tr.shape == [37, tr.longest_query_length, 9]
    x, y, i = tr[0]
    assert x.shape == (16, 9) and y.shape == (16,) and i.dtype == torch.int64
    a, b = ED.create_data_loaders(tr, va, num_workers=3, batch_size=5)
    assert isinstance(a, ED.DeviceLoader) and isinstance(b, ED.DeviceLoader) and a.shuffle and not b.shuffle and a.batch_size == 5
    launch._state["world"], launch._state["rank"] = 4, 2
    try:
        a, b = launch.create_data_loaders(tr, va, num_workers=0, batch_size=5)
        assert a.batch_size == 20 and (a.rank, a.world) == (2, 4) and (b.rank, b.world) == (2, 4)
        host = TensorDataset(torch.arange(50).float())
        c, d = launch.create_data_loaders(host, host, num_workers=0, batch_size=5)
        assert isinstance(c, DataLoader) and c.batch_size == 20
    finally:
        launch._state["world"], launch._state["rank"] = 1, 0
    with pytest.raises(NotImplementedError, match="GCS"):
        ED.load_libsvm_role("gs://bucket/data", "train", device="cpu")


@pytest.mark.skipif(not reference_available(), reason="needs a checkout of allegro/allRank (ALLRANK_REFERENCE)")
def test_install_rebinds_and_restores_the_loader_names(monkeypatch):
    import sys
    from oracle.ref_loader import load_reference
    load_reference(stable_sort=False)
    import allrank.data.dataset_loading as RD
    import allrank.models.model as RM
    import allrank_amd
    from allrank_amd import data as ED, model as EMod
    allrank_amd.uninstall()
    sys.modules.pop("allrank.main", None)
    orig = {n: getattr(RD, n) for n in ("load_libsvm_dataset", "create_data_loaders", "load_libsvm_dataset_role")}
    orig_make = RM.make_model
    try:
        done = allrank_amd.install(fit=True)                 # no GPU here: data=None keeps the reference's host loaders
        assert "allrank.data.dataset_loading.load_libsvm_dataset" not in done and RD.load_libsvm_dataset is orig["load_libsvm_dataset"]
        # allrank.main imported AFTER the first install: its `from ... import` copies are already the engine's objects
        import allrank.main as M
        assert M.make_model is EMod.make_model
        done = allrank_amd.install(fit=True, data=True)
        for n in ("allrank.data.dataset_loading.load_libsvm_dataset", "allrank.main.load_libsvm_dataset",
                  "allrank.data.dataset_loading.create_data_loaders", "allrank.main.create_data_loaders"):
            assert n in done
        assert RD.load_libsvm_dataset is ED.load_libsvm_dataset and M.load_libsvm_dataset is ED.load_libsvm_dataset
        assert RD.create_data_loaders is ED.create_data_loaders and M.create_data_loaders is ED.create_data_loaders
        assert RD.load_libsvm_dataset_role is orig["load_libsvm_dataset_role"]          # rank_and_click.py:63 keeps the host dataset
    finally:
        allrank_amd.uninstall()
    import allrank.main as M
    for n, f in orig.items():
        assert getattr(RD, n) is f
    # main's names go back to the REFERENCE's objects, not to the engine's
    assert M.make_model is orig_make and M.load_libsvm_dataset is orig["load_libsvm_dataset"] and M.create_data_loaders is orig["create_data_loaders"]
    import allrank.training.train_utils as RT
    assert M.fit is RT.fit and RT.fit.__module__ == "allrank.training.train_utils"


@pytest.mark.skipif(not reference_available(), reason="needs a checkout of allegro/allRank (ALLRANK_REFERENCE)")
def test_bare_install_never_creates_a_process_group(monkeypatch, caplog):
    """under a launcher environment (torchrun exports RANK / WORLD_SIZE) install() used to call init_process_group itself"""
    import torch.distributed as dist
    from oracle.ref_loader import load_reference
    load_reference(stable_sort=False)
    import allrank_amd
    from allrank_amd import launch
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", str(launch.free_port()))
    assert not dist.is_initialized()
    try:
        with caplog.at_level("WARNING"):
            done = allrank_amd.install()
        assert not dist.is_initialized() and launch.world_size() == 1
        assert not any("get_torch_device" in n for n in done)
        assert any("does not create one" in r.getMessage() for r in caplog.records)
    finally:
        allrank_amd.uninstall()
        launch.shutdown()


def test_device_rule_under_a_per_rank_visible_device(monkeypatch):
    """a launcher that masks every rank down to its own GPU (HIP_VISIBLE_DEVICES=<r>) -> the rank's GPU is index 0"""
    from allrank_amd import launch
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    assert launch._device_of(5, "nccl") == torch.device("cuda", 0)
    assert launch._device_of(5, "nccl", devices=["0"] * 8) == torch.device("cuda", 0)      # (no 'several ranks on one GPU' refusal)
    # one visible GPU that is NOT a per-rank mask (a one-GPU box, two ranks asked for): still refused with the clear message
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    with pytest.raises(RuntimeError, match="only 1 device"):
        launch._device_of(1, "nccl")
    with pytest.raises(RuntimeError, match="several ranks on one GPU"):
        launch._device_of(1, "nccl", devices=["0", "0"])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert launch._device_of(3, "nccl") == torch.device("cuda", 3)
