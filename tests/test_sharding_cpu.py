"""world_size-2 checks of the slate-sharded data-parallel path on CPU (gloo): the reduction algebra of SURVEY.md §8e
(loss normalised by the GLOBAL batch, gradients SUMMED, batch-global normalisers all-reduced) and the flat gradient
buffer exchange.  The per-rank arithmetic is stood in for by the numpy oracle (the HIP kernels need a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from allrank_amd import parallel, sharding
        from oracle import ltr_oracle as O
        from tests.golden.make_inputs import make_inputs
        G, L = 7, 33                                   # uneven: 4 + 3 slates
        s, y = make_inputs(G, L, 5)
        lo, hi = parallel.shard_slates(G, rank, world)
        out = {}
        # (a) mean-type loss: local sum / GLOBAL batch, then SUM over ranks == reference loss on the gathered batch
        with sharding.shard_context(G):
            div = sharding.batch_divisor(hi - lo)
            assert div == G
            per = O.approxndcg(s[lo:hi], y[lo:hi])[2]
            local_loss = torch.tensor([-float(per.sum()) / div], dtype=torch.float64)
            g_local = O.approxndcg(s[lo:hi], y[lo:hi])[1] * (hi - lo) / div       # oracle grad is /local B
            sharding.allreduce_sum_(local_loss)
        out["approx_loss"] = float(local_loss)
        out["approx_ref"] = float(O.approxndcg(s, y)[0])
        out["approx_grad_err"] = float(np.abs(g_local - O.approxndcg(s, y)[1][lo:hi]).max())
        # (b) neuralNDCG: the normaliser is the GLOBAL count of slates with idcg != 0 (neuralNDCG.py:69)
        with sharding.shard_context(G):
            idcg = O.dcg(y[lo:hi], y[lo:hi], [L])[0][:, 0]
            cnt = torch.tensor([float((idcg != 0).sum())], dtype=torch.float64)
            sharding.allreduce_sum_(cnt)
        out["cnt"] = float(cnt)
        out["cnt_ref"] = float((O.dcg(y, y, [L])[0][:, 0] != 0).sum())
        # (c) flat gradient buffer: one all_reduce sums every parameter gradient
        lin = torch.nn.Linear(5, 3)
        torch.manual_seed(0)
        fg = parallel.FlatGradients(lin.parameters())
        x = torch.full((2, 5), float(rank + 1))
        lin(x).sum().backward()
        assert lin.weight.grad.data_ptr() == fg.flat.data_ptr()          # autograd accumulated INTO the flat views
        fg.all_reduce()
        out["wgrad"] = lin.weight.grad.clone().numpy()
        out["bgrad"] = lin.bias.grad.clone().numpy()
        fg.zero()
        assert float(lin.weight.grad.abs().sum()) == 0.0
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_world_size_2_reduction_algebra():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert abs(o["approx_loss"] - o["approx_ref"]) < 1e-6
        assert o["approx_grad_err"] < 1e-7
        assert o["cnt"] == o["cnt_ref"]
        # each rank contributed x = rank+1 in every input slot, 2 rows: weight grad = 2*(1+2) = 6 per entry
        assert np.allclose(o["wgrad"], 6.0) and np.allclose(o["bgrad"], 4.0)


def test_shard_slates_partition():
    from allrank_amd import parallel
    for n, w in [(64, 8), (7, 2), (5, 8), (1, 1)]:
        spans = [parallel.shard_slates(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_shard_context_is_thread_local():
    """two replica threads of one process (the reference's nn.DataParallel layout, model_utils.py:40-53) each see their own
    divisor; a thread that never entered a context sees none"""
    import threading
    from allrank_amd import sharding
    seen, barrier = {}, threading.Barrier(3)

    def replica(name, gb):
        if gb is None:
            barrier.wait()
            seen[name] = (sharding.active(), sharding.batch_divisor(5))
            barrier.wait()
            return
        with sharding.shard_context(gb):
            barrier.wait()                               # all three threads are inside their (non-)contexts at the same time
            seen[name] = (sharding.active(), sharding.batch_divisor(5))
            barrier.wait()

    ts = [threading.Thread(target=replica, args=a) for a in (("a", 16), ("b", 48), ("c", None))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert seen == {"a": (True, 16.0), "b": (True, 48.0), "c": (False, 5.0)}
    assert not sharding.active()
