"""torchrun worker for test_sharded_step_equals_single_rank_step (2 ranks on one GPU, gloo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allrank_amd.engine import FusedTrainer  # noqa: E402
from allrank_amd.model import make_model  # noqa: E402
from allrank_amd import parallel  # noqa: E402


def build():
    torch.manual_seed(7)
    return make_model(dict(sizes=[32], input_norm=False, activation=None, dropout=0.0),
                      dict(N=1, d_ff=64, h=4, positional_encoding=None, dropout=0.0),
                      dict(d_output=1, output_activation=None), 20).to("cuda:0")


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    G, L = 8, 40
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.standard_normal((G, L, 20)).astype(np.float32), device="cuda:0")
    y = torch.tensor(rng.integers(0, 5, (G, L)).astype(np.float32), device="cuda:0")
    y[5, 30:] = -1
    for loss_name, args in (("approxNDCGLoss", {}), ("neuralNDCG", {}), ("lambdaLoss", dict(weighing_scheme="lambdaRank_scheme", reduction="mean")),
                            ("rankNet_weightByGTDiff", {})):
        lo, hi = parallel.shard_slates(G, rank, world)
        m_sh = build()
        ft = FusedTrainer(m_sh, loss_name, args, hi - lo, L, lr=1e-3, world_size=world, use_graph=False, gemm="hipblaslt")
        share = ft.step(x[lo:hi], y[lo:hi], global_batch=G).clone()
        dist.all_reduce(share)
        m_1 = build()
        f1 = FusedTrainer(m_1, loss_name, args, G, L, lr=1e-3, world_size=1, use_graph=False, gemm="hipblaslt")
        full = f1.step(x, y)
        if loss_name == "lambdaLoss":
            pass                                  # mean over the global pair count: every rank already holds loss_share
        assert abs(share.item() - full.item()) <= 1e-5 * (1 + abs(full.item())), (loss_name, share.item(), full.item())
        gerr = (ft.flat_g - f1.flat_g).abs().max().item() / max(f1.flat_g.abs().max().item(), 1e-12)
        assert gerr < 1e-4, (loss_name, "grad", gerr)
        werr = (ft.flat_p - f1.flat_p).abs().max().item()
        assert werr <= 2.1e-3, (loss_name, "weights", werr)   # one Adam step of lr=1e-3; ~0-gradient params may flip sign
    # the slate-resident FC + ListNet step (csrc/ltrx_fcstep.hip), sharded: gradients only in the kernels, all-reduce(SUM) of the flat
    # buffer, then the flat-buffer Adam -- both forms (two-layer MFMA evaluation with ReLU, linear-scorer collapse) against one rank
    def build_fc(act):
        torch.manual_seed(11)
        return make_model(dict(sizes=[48], input_norm=False, activation=act, dropout=0.0), None, dict(d_output=1, output_activation=None), 20).to("cuda:0")
    lo, hi = parallel.shard_slates(G, rank, world)
    for act, mode in (("ReLU", True), (None, True), (None, "collapse")):
        m_sh, m_1 = build_fc(act), build_fc(act)
        ft = FusedTrainer(m_sh, "listNet", {}, hi - lo, L, lr=1e-3, world_size=world, use_graph=False, fc_step=mode)
        f1 = FusedTrainer(m_1, "listNet", {}, G, L, lr=1e-3, world_size=1, use_graph=False, fc_step=mode)
        assert ft.fcstep == mode and f1.fcstep == mode
        share = ft.step(x[lo:hi], y[lo:hi], global_batch=G).clone()
        dist.all_reduce(share)
        full = f1.step(x, y)
        assert abs(share.item() - full.item()) <= 1e-5 * (1 + abs(full.item())), ("fcstep", act, mode, share.item(), full.item())
        gerr = (ft.flat_g - f1.flat_g).abs().max().item() / max(f1.flat_g.abs().max().item(), 1e-12)
        assert gerr < 1e-4, ("fcstep", act, mode, "grad", gerr)
        werr = (ft.flat_p - f1.flat_p).abs().max().item()
        assert werr <= 2.1e-3, ("fcstep", act, mode, "weights", werr)
    # the CAPTURED sharded step (hipGraph segments with the collectives between them, engine.FusedTrainer._capture) == the
    # eager sharded step, bit for bit, step after step (same kernels, same order, same collectives), and == the one-rank step
    # on the gathered batch; five steps = two eager warm-ups, the capture step, two replays
    lo, hi = parallel.shard_slates(G, rank, world)
    for loss_name, args in (("approxNDCGLoss", {}), ("neuralNDCG", {}), ("lambdaLoss", dict(weighing_scheme="lambdaRank_scheme", reduction="mean"))):
        m_g, m_e, m_1 = build(), build(), build()
        tg = FusedTrainer(m_g, loss_name, args, hi - lo, L, lr=1e-3, world_size=world, use_graph=True)
        te = FusedTrainer(m_e, loss_name, args, hi - lo, L, lr=1e-3, world_size=world, use_graph=False)
        t1 = FusedTrainer(m_1, loss_name, args, G, L, lr=1e-3, world_size=1, use_graph=True)
        for step in range(5):
            lg = tg.step(x[lo:hi], y[lo:hi], global_batch=G).clone()
            le = te.step(x[lo:hi], y[lo:hi], global_batch=G).clone()
            l1 = t1.step(x, y).clone()
            assert torch.equal(lg, le), (loss_name, step, "graph vs eager loss", lg.item(), le.item())
            assert torch.equal(tg.flat_p, te.flat_p), (loss_name, step, "graph vs eager weights")
            tot = lg.clone()
            dist.all_reduce(tot)
            if step == 0:          # (free-running Adam trajectories decorrelate after the first update; step 0 is exact algebra)
                assert abs(tot.item() - l1.item()) <= 1e-5 * (1 + abs(l1.item())), (loss_name, tot.item(), l1.item())
                gerr = (tg.flat_g - t1.flat_g).abs().max().item() / max(t1.flat_g.abs().max().item(), 1e-12)
                assert gerr < 2e-4, (loss_name, "grad vs one rank", gerr)
        assert tg.capture_fallback is None, (loss_name, tg.capture_fallback)          # the captured sharded path really ran
        assert tg.graph is not None and len(tg.graph) >= 1 + len(tg._buckets), (loss_name, "segments", None if tg.graph is None else len(tg.graph))
    # count-normalised pointwise losses through the plugin functions: the rank shares (each divided by the GLOBAL count,
    # all-reduced inside the loss) add up to the single-process value, and so do the gradients
    from allrank_amd import losses as E, sharding
    lo, hi = parallel.shard_slates(G, rank, world)
    p = torch.sigmoid(x[:, :, 0]).contiguous()
    p3 = torch.sigmoid(x[:, :, :3]).contiguous()
    yb = torch.where(y == -1, y, (y >= 2).float())
    for fn, pred, tgt, kw in ((E.bce, p, yb, {}), (E.ordinal, p3, y, dict(n=3)), (E.rankNet, x[:, :, 1].contiguous(), y, {}),
                              (E.pointwise_rmse, p, y, dict(no_of_levels=4)), (E.binary_listNet, x[:, :, 2].contiguous(), yb, {})):
        full_in = pred.clone().requires_grad_(True)
        full = fn(full_in, tgt, **kw)
        full.backward()
        part_in = pred[lo:hi].clone().requires_grad_(True)
        with sharding.shard_context(G, None):
            part = fn(part_in, tgt[lo:hi], **kw)
        part.backward()
        tot = part.detach().clone()
        dist.all_reduce(tot)
        assert abs(tot.item() - full.item()) <= 1e-5 * (1 + abs(full.item())), (fn.__name__, tot.item(), full.item())
        gerr = (part_in.grad - full_in.grad[lo:hi]).abs().max().item()
        assert gerr <= 1e-6 + 1e-4 * full_in.grad.abs().max().item(), (fn.__name__, "grad", gerr)
    if rank == 0:
        print("EQUIV_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
