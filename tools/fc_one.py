"""five steps of the slate-resident FC + ListNet step (csrc/ltrx_fcstep.hip) at FB x FL x FF, hidden FH -- the child process of
bench.py's PMC passes for the fc_listnet workload (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) and of tools/lab/pmc_fc.sh"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd.model import make_model
from allrank_amd.engine import FusedTrainer
B, L, F, H = (int(os.environ.get(k, d)) for k, d in (("FB", "256"), ("FL", "240"), ("FF", "136"), ("FH", "96")))
torch.manual_seed(42)
model = make_model(dict(sizes=[H], input_norm=False, activation=None, dropout=0.0), None, dict(d_output=1, output_activation=None), F).cuda()
ft = FusedTrainer(model, "listNet", {}, B, L, lr=1e-3, use_graph=False)
assert ft.fcstep
g = torch.Generator().manual_seed(1)
x = torch.randn(2 * B, L, F, generator=g).cuda()
y = torch.randint(0, 5, (2 * B, L), generator=g).float().cuda()
for i in range(5):
    ft.step(x[(i % 2) * B:(i % 2 + 1) * B], y[(i % 2) * B:(i % 2 + 1) * B])
torch.cuda.synchronize()
