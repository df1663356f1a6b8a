"""the NeuralNDCG loss alone at the bench shape (256 slates x 240 items, tau 1, k None), 5 calls (rocprofv3 passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd.losses import FusedLoss
B, L = int(os.environ.get("NB", "256")), int(os.environ.get("NL", "240"))
g = torch.Generator().manual_seed(1)
s = torch.randn(B, L, generator=g).cuda()
y = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * L, replacement=True, generator=g).view(B, L).float().cuda()
fl = FusedLoss("neuralNDCG", B, L, "cuda", temperature=1.0, k=None)
for _ in range(5):
    fl.run(s, y, float(B))
torch.cuda.synchronize()
