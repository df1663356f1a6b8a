"""cycle stamps of one workgroup of the block-resident NeuralNDCG forward (lab build with -DLTRX_NEURAL_STAMP=<block>): where the ~12 k
cycles of a Sinkhorn step go.  usage (GPU box): LTRX_LIB_PATH=tools/lab/ab/libltrx_nstamp.so python tools/lab/neural_stamps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from allrank_amd import _lib as LB
from allrank_amd.losses import FusedLoss
raw = ctypes.CDLL(LB.LIB_PATH)
B, L = 256, 240
g = torch.Generator().manual_seed(1)
s = torch.randn(B, L, generator=g).cuda()
y = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * L, replacement=True, generator=g).view(B, L).float().cuda()
fl = FusedLoss("neuralNDCG", B, L, "cuda", temperature=1.0, k=None)
for _ in range(3):
    fl.run(s, y, float(B))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (12 * 8 * 8))()
raw.ltrx_debug_neural_stamps.argtypes = [ctypes.c_void_p]
assert raw.ltrx_debug_neural_stamps(buf) == 0
st = [[[buf[(w * 8 + k) * 8 + p] for p in range(8)] for k in range(8)] for w in range(12)]
print("per Sinkhorn step (cycles): [col sums+publish] [barrier] [col total+scale] [row sums+publish] [barrier] [row total+scale] | step total")
for w in (0, 1, 5, 11):
    print("wave %d" % w)
    for k in range(1, 7):
        a = st[w][k]
        nxt = st[w][k + 1][0]
        print("   it %2d: %6d %6d %6d %6d %6d %6d | %6d" % (20 + k, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5], nxt - a[0]))
