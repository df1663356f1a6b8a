"""cycle stamps of one workgroup of the slate-resident FC + ListNet kernel (lab build with -DLTRX_FC_STAMP=<block>): where the time of
a slate goes.  usage (GPU box): LTRX_LIB_PATH=tools/lab/ab/libltrx_fcstamp.so python tools/lab/fc_stamps.py [slates]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from allrank_amd import _lib as LB
from tools.fcstep_check import build, batch
from allrank_amd.engine import FusedTrainer
raw = ctypes.CDLL(LB.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L, F, H = 240, 136, 96
cfg = dict(n_features=F, fc_sizes=[H], fc_activation=None, fc_input_norm=False, N=0, d_ff=0, h=1, output_activation=None)
x, y = batch(np.random.default_rng(7), B, L, F, [])
xt, yt = torch.tensor(x, device="cuda"), torch.tensor(y, device="cuda")
m, _ = build(cfg, 3)
ft = FusedTrainer(m, "listNet", {}, B, L, lr=1e-3, use_graph=False)
for _ in range(5):
    ft.step(xt, yt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (12 * 10 * 10))()
raw.ltrx_debug_fc_stamps.argtypes = [ctypes.c_void_p]
assert raw.ltrx_debug_fc_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64).reshape(12, 10, 10)
nit = min(10, (B + 255) // 256)
print("B=%d: per slate (cycles): [stage: loads+split+store] [barrier1] [forward+partials] [barrier2] [listnet (wave 0)] [barrier3] [backward] [barrier4] | slate total" % B)
for w in (0, 1, 6, 11):
    print("wave %d" % w)
    for it in range(nit):
        a = st[w][it]
        print("   slate %d: " % it + " ".join("%6d" % (a[k + 1] - a[k]) for k in range(8)) + " | %6d" % (a[8] - a[0]))
