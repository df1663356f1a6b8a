"""cycle stamps of one workgroup of the four-wave x 64-query attention forward (lab build with -DLTRX_MHA_STAMP=<block>): where the
cycles of a tile go.  usage (GPU box): LTRX_LIB_PATH=tools/lab/ab/libltrx_stamp.so python tools/lab/mha_w64_stamps.py [B L]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from allrank_amd import _lib as LB
lib = LB.lib()
raw = ctypes.CDLL(LB.LIB_PATH)
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 240)
h, dk = 8, 64
d = h * dk
qkv = torch.randn(B * L, 3 * d, device="cuda")
o = torch.empty(B * L, d, device="cuda")
lse = torch.empty(B, h, L, device="cuda")
mask = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
P = LB.ptr
for _ in range(3):
    LB.check(lib.ltrx_mha_fwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), B, L, h, dk, 3 * d, P(o), d, P(lse), 0.0, 0,
                              None, None, None, 1, None), "fwd")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 40 * 8))()
raw.ltrx_debug_mha_stamps.argtypes = [ctypes.c_void_p]
assert raw.ltrx_debug_mha_stamps(buf) == 0
s = [[[buf[(w * 40 + k) * 8 + p] for p in range(8)] for k in range(40)] for w in range(8)]
nw = 4
t0 = min(s[w][32][0] for w in range(nw))
nkt = (L + 31) // 32
for w in range(nw):
    print("wave %d: start +%d, prologue done +%d, loop done +%d, end +%d cycles" % (w, s[w][32][0] - t0, s[w][32][1] - t0, s[w][33][0] - t0, s[w][34][0] - t0))
    print("   t: [ops 0-7: softmax head] [8-23: P V + exp A] [24-39: S + exp B] [40-47: staging] [tail: copies / rescale] [barrier]   | iteration start")
    for k in range(nkt):
        a = s[w][k]
        print("   %2d: %6d %6d %6d %6d %6d %6d   | +%d" % (k, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5], a[0] - t0))
