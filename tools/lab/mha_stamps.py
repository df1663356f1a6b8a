"""cycle stamps of one workgroup of the resident attention forward (lab build with -DLTRX_MHA_STAMP=<block>): where a (slate, head)'s
~24 us go.  usage (GPU box): LTRX_LIB_PATH=tools/lab/ab/libltrx_stamp.so python tools/lab/mha_stamps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from allrank_amd import _lib as LB
lib = LB.lib()
raw = ctypes.CDLL(LB.LIB_PATH)
B, L, h, dk = 256, 240, 8, 64
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
t0 = min(s[w][32][0] for w in range(8))
for w in (0, 3, 4, 7):
    print("wave %d: start +%d, end(before stores) +%d, end +%d cycles" % (w, s[w][32][0] - t0, s[w][33][0] - t0, s[w][34][0] - t0))
    print("   kt: [top->staged] [barrier wait] [S mfma] [softmax] [PV]   (cycles)")
    for k in range(8):
        a = s[w][k]
        print("   %2d: %6d %6d %6d %6d %6d   | tile start +%d" % (k, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[0] - t0))
