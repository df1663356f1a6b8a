"""cycle stamps of one workgroup of the resident attention dK/dV kernel (lab build -DLTRX_MHA_STAMP=<block> -DLTRX_MHA_STAMP_DKDV).
usage (GPU box): LTRX_LIB_PATH=tools/lab/ab/libltrx_dstamp.so python tools/lab/mha_dkdv_stamps.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from allrank_amd import _lib as LB
lib = LB.lib()
raw = ctypes.CDLL(LB.LIB_PATH)
B, L, h, dk = 256, 240, 8, 64
d = h * dk
qkv = torch.randn(B * L, 3 * d, device="cuda")
do = torch.randn(B * L, d, device="cuda")
o = torch.empty(B * L, d, device="cuda")
lse = torch.empty(B, h, L, device="cuda")
dqkv = torch.empty(B * L, 3 * d, device="cuda")
mask = torch.zeros(B, L, dtype=torch.uint8, device="cuda")
ws = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, L, h, dk, 1), 64), dtype=torch.uint8, device="cuda")
P = LB.ptr
LB.check(lib.ltrx_mha_fwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), B, L, h, dk, 3 * d, P(o), d, P(lse), 0.0, 0,
                          None, None, None, 1, None), "fwd")
for _ in range(3):
    LB.check(lib.ltrx_mha_bwd(P(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, P(mask), P(o), P(do), P(lse), B, L, h, dk, 3 * d, d,
                              P(dqkv), dqkv.data_ptr() + 4 * d, dqkv.data_ptr() + 8 * d, 3 * d, 0.0, 0, None, None, None, 1, P(ws), None), "bwd")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 40 * 8))()
raw.ltrx_debug_mha_stamps.argtypes = [ctypes.c_void_p]
assert raw.ltrx_debug_mha_stamps(buf) == 0
s = [[[buf[(w * 40 + k) * 8 + p] for p in range(8)] for k in range(40)] for w in range(8)]
t0 = min(s[w][32][0] for w in range(8))
for w in (0, 3, 4, 7):
    print("wave %d: start +%d, loop +%d, loop end +%d, end +%d cycles" % (w, s[w][32][0] - t0, s[w][32][1] - t0, s[w][33][0] - t0, s[w][34][0] - t0))
    print("   qt: [delta+sstore] [barrier wait] [S,dP mfma] [softmax bwd] [dS stores+next loads] [dV] [dK]   (cycles)   | tile start, tile total")
    for k in range(8):
        a = s[w][k]
        nxt = s[w][k + 1][0] if k < 7 else s[w][33][0]
        print("   %2d: %6d %6d %6d %6d %6d %6d %6d   | +%d  %d" % (k, a[1] - a[0], a[2] - a[1], a[3] - a[2], a[4] - a[3], a[5] - a[4], a[6] - a[5],
                                                                a[7] - a[6], a[0] - t0, nxt - a[0]))
