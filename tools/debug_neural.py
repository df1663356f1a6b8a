import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allrank_amd import _lib as L
lib = L.lib()
def a64(x): return (x + 63) & ~63
for big, tau in ((-1e10, 1.0), (-1e10, 0.01)):
    yp = torch.tensor([[0.5, big]], device="cuda"); yt = torch.tensor([[1.0, 0.0]], device="cuda")
    B, SL, mi = 1, 2, 50
    ws = torch.zeros(lib.ltrx_neuralndcg_workspace_bytes(B, SL, mi), dtype=torch.uint8, device="cuda")
    idcg = torch.empty(B, device="cuda"); cnt = torch.empty(1, device="cuda"); it = torch.empty(1, dtype=torch.int32, device="cuda")
    loss = torch.empty(1, device="cuda"); per = torch.empty(B, device="cuda")
    st = L.stream_of(yp)
    L.check(lib.ltrx_neuralndcg_prepare(L.ptr(yt), B, SL, -1.0, 0, 1, L.ptr(idcg), L.ptr(cnt), L.ptr(ws), st), "p")
    L.check(lib.ltrx_neuralndcg_fwd_bwd(L.ptr(yp), L.ptr(yt), L.ptr(idcg), L.ptr(cnt), B, SL, -1.0, tau, 1, 0, 0, mi, 1e-6, L.ptr(loss), L.ptr(per), None, L.ptr(it), 0, L.ptr(ws), st), "f")
    torch.cuda.synchronize()
    o_per = 0; o_res = a64(B*4); o_t = o_res + a64(B*mi*4); o_cn = o_t + 64; o_rn = o_cn + a64(B*mi*SL*4); o_S = o_rn + a64(B*mi*SL*4)
    f = ws.view(torch.float32)
    print("big", big, "tau", tau, "loss", loss.item(), "per", per.tolist(), "idcg", idcg.tolist(), "cnt", cnt.item(), "T", it.item())
    print(" res", f[o_res//4:o_res//4+5].tolist())
    print(" cn", f[o_cn//4:o_cn//4+6].tolist(), " rn", f[o_rn//4:o_rn//4+6].tolist())
    print(" S", f[o_S//4:o_S//4+4].tolist())
