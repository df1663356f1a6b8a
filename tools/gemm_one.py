import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import _lib as LB
lib = LB.lib()
m, n, k = int(os.environ.get('GM', '15360')), int(os.environ.get('GN', '2048')), int(os.environ.get('GK', '512'))
A = torch.randn(m, k, device="cuda"); B = torch.randn(n, k, device="cuda") / k ** 0.5; bias = torch.randn(n, device="cuda")
C = torch.empty(m, n, device="cuda")
if os.environ.get("GZERO"):           # DVFS probe: all-zero operands draw less power -> the same kernel at a higher clock
    A.zero_(); B.zero_()
GV = int(os.environ.get("GV", "0"))          # tile argument of the calls (0 = auto)
PREC = int(os.environ.get("GPREC", "0"))       # 0 three products, 1 strict, 2 plain bf16
for _ in range(5):
    LB.check(lib.ltrx_gemm_nt(LB.ptr(A), k, LB.ptr(B), k, None, LB.ptr(C), n, m, n, k, LB.ptr(bias), 0, None, 0, 0.0, 0, None, PREC, GV, None), "nt")
if os.environ.get("GONLY") == "nt":
    torch.cuda.synchronize()
    sys.exit(0)
dY = torch.randn(m, n, device="cuda"); X = torch.randn(m, k, device="cuda")
if os.environ.get("GZERO"):
    dY.zero_(); X.zero_()
gW = torch.empty(n, k, device="cuda"); gb = torch.empty(n, device="cuda")
ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(m, n, k), 64), dtype=torch.uint8, device="cuda")
for _ in range(5):
    LB.check(lib.ltrx_gemm_tn(LB.ptr(dY), n, LB.ptr(X), k, LB.ptr(gW), LB.ptr(gb), m, n, k, PREC, GV if GV == 1 else 0, LB.ptr(ws), None), "tn")
torch.cuda.synchronize()
