"""NeuralNDCG kernel paths A/B: path 0 = block-resident kernels (round 3), 1 = general L2-streaming kernels (the round-2 row-resident
register kernels were path 2 until they were removed: profiles/r03_neuralndcg_ab.md).  Prints loss / gradient agreement between the
paths and us per call (HIP events); run per library variant with LTRX_LIB_PATH=tools/lab/ab/libltrx_TAG.so."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allrank_amd import losses as E
DEV = "cuda"


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


for (B, L) in [(256, 240), (256, 200), (256, 100)]:
    g = torch.Generator().manual_seed(L)
    s = torch.randn(B, L, generator=g).to(DEV)
    y = torch.multinomial(torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01]), B * L, replacement=True, generator=g).view(B, L).float().to(DEV)
    y[1, L // 2:] = -1
    y[2] = 0
    rec, outs = dict(B=B, L=L), {}
    for path in (0, 1):
        sp = s.clone().requires_grad_(True)
        with E.neural_kernel_path(path):
            l = E.neuralNDCG(sp, y, temperature=1.0, k=None)
            l.backward()
            outs[path] = (l.item(), sp.grad.clone())
            if path == 0:
                rec["path%d_us" % path] = round(min(timeit(lambda: E.neuralNDCG(sp, y, temperature=1.0, k=None)) for _ in range(3)), 1)
    for path in (1,):
        rec["loss_diff_0v%d" % path] = abs(outs[0][0] - outs[path][0])
        rec["grad_maxdiff_0v%d_rel" % path] = float((outs[0][1] - outs[path][1]).abs().max() / outs[path][1].abs().max())
    rec["loss"] = outs[0][0]
    print(json.dumps(rec), flush=True)
