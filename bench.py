#!/usr/bin/env python
"""bench.py -- slate-items/s of the listwise-LTR training step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (allrank/training/train_utils.py:18-29: mask -> model forward -> listwise loss
-> backward -> Adam step -> zero_grad) over one batch of synthetic WEB30K-shaped slates that is already resident in
HBM.  Default workload = BASELINE.json configs[2], the one the north_star target is quoted on:
F=136, slate_len 240, FCModel[512] -> 2-layer self-attention (d_model 512, h 8, d_ff 2048) -> ApproxNDCG, Adam 1e-3,
dense slates, 256 slates per GPU by default (the reference's batch_size 64 point is measured too and reported as
`value_at_64_slates_per_gpu`; reproducibility/configs/*: batch_size 64, slate_length 240).
`--workload fc_listnet` runs configs[1] (FCModel[96] + ListNet).  Weak scaling: every rank gets its own
`--slates-per-gpu` slates, losses are normalised by the global batch and gradients summed over RCCL.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant hand-written kernel of the step (timed live with
HIP events on the launch stream, in a side pass after the timed region; FC-only workloads: the whole step against the
HBM roof; NeuralNDCG workloads add `roofline_loss_kernels` against the fp32 vector peak); `cpu_baseline` is the reference step
restated with the torch CPU operators the reference itself calls (oracle/torch_port.py, kind "port", pinned to the numpy oracle;
losses it does not cover: the numpy oracle) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# dmabuf IPC (RCCL peer buffers across the ranks of a node) -- whoever launched this rank (torchrun, the driver, _self_spawn):
# set before the HIP runtime comes up; already exported on the GPU boxes, a no-op there
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: v_mfma_f32_32x32x16_bf16, dense (2:1-sparse marketing figure excluded)
PEAK_HBM_GBPS = 8000.0

WORKLOADS = {
    "attn_approxndcg": dict(desc="WEB30K-synth F=136 L=240, fc[512] + 2x self-attention(d512,h8,d_ff2048) + ApproxNDCG (BASELINE configs[2])",
                            n_features=136, fc_sizes=[512], N=2, h=8, d_ff=2048, loss="approxNDCGLoss"),
    "fc_listnet": dict(desc="WEB30K-synth F=136 L=240, FCModel[96] + ListNet (BASELINE configs[1])",
                       n_features=136, fc_sizes=[96], N=0, h=1, d_ff=0, loss="listNet"),
    "attn_neuralndcg": dict(desc="WEB30K-synth F=136 L=240, fc[512] + 2x self-attention(d512,h8,d_ff2048) + NeuralNDCG tau=1 (BASELINE configs[3])",
                            n_features=136, fc_sizes=[512], N=2, h=8, d_ff=2048, loss="neuralNDCG",
                            loss_args=dict(temperature=1.0, powered_relevancies=True, k=None, stochastic=False)),
    "attn_lambdarank": dict(desc="WEB30K-synth F=136 L=240, fc[512] + 2x self-attention(d512,h8,d_ff2048) + lambdaLoss(lambdaRank_scheme) (BASELINE configs[3])",
                            n_features=136, fc_sizes=[512], N=2, h=8, d_ff=2048, loss="lambdaLoss",
                            loss_args=dict(weighing_scheme="lambdaRank_scheme", k=None, mu=10.0, sigma=1.0)),
    "attn1024_listmle": dict(desc="synthetic F=1024 L=1024, fc[512] + 2x self-attention(d512,h8,d_ff2048) + ListMLE (BASELINE configs[4])",
                             n_features=1024, fc_sizes=[512], N=2, h=8, d_ff=2048, loss="listMLE", slate_len=1024, slates=16),
}


def train_flops_per_item(w, L):
    """3*[2Fd + N(8d^2 + 4Ld + 4 d d_ff) + 2d] - 2Fd"""
    F, d, N, dff = w["n_features"], w["fc_sizes"][-1], w["N"], w["d_ff"]
    fwd = 2 * F * d + N * (8 * d * d + 4 * L * d + 4 * d * dff) + 2 * d
    return 3 * fwd - 2 * F * d


def synth_batch(n_slates, L, F, seed, device, ragged=False):
    """SURVEY.md §8d recipe: x ~ N(0,1), labels ~ Cat(.52,.32,.13,.02,.01), 3% of slates all-zero; dense slates, or
    (ragged) WEB30K-like lengths n_valid ~ clip(round(lognormal(ln 100, 0.6)), 1, L) padded like FixLength._pad."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n_slates, L, F), generator=g, dtype=torch.float32)
    probs = torch.tensor([0.52, 0.32, 0.13, 0.02, 0.01])
    y = torch.multinomial(probs, n_slates * L, replacement=True, generator=g).view(n_slates, L).float()
    zero = torch.rand(n_slates, generator=g) < 0.03
    y[zero] = 0.0
    idx = torch.arange(L).expand(n_slates, L).contiguous()
    if ragged:
        nv = torch.exp(torch.randn(n_slates, generator=g) * 0.6 + np.log(100.0)).round().clamp(1, L).long()
        pad = torch.arange(L)[None, :] >= nv[:, None]
        y[pad] = -1.0
        x[pad] = 0.0
        idx = idx.clone()
        idx[pad] = -1
    return x.to(device), y.to(device), idx.to(device)


def build_model(w, device, dropout=0.0):
    from allrank_amd.model import make_model
    tr = dict(N=w["N"], d_ff=w["d_ff"], h=w["h"], positional_encoding=None, dropout=dropout) if w["N"] else None
    fc = dict(sizes=list(w["fc_sizes"]), input_norm=False, activation=None, dropout=0.0)
    torch.manual_seed(42)                      # allrank/main.py:36
    return make_model(fc, tr, dict(d_output=1, output_activation=None), w["n_features"]).to(device)


def time_kernels(w, B, L, device):
    """side pass: time the hand-written kernels of one step with HIP events on the launch stream."""
    from allrank_amd import ops, losses as E
    res = {}

    def ev(fn, iters=10):
        fn()
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters

    d, h = w["fc_sizes"][-1], w["h"]
    from allrank_amd import _lib as LB
    lib = LB.lib()
    if w["N"]:
        # the dominant kernel of the step: the split-bf16 NT GEMM, timed at the FFN-1 shape [B*L, d] x [d_ff, d]^T
        Mrows, Nn, Kk = B * L, w["d_ff"], d
        A_ = torch.randn(Mrows, Kk, device=device)
        W_ = torch.randn(Nn, Kk, device=device) / Kk ** 0.5
        b_ = torch.randn(Nn, device=device)
        C_ = torch.empty(Mrows, Nn, device=device)
        st = LB.stream_of(A_)
        t_g = ev(lambda: LB.check(lib.ltrx_gemm_nt(LB.ptr(A_), Kk, LB.ptr(W_), Kk, None, LB.ptr(C_), Nn, Mrows, Nn, Kk, LB.ptr(b_), 1,
                                                   None, 0, 0.0, 0, None, 0, 0, st), "gemm_nt"))
        n_nt = 1 + 8 * w["N"] - 1          # fwd: fc + 4/layer; dgrad: 4/layer  (head GEMV is a separate kernel)
        _t256 = ((Mrows + 255) // 256) * (Nn // 256)
        big = (Nn % 256 == 0 and Kk % 32 == 0 and (_t256 >= 360 or 168 <= _t256 <= 256))   # ltrx_gemm.hip dispatch
        res["%s @FFN1" % ("ltrx_gemm_nt256_kernel" if big else "ltrx_gemm_nt_kernel<2,128,32>")] = dict(
            sec=t_g, flops=2.0 * Mrows * Nn * Kk, launches_per_step=n_nt, shape=[Mrows, Nn, Kk])
        del A_, W_, b_, C_
    y = torch.zeros(B, L, device=device)
    s = torch.randn(B, L, device=device, requires_grad=True)
    lossfn = getattr(E, w["loss"])
    largs = w.get("loss_args", {})
    res["loss_fwd_bwd"] = dict(sec=ev(lambda: lossfn(s, y, **largs)), launches_per_step=1)
    if w["N"]:
        dk = d // h
        qkv = torch.randn(B, L, 3 * d, device=device)
        mask = torch.zeros(B, L, dtype=torch.bool, device=device)
        fl = 4.0 * B * h * L * L * dk
        go = torch.randn(B, L, d, device=device)
        o_ = torch.empty(B, L, d, device=device)
        lse_ = torch.empty(B, h, L, device=device)
        dqkv = torch.empty(B, L, 3 * d, device=device)
        m8 = mask.to(torch.uint8)
        ws_ = torch.empty(max(lib.ltrx_mha_bwd_workspace_bytes(B, L, h, dk, 1), 64), dtype=torch.uint8, device=device)
        st2 = LB.stream_of(qkv)
        t_f = ev(lambda: LB.check(lib.ltrx_mha_fwd(LB.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, LB.ptr(m8), B, L, h, dk,
                                                   3 * d, LB.ptr(o_), d, LB.ptr(lse_), 0.0, 0, None, None, None, 1, st2), "mha_fwd"))
        res["ltrx_mha_fwd (res split-bf16)"] = dict(sec=t_f, flops=fl, launches_per_step=w["N"])
        t_b = ev(lambda: LB.check(lib.ltrx_mha_bwd(LB.ptr(qkv), qkv.data_ptr() + 4 * d, qkv.data_ptr() + 8 * d, LB.ptr(m8), LB.ptr(o_),
                                                   LB.ptr(go), LB.ptr(lse_), B, L, h, dk, 3 * d, d, LB.ptr(dqkv), dqkv.data_ptr() + 4 * d,
                                                   dqkv.data_ptr() + 8 * d, 3 * d, 0.0, 0, None, None, None, 1, LB.ptr(ws_), st2), "mha_bwd"))
        res["ltrx_mha_bwd (dq+dkdv, res split-bf16)"] = dict(sec=t_b, flops=2.5 * fl, launches_per_step=w["N"])
        x = torch.randn(B * L, d, device=device)
        r = torch.randn(B * L, d, device=device)
        a = torch.ones(d, device=device)
        bb = torch.zeros(d, device=device)
        with torch.no_grad():               # (the step's LayerNorm reads ONE stream: the residual sum is the closing projection's epilogue)
            t_ln = ev(lambda: ops.layer_norm(x, a, bb))
        res["ltrx_layernorm_fwd_kernel"] = dict(sec=t_ln, bytes=2.0 * B * L * d * 4, launches_per_step=2 * w["N"] + 1)
    return res


def write_synth_libsvm(path, lengths, F, seed):
    """A WEB30K-shaped libsvm text file, written with vectorised numpy (one byte matrix, no per-line Python): line
    ``<label> qid:<6 digits> 1:0.dddd 2:0.dddd ... F:0.dddd``; labels ~ Cat(.52,.32,.13,.02,.01), features uniform on
    the 4-decimal grid of [0, 1) (WEB30K's features are min-max scaled, normalize_features.py), ``lengths[q]`` lines per query.
    Returns (X f32[n, F], y f32[n], qid i64[n]) -- the arrays the text encodes exactly (both parsers read them back bit for bit)."""
    rng = np.random.default_rng(seed)
    lengths = np.asarray(lengths, dtype=np.int64)
    n = int(lengths.sum())
    vals = rng.integers(0, 10000, size=(n, F), dtype=np.int32)
    y = rng.choice(5, size=n, p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.int32)
    qid = np.repeat(np.arange(len(lengths), dtype=np.int64) + 100000, lengths)
    head = b"0 qid:000000"
    cols = [b" %d:0.0000" % (k + 1) for k in range(F)]
    row = np.frombuffer(head + b"".join(cols) + b"\n", dtype=np.uint8)
    t4 = np.arange(10000)
    table = (48 + np.stack([t4 // 1000, t4 // 100 % 10, t4 // 10 % 10, t4 % 10], 1)).astype(np.uint8)       # value -> its four ASCII digits
    groups, off, k = [], len(head), 0
    while k < F:                                            # columns whose index has the same number of digits form one [m, g, width] block
        width = len(cols[k])
        g = 0
        while k + g < F and len(cols[k + g]) == width:
            g += 1
        groups.append((k, g, width, off))
        off += g * width
        k += g
    chunk = 16384                                           # one reused 24-MB byte matrix: the template row, then the digits
    buf = np.broadcast_to(row, (chunk, row.size)).copy()
    with open(path, "wb") as fh:
        for a in range(0, n, chunk):
            m = min(chunk, n - a)
            text = buf[:m]
            text[:, 0] = 48 + y[a:a + m]
            for j in range(6):
                text[:, 6 + j] = 48 + (qid[a:a + m] // 10 ** (5 - j)) % 10
            digits = table[vals[a:a + m]]                   # [m, F, 4]
            for (k, g, width, off) in groups:
                text[:, off:off + g * width].reshape(m, g, width)[:, :, width - 4:] = digits[:, k:k + g]
            text.tofile(fh)
    return (vals.astype(np.float64) / 10000.0).astype(np.float32), y.astype(np.float32), qid


def _web30k_lengths(n_queries, L, seed, dense=False):
    if dense:
        return np.full(n_queries, L, dtype=np.int64)
    rng = np.random.default_rng(seed)                      # clip(round(lognormal(ln 100, 0.6)), 1, L)
    return np.clip(np.round(np.exp(rng.standard_normal(n_queries) * 0.6 + np.log(100.0))), 1, L).astype(np.int64)


def end_to_end_main(w, B, L, device, workdir, dense, n_queries, epochs, host_loader=None, gemm="split_bf16"):
    """What an UNMODIFIED allrank/main.py does after ``allrank_amd.install(fit=True)`` (main.py:36-102), timed end to end on a synthetic
    WEB30K-shaped libsvm job: seeds -> ``load_libsvm_dataset`` (files parsed on the GPU, slates resident in HBM) -> ``create_data_loaders``
    (DeviceLoader) -> make_model -> Adam -> ``fit`` (explicit step; validation pass; metrics).  Reported: slots and valid items per
    second over the TRAINING pass of the last epoch (wall clock inside fit(), loader + step + train metrics), and the whole epoch.
    ``host_loader``: a module with the reference loader's interface (oracle/loader_oracle.py, handed in by the cpu_baseline leg
    only) -> the same fit() fed by torch DataLoader + FixLength on the host, i.e. what main.py gets WITHOUT the loader rebinding."""
    import types
    from functools import partial
    from allrank_amd import data as ED, fit as EF, losses as E
    lens_tr = _web30k_lengths(n_queries, L, 11, dense)
    lens_va = _web30k_lengths(max(B, n_queries // 8), L, 12, dense)
    os.makedirs(workdir, exist_ok=True)
    t0 = time.perf_counter()
    for role, lens, sd in (("train", lens_tr, 1), ("vali", lens_va, 2)):
        f = os.path.join(workdir, "%s.txt" % role)
        if not os.path.exists(f):
            write_synth_libsvm(f, lens, w["n_features"], sd)
    t_write = time.perf_counter() - t0
    torch.manual_seed(42)                                   # main.py:36-38
    torch.cuda.manual_seed_all(42)
    np.random.seed(42)
    t0 = time.perf_counter()
    if host_loader is None:
        tr_ds, va_ds = ED.load_libsvm_dataset(workdir, L, "vali", device=device)             # main.py:57-61
        tr, va = ED.DeviceLoader(tr_ds, B, shuffle=True), ED.DeviceLoader(va_ds, B, shuffle=False)     # main.py:67-68, one GPU
    else:
        tr_ds, va_ds = host_loader.load_libsvm_dataset(workdir, L, "vali")
        tr, va = host_loader.create_data_loaders(tr_ds, va_ds, num_workers=0, batch_size=B)  # (in-process: no worker start-up per pass,
        #                                                                                       the faster setting for a short epoch)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    model = build_model(w, device, 0.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss_func = partial(getattr(E, w["loss"]), **w.get("loss_args", {}))
    config = types.SimpleNamespace(metrics={"ndcg": [5]}, val_metric="ndcg_5", detect_anomaly=False)
    res = EF.fit(epochs=epochs, model=model, loss_func=loss_func, optimizer=opt, scheduler=None, train_dl=tr, valid_dl=va, config=config,
                 gradient_clipping_norm=None, early_stopping_patience=100, device=device, output_dir=workdir, tensorboard_output_path=None,
                 gemm=gemm)
    log = EF.last_run["epoch_log"]
    last = log[-1]
    valid = int(np.minimum(lens_tr, L).sum())
    steady = log[1:] if len(log) > 1 else log
    tsum = sum(e["train_s"] for e in steady)
    return {"slots_per_s": round(sum(e["slots"] for e in steady) / tsum, 1), "valid_items_per_s": round(valid * len(steady) / tsum, 1),
            "valid_fraction": round(valid / float(last["slots"]), 4), "train_pass_s": round(last["train_s"], 4),
            "validation_pass_s": round(last["val_s"], 4), "first_epoch_train_pass_s": round(log[0]["train_s"], 4),
            "epochs": len(log), "steps_per_epoch": len(tr), "queries": int(n_queries), "engine": EF.last_run["engine"],
            "variable_length": bool(EF.last_run["compact"]), "file_mb": round(os.path.getsize(os.path.join(workdir, "train.txt")) / 1e6, 1),
            "write_file_s": round(t_write, 2), "load_and_parse_s": round(t_load, 2),
            "val_ndcg_5": float(res["val_metrics"]["ndcg_5"]), "loader": "allrank_amd.data.DeviceLoader (HBM-resident)" if host_loader is None
            else "torch DataLoader + FixLength on the host (the reference's loader, restated: oracle/loader_oracle.py), num_workers=0"}


def cpu_baseline(w, L, seconds_budget=20.0):
    """the reference training step on this box's host cores, on a bounded sample of the same workload.  For the workloads
    whose loss it covers this is oracle/torch_port.py -- the same computation stated with the torch CPU operators the
    reference itself calls (kind "port", torch threads = all cores); otherwise the numpy oracle (oracle/model_oracle.py)."""
    from oracle import ltr_oracle as O, model_oracle as M, torch_port as TP
    cfg = dict(n_features=w["n_features"], fc_sizes=list(w["fc_sizes"]), fc_activation=None, fc_input_norm=False, N=w["N"],
               d_ff=w["d_ff"], h=w["h"], output_activation=None)
    Bs = 16 if w["N"] else 64
    rng = np.random.default_rng(0)
    x = rng.standard_normal((Bs, L, w["n_features"])).astype(np.float32)
    y = rng.choice(5, size=(Bs, L), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
    params = M.init_params(cfg, seed=0)
    used = os.cpu_count()
    if w["loss"] in TP.LOSSES:
        stp = TP.Stepper(params, cfg, w["loss"], lr=1e-3)
        xt, yt = torch.tensor(x), torch.tensor(y)
        step = lambda: stp.step(xt, yt)  # noqa: E731
        # torch's intra-op pool does not scale to every core of a big host (256 threads on this workload run 200x slower
        # than 8): take the best of a few pool sizes, each timed on a couple of steps, and report the one used
        best = None
        for nt in sorted({min(8, used), min(16, used), min(32, used), min(64, used)}):
            torch.set_num_threads(nt)
            step()
            t0 = time.perf_counter()
            k = 0
            while k < 8 and time.perf_counter() - t0 < 2.5:
                step()
                k += 1
            rate = k / (time.perf_counter() - t0)
            if best is None or rate > best[0]:
                best = (rate, nt)
        used = best[1]
        torch.set_num_threads(used)
        seconds_budget = 12.0
        what = "torch CPU port of the reference step (oracle/torch_port.py), fp32, %d torch threads (best of 8/16/32/64)" % used
    else:
        opt = M.Adam(params, lr=1e-3)
        la = dict(w.get("loss_args", {}))
        la.pop("stochastic", None)
        lossfn = {"neuralNDCG": lambda s, t: O.neuralndcg(s, t, **la), "lambdaLoss": lambda s, t: O.lambdaloss(s, t, **la),
                  "listMLE": lambda s, t: O.listmle(s, t, np.arange(L))}[w["loss"]]
        step = lambda: M.train_step(params, cfg, opt, x, y, lossfn)  # noqa: E731
        what = "numpy oracle (oracle/model_oracle.py), fp32, BLAS on all cores"
    step()        # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 200:
            break
    out = dict(value=round(n * Bs * L / el, 1), unit="slate-items/s", cores=used, host_cores=os.cpu_count(), kind="port",
               sample="%d training steps of %d slates x %d items, %s" % (n, Bs, L, what))
    # the REAL reference (allegro/allRank's own loss_batch on CPU torch) cannot run on the GPU box; its timing on the build
    # container's cores is committed next to the script that produced it (tests/golden/make_ref_cpu_timing.py)
    try:
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_cpu_timing.json")))
        if w["N"] == 2 and w["d_ff"] == 2048 and w["n_features"] == 136 and L == 240 and w["loss"] == "approxNDCGLoss":
            out["reference_build_box"] = dict(value=ref["value"], unit=ref["unit"], cores=ref["cores"], cpu=ref.get("cpu"),
                                              kind="reference (allrank loss_batch on CPU torch %s, build container)" % ref.get("torch"),
                                              points=ref["points"])
    except Exception:
        pass
    # round 4: ONE metered run with the reference staged in an ignored scratch directory on the GPU box (never committed; removed
    # after the run): its own loss_batch on that box's host cores, and on the MI355X itself through stock PyTorch-ROCm.  Static
    # figures with their provenance (profiles/r04_reference_on_gpu_box.md), printed beside the live numbers of this run.
    try:
        if w["N"] == 2 and w["d_ff"] == 2048 and w["n_features"] == 136 and L == 240 and w["loss"] == "approxNDCGLoss":
            rc = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_cpu_timing_gpubox.json")))
            out["reference_gpu_box_cpu"] = dict(value=rc["value"], unit=rc["unit"], host_cores=rc["cores"], cpu=rc.get("cpu"),
                                                kind="reference (allrank loss_batch on CPU torch %s, GPU box host, round-4 staged run)" % rc.get("torch"),
                                                points=rc["points"])
            rg = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_gpu_timing_gpubox.json")))
            out["reference_on_this_gpu"] = dict(value=rg["value"], unit=rg["unit"],
                                                kind="reference (unmodified allrank loss_batch, stock PyTorch-ROCm %s on the MI355X, round-4 staged run)" % rg.get("torch"),
                                                points=rg["points"])
    except Exception:
        pass
    return out


def reference_loader_leg(w, B, L, device, gemm, e2e):
    """cpu_baseline leg, data side: the reference's host loader (torch DataLoader + FixLength + ToTensor, restated in
    oracle/loader_oracle.py and pinned to the reference's own loaders) on this box's host cores -- (i) its raw rate, slots per second
    of one shuffled epoch at num_workers 0 and 1; (ii) the SAME fit() as `end_to_end_main`, fed by it: what an unmodified main.py
    gets from install(fit=True, data=False)."""
    import shutil
    import tempfile
    from oracle import loader_oracle as LO
    d = tempfile.mkdtemp(prefix="ltrx_e2e_host_")
    rec = {}
    try:
        fed = end_to_end_main(w, B, L, device, d, dense=False, n_queries=6 * B, epochs=2, host_loader=LO, gemm=gemm)
        rec["fit_fed_by_reference_loader"] = fed
        tr_ds, va_ds = LO.load_libsvm_dataset(d, L, "vali")
        for nw in (0, 1):
            tr, _ = LO.create_data_loaders(tr_ds, va_ds, num_workers=nw, batch_size=B)
            t0 = time.perf_counter()
            n = sum(int(xb.shape[0]) for xb, _, _ in tr)
            rec["loader_slots_per_s_num_workers_%d" % nw] = round(n * L / (time.perf_counter() - t0), 1)
        if isinstance(e2e, dict):
            rec["device_loader_over_reference_loader"] = round(e2e["ragged"]["slots_per_s"] / fed["slots_per_s"], 2)
        rec["kind"] = "port (oracle/loader_oracle.py == allrank/data/dataset_loading.py:19-248, pinned bit for bit in tests/test_loader_cpu.py)"
        rec["host_cores"] = os.cpu_count()
    except Exception as e:  # noqa: BLE001
        rec["error"] = repr(e)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return rec


def measure_hbm_traffic(M, N, K, timeout_s=90, script="gemm_one.py", env_extra=None, kernels=("nt256",)):
    """HBM-side bytes per launch of the dominant GEMM kernel at the benchmarked shape, from rocprofv3 PMC counters collected
    the way MI355X_MICROARCH.md (HBM section) prescribes: separate --pmc passes (FETCH_SIZE, then WRITE_SIZE; --kernel-trace only
    beside them), counters in KB, FETCH_SIZE doubled on gfx950 (it tallies 128-B requests at 64 B for wide coalesced reads).
    Runs AFTER the timed region, as two child processes executing tools/gemm_one.py (the same kernel, shape and library; 5
    launches, the mean of launches 2..5 is used).  Returns (bytes, detail) or (None, reason)."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "skipped: this run is itself being profiled (no nested rocprofv3)"
    root = os.path.dirname(os.path.abspath(__file__))
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ltrx_pmc_")
        try:
            env = dict(os.environ, GM=str(M), GN=str(N), GK=str(K), GONLY="nt", TMPDIR=d)
            env.update(env_extra or {})
            subprocess.run([exe, "--pmc", counter, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--",
                            sys.executable, os.path.join(root, "tools", script)], cwd=d, env=env, timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, "no counter_collection.csv from the %s pass" % counter
            tot = 0.0
            rows_ = list(csv.DictReader(open(files[0])))
            for kn in kernels:                  # (several kernels: the launches of one step; their per-launch means are summed)
                per = [float(r["Counter_Value"]) for r in rows_ if kn in r["Kernel_Name"] and r["Counter_Name"] == counter]
                if len(per) < 2:
                    return None, "kernel %s not found in the %s pass" % (kn, counter)
                tot += sum(per[1:]) / (len(per) - 1)
            vals[counter] = tot
        except Exception as e:
            return None, "%s pass failed: %r" % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    rd, wr = 2.0 * vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return rd + wr, dict(read_bytes=rd, write_bytes=wr, fetch_size_kb_raw=vals["FETCH_SIZE"], write_size_kb_raw=vals["WRITE_SIZE"],
                         correction="FETCH_SIZE x 2 (gfx950, wide coalesced reads; MI355X_MICROARCH.md HBM section), counters in KB",
                         collection="two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) of tools/%s after the timed region, mean of launches 2..5" % script)


def _self_spawn(n):
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL peer buffers) -- see the environment notes
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def gemm_error_vs_fp64(w, B, L, device, gemm):
    """max |C - C_fp64| / max(|A| |W|^T) of the FFN-1 projection at the benchmarked shape (full launch, so the kernel the
    step uses is the one measured; the fp64 reference covers a 4096-row sample of it)."""
    from allrank_amd import _lib as LB
    lib = LB.lib()
    g = torch.Generator(device=device).manual_seed(1234)
    Mrows, Nn, Kk = B * L, w["d_ff"], w["fc_sizes"][-1]
    A_ = torch.randn(Mrows, Kk, device=device, generator=g)
    W_ = torch.randn(Nn, Kk, device=device, generator=g) / Kk ** 0.5
    b_ = torch.randn(Nn, device=device, generator=g)
    C_ = torch.empty(Mrows, Nn, device=device)
    if gemm == "hipblaslt":
        torch.addmm(b_, A_, W_.t(), out=C_)
    else:
        LB.check(lib.ltrx_gemm_nt(LB.ptr(A_), Kk, LB.ptr(W_), Kk, None, LB.ptr(C_), Nn, Mrows, Nn, Kk, LB.ptr(b_), 0, None, 0, 0.0, 0, None,
                                  {"split_bf16_strict": 1, "bf16": 2}.get(gemm, 0), 0, LB.stream_of(A_)), "gemm_nt")
    rows = torch.linspace(0, Mrows - 1, min(4096, Mrows), device=device).long()
    ref = A_[rows].double() @ W_.double().t() + b_.double()
    scale = (A_[rows].abs().double() @ W_.abs().double().t()).max()
    return float(((C_[rows].double() - ref).abs().max() / scale).item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # >= 200 timed steps after >= 20 warm-up steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="attn_approxndcg", choices=sorted(WORKLOADS))
    ap.add_argument("--slates-per-gpu", type=int, default=256)
    ap.add_argument("--slate-len", type=int, default=240)
    ap.add_argument("--ragged", action="store_true",
                    help="WEB30K-like slate lengths (lognormal, mean ~100 of 240 slots); also reports valid items/s")
    ap.add_argument("--compact", action="store_true",
                    help="with --ragged: variable-length execution (FusedTrainer(compact=True)) -- padded slots are skipped")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-pass", action="store_true", help="profiling runs: skip the 64-slate side measurement")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="transformer dropout (the shipped reference configs train with 0.1-0.4); masks are generated in-kernel")
    ap.add_argument("--gemm", default="split_bf16", choices=["split_bf16", "split_bf16_strict", "hipblaslt", "bf16"],
                    help="dense projections: libltrx fp32-accurate split-bf16 MFMA GEMMs (default), hipBLASLt fp32, or bf16 = the "
                         "one-product throughput mode (GEMMs and attention; outside the 1e-5 parity contract)")
    ap.add_argument("--no-weight-images", action="store_true",
                    help="A/B: split the weight operand of the GEMMs on the fly in every tile instead of once per optimizer step")
    ap.add_argument("--force-dist", action="store_true",
                    help="one GPU: initialise a ONE-rank nccl (= RCCL) process group and run the SHARDED, captured step on it (bucketed "
                         "all-reduce, normaliser all-reduces, hipGraph segments cut at every collective) -- the collective path of an "
                         "8-GPU run exercised on the one GPU a build box has; `comm` is filled in")
    ap.add_argument("--engine", default="fused", choices=["fused", "autograd"],
                    help="fused: explicit hipGraph-captured step (engine.FusedTrainer); autograd: nn.Module + torch autograd/Adam")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL), exactly
        # the command the docstring shows; rank 0 of the child job prints the JSON line
        raise SystemExit(_self_spawn(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)                # (test hook: several ranks may share one GPU with LTRX_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    diag = {"rank_devices": [str(device)], "backend": None, "preflight": None}
    forced = bool(args.force_dist and world == 1)
    if forced:
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LTRX_DIST_BACKEND", "nccl")       # nccl == RCCL on ROCm
        diag["backend"] = backend
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            # first contact with the collective backend, before anything is built on it: one small all-reduce on the device, checked
            # (sum of the ranks), and the device of every rank gathered for the line -- a failure here is reported as a JSON line
            # (`error`, `comm`) instead of a bare traceback, so a failed first RCCL run is diagnosable from the record alone
            t = torch.full((1024,), float(rank + 1), device=device)
            t0 = time.perf_counter()
            dist.all_reduce(t)
            torch.cuda.synchronize()
            ok = bool((t == world * (world + 1) / 2).all().item())
            devs = [None] * world
            dist.all_gather_object(devs, "%s (%s)" % (device, torch.cuda.get_device_name(device)))
            diag.update(rank_devices=devs, preflight=dict(ok=ok, first_allreduce_ms=round((time.perf_counter() - t0) * 1e3, 2)))
            if not ok:
                raise RuntimeError("preflight all-reduce returned a wrong sum")
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(json.dumps({"metric": "slate-items/sec training (WEB30K synth, slate 240)", "value": None, "unit": "slate-items/s",
                                  "n_gpus": world, "error": "collective backend failed at first contact: %r" % (e,), "comm": diag}), flush=True)
            raise

    from allrank_amd import losses as E
    from allrank_amd.engine import Trainer, FusedTrainer
    w = WORKLOADS[args.workload]
    B, L = args.slates_per_gpu, args.slate_len
    if "slate_len" in w and args.slate_len == 240:
        L = w["slate_len"]
    if "slates" in w and args.slates_per_gpu == 256:
        B = w["slates"]
    model = build_model(w, device, args.dropout)
    if args.compact:
        args.ragged = True
    if args.engine == "fused":
        trainer = FusedTrainer(model, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=world, use_graph=True, gemm=args.gemm,
                               compact=args.compact, weight_images=not args.no_weight_images, force_dist=forced)   # Adam 1e-3: approxndcg.json:28-33
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        _lf, _la = getattr(E, w["loss"]), w.get("loss_args", {})
        trainer = Trainer(model, (lambda sc, yt: _lf(sc, yt, **_la)), opt, None, world, None)
    n_batches = 8
    x, y, idx = synth_batch(n_batches * B, L, w["n_features"], 42 + rank, device, ragged=args.ragged)

    lens_host = (y != -1).sum(1).cpu() if args.compact else None     # a data loader knows its slate lengths on the host

    def one_step(i):
        j = (i % n_batches) * B
        if args.compact:
            return trainer.step(x[j:j + B], y[j:j + B], idx[j:j + B], global_batch=B * world, lengths=lens_host[j:j + B])
        return trainer.step(x[j:j + B], y[j:j + B], idx[j:j + B], global_batch=B * world)

    step_fallback = None
    if (world > 1 or forced) and args.engine == "fused" and trainer.use_graph and args.warmup >= 3:
        # The captured sharded step -- hipGraph segments with the collectives between them -- met RCCL for the first time on whatever
        # node runs this.  Captured-or-eager is decided by ALL ranks together, before any rank runs a step in either form (ADVICE r5:
        # a rank that fell back on its own used to restart its warm-up while its peers sat in the all-reduces of theirs): two eager
        # warm-up steps everywhere (a failure there is a real failure and raises), then every rank records its capture WITHOUT
        # executing it (no collective is issued while recording; a failed capture is closed), then one MAX all-reduce of the
        # "could not capture" flags over a CPU (gloo) side group.  If any rank failed, every rank measures the same arithmetic as
        # eager launches and the line says why.
        ctl = dist.new_group(backend="gloo")
        for i in range(2):
            loss = one_step(i)
        ok = trainer.ensure_captured(B * world)
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=ctl)
        if int(flag.item()):
            step_fallback = trainer.capture_fallback or "a peer rank could not capture the sharded step"
            trainer.use_graph = False
            trainer._graphs.clear()
        for i in range(2, args.warmup):
            loss = one_step(i)
    else:
        for i in range(args.warmup):
            loss = one_step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(args.warmup + i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last_loss = float(loss.item())
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the roofline kernel (FFN-1 GEMM) timed where it runs: HIP events around its launches inside eager training steps that
    # follow the timed region (same stream, same neighbours, same clocks as the step; a replayed hipGraph has no place for them)
    probe_us, wgrad_us = None, None
    if args.engine == "fused" and w["N"]:
        use_graph = trainer.use_graph
        trainer.use_graph, trainer.probe, trainer.probe_wgrad = False, [], []
        for i in range(6):
            one_step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        ts = [a.elapsed_time(b) for a, b in trainer.probe[w["N"]:]]          # first probed step dropped
        probe_us = 1e3 * sum(ts) / len(ts)
        # the single longest launch of the step: the grouped weight gradient of an encoder layer (four dW = dY^T X over the same rows)
        tw = [a.elapsed_time(b) for a, b, n_, took_ in trainer.probe_wgrad[w["N"]:] if n_ == 4 and took_]
        wgrad_us = (1e3 * sum(tw) / len(tw)) if tw else None
        trainer.use_graph, trainer.probe, trainer.probe_wgrad = use_graph, None, None

    # multi-rank: how much of the gradient all-reduce is exposed = timed step - the same step with the collective skipped
    comm = None
    if (world > 1 or forced) and args.engine == "fused":
        trainer.comm_enabled = False
        for i in range(2):
            one_step(i)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            one_step(i)
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_nocomm = float(t.item()) / args.steps * 1e3
        ms = dt / args.steps * 1e3
        segs = trainer._graphs.get((float(B * world), True))
        comm = dict(allreduce_bytes_per_step=int(4 * trainer.nflat), buckets=len(trainer._buckets), backend=dist.get_backend(),
                    rank_devices=diag["rank_devices"], preflight=diag["preflight"],
                    captured=bool(trainer.use_graph and segs), graph_segments=(len(segs) if segs else 0),
                    capture_fallback=trainer.capture_fallback, step_fallback=step_fallback,
                    ms_per_step_without_allreduce=round(ms_nocomm, 4), exposed_ms=round(max(ms - ms_nocomm, 0.0), 4),
                    overlapped=bool(ms - ms_nocomm < 0.05 * ms))      # "overlapped" = less than 5 % of the step is exposed

    if rank == 0:
        items = args.steps * B * L * world
        value = items / dt
        fl_item = train_flops_per_item(w, L)
        kern = time_kernels(w, B, L, device)
        for kn, kv in kern.items():
            if kn.endswith("@FFN1") and probe_us is not None and not args.compact:
                kv["sec_back_to_back"], kv["sec"] = kv["sec"], probe_us * 1e-6
        # dominant hand-written kernel by time per step
        name, k = max(kern.items(), key=lambda kv: kv[1]["sec"] * kv[1]["launches_per_step"])
        if name.startswith("ltrx_gemm"):
            alg = k["flops"] / k["sec"] / 1e12            # algorithmic: the 2*M*N*K flop of the fp32 GEMM it replaces
            traffic, traffic_detail = (None, "skipped (--no-side-pass)")
            if not args.no_side_pass and world == 1 and name.endswith("@FFN1"):
                traffic, traffic_detail = measure_hbm_traffic(B * L, w["d_ff"], w["fc_sizes"][-1])
            roof = dict(kernel=name, bound="mfma", achieved=round(alg, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(alg / PEAK_BF16_MFMA_TFLOPS, 4), traffic=traffic,
                        traffic_detail=traffic_detail,
                        algorithmic_bytes_per_launch=4.0 * (B * L * w["fc_sizes"][-1] + w["d_ff"] * w["fc_sizes"][-1] + B * L * w["d_ff"]),
                        traffic_source="live: measure_hbm_traffic() (rocprofv3 --pmc child passes); committed copy of the same passes: profiles/r03_pmc_gemm256.md",
                        avg_launch_us=round(k["sec"] * 1e6, 1),
                        timing="HIP events around the %d FFN-1 launches of 5 eager training steps after the timed region" % (5 * w["N"]),
                        back_to_back_launch_us=round(k.get("sec_back_to_back", k["sec"]) * 1e6, 1),
                        algorithmic_flops_per_launch=k["flops"], arithmetic="bf16 MFMA, fp32 accumulate, 3 products per fp32 product (split-bf16)",
                        executed_mfma_tflops=round(3 * alg, 1), executed_frac=round(3 * alg / PEAK_BF16_MFMA_TFLOPS, 4),
                        vs_exact_fp32_mfma_peak=round(alg / PEAK_FP32_MFMA_TFLOPS, 3))
        elif "flops" in k:
            ach = k["flops"] / k["sec"] / 1e12
            roof = dict(kernel=name, bound="mfma", achieved=round(ach, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), traffic=None,
                        avg_launch_us=round(k["sec"] * 1e6, 1), algorithmic_flops_per_launch=k["flops"])
        else:
            by = k.get("bytes", 12.0 * B * L)
            ach = by / k["sec"] / 1e9
            roof = dict(kernel=name, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBPS, unit="GB/s",
                        frac=round(ach / PEAK_HBM_GBPS, 4), traffic=None, avg_launch_us=round(k["sec"] * 1e6, 1),
                        algorithmic_bytes_per_launch=by)
        if not w["N"]:
            # FCModel-only workloads (BASELINE configs[1]): 53 kFLOP per item against 552 B -- the HBM roofline binds.  Algorithmic bytes
            # per step = SURVEY 8(d)'s per-item figure (features once, label, score) x items per step; the step is the two launches of
            # ltrx_fc_listnet_step (slate-resident forward / loss / backward kernel + partial reduce with Adam), timed with HIP events
            # on the launch stream over steps that follow the timed region
            by = float(4 * w["n_features"] + 8) * B * L
            fcstep = bool(getattr(trainer, "fcstep", False))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(20):
                one_step(args.warmup + args.steps + i)
            e1.record()
            torch.cuda.synchronize()
            step_us = e0.elapsed_time(e1) / 20 * 1e3
            traffic, traffic_detail = (None, "skipped")
            if fcstep and not args.no_side_pass and world == 1:
                traffic, traffic_detail = measure_hbm_traffic(0, 0, 0, script="fc_one.py", kernels=("ltrx_fc_listnet_kernel", "ltrx_fc_reduce_kernel"),
                                                              env_extra=dict(FB=str(B), FL=str(L), FF=str(w["n_features"]), FH=str(w["fc_sizes"][0])))
            ach = by / (step_us * 1e-6) / 1e9
            roof = dict(kernel=("ltrx_fc_listnet_kernel + ltrx_fc_reduce_kernel (the whole step: two launches; per-kernel split in "
                                "profiles/r04_bench_fc_listnet_kernel_stats.md)" if fcstep else
                                "whole step (FC GEMMs + score head + ListNet + Adam: ~15 launches)"),
                        bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBPS, unit="GB/s", frac=round(ach / PEAK_HBM_GBPS, 4),
                        traffic=traffic, traffic_detail=traffic_detail, algorithmic_bytes_per_launch=by, avg_launch_us=round(step_us, 1),
                        timing="HIP events around 20 training steps after the timed region (both launches of a step)",
                        host_bound_note="wall-clock per step in the timed region: %.1f us" % (dt / args.steps * 1e6))
        wg_roof = None
        if wgrad_us:
            d_, dff_ = w["fc_sizes"][-1], w["d_ff"]
            flw = 2.0 * B * L * (3 * d_ * d_ + d_ * d_ + 2 * d_ * dff_)          # 2 M (N K) summed over the four projections of a layer
            tfw = flw / (wgrad_us * 1e-6) / 1e12
            wg_roof = dict(kernel="ltrx_gemm_tn256_kernel (ltrx_gemm_tn_group: the four weight gradients of an encoder layer in one launch)",
                           bound="mfma", achieved=round(tfw, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(tfw / PEAK_BF16_MFMA_TFLOPS, 4),
                           executed_mfma_tflops=round(3 * tfw, 1), executed_frac=round(3 * tfw / PEAK_BF16_MFMA_TFLOPS, 4),
                           avg_launch_us=round(wgrad_us, 1), launches_per_step=w["N"], algorithmic_flops_per_launch=flw,
                           algorithmic_bytes_per_launch=4.0 * B * L * (8 * d_ + 2 * dff_),     # dY and X of the four problems, read once
                           timing="HIP events around the grouped launch inside 5 eager training steps after the timed region")
        att_roof = None
        if w["N"] and "ltrx_mha_fwd (res split-bf16)" in kern:
            # the attention kernels against the matrix-core roof: algorithmic flops 4 L^2 d_k per (slate, head)
            # forward, 10 L^2 d_k backward (five tile products: S, dP, dV, dK in the dK/dV kernel, dQ in the second kernel), each
            # contraction executed as 3 bf16 MFMA products; HBM side: the compulsory tensors and the dS hand-over between the two
            # backward kernels (fp32 [B, h, LK, LK], written once and read once).  Timed back to back after the timed region.
            d_, h_ = w["fc_sizes"][-1], w["h"]
            LK = (L + 63) // 64 * 64
            kf, kb = kern["ltrx_mha_fwd (res split-bf16)"], kern["ltrx_mha_bwd (dq+dkdv, res split-bf16)"]
            def _ar(k_, io_bytes, extra=None):
                tf = k_["flops"] / k_["sec"] / 1e12
                r_ = dict(avg_launch_us=round(k_["sec"] * 1e6, 1), algorithmic_flops_per_launch=k_["flops"], achieved=round(tf, 1), unit="TFLOP/s",
                          peak=PEAK_BF16_MFMA_TFLOPS, frac=round(tf / PEAK_BF16_MFMA_TFLOPS, 4), executed_mfma_tflops=round(3 * tf, 1),
                          executed_frac=round(3 * tf / PEAK_BF16_MFMA_TFLOPS, 4), algorithmic_bytes_per_launch=io_bytes,
                          hbm_gbps_if_compulsory_only=round(io_bytes / k_["sec"] / 1e9, 1))
                r_.update(extra or {})
                return r_
            ds_b = 2.0 * 4.0 * B * h_ * LK * LK
            att_roof = dict(bound="mfma (issue-side: the softmax / split arithmetic and the LDS fragment reads ADD to the matrix time, they do not hide behind it)", launches_per_step=w["N"],
                            forward=_ar(kf, 4.0 * B * L * 4 * d_ + 4.0 * B * h_ * L),
                            backward=_ar(kb, 4.0 * B * L * 8 * d_ + 4.0 * B * h_ * L,
                                         dict(ds_handover_bytes=ds_b, hbm_gbps_with_ds_handover=round((4.0 * B * L * 8 * d_ + ds_b) / kb["sec"] / 1e9, 1),
                                              kernels="ltrx_mha_bwd_dkdv_res_kernel + ltrx_mha_bwd_dq_res_kernel")),
                            notes="profiles/r06_attention_forward_experiments.md (round 6: the tile period is the SUM of its MFMA, vector, LDS and staging parts; 64-query forward and phase shift slower; instruction diet: forward 182 -> 162 us, dK/dV 343 -> 325 us), profiles/r04_pmc_step_bytes_256.md (PMC traffic of both backward kernels), profiles/NOTES.md (chunked / 24-bit dS hand-over: measured slower)")
        loss_roof = None
        if w["loss"].startswith("neuralNDCG"):
            # the Sinkhorn kernels are VALU bound (no contraction): algorithmic flops = n^2 x (4 per forward step + 6 per backward step)
            # x max_iter + ~10 n^2 for NeuralSort / softmax and its backward, against the fp32 vector peak
            it_ = int(w.get("loss_args", {}).get("max_iter", 50))
            fl = float(B) * L * L * (10.0 * it_ + 10.0)
            sec = kern["loss_fwd_bwd"]["sec"]
            loss_roof = dict(kernel="ltrx_neural_forward_blk_kernel + ltrx_neural_backward_blk_kernel (whole plugin call)", bound="valu",
                             achieved=round(fl / sec / 1e12, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                             frac=round(fl / sec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), algorithmic_flops_per_launch=fl,
                             avg_launch_us=round(sec * 1e6, 1))
        out = {
            "metric": "slate-items/sec training (WEB30K synth, slate 240)", "value": round(value, 1),
            "unit": "slate-items/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (split-bf16x3 contractions: GEMMs and attention)" if (args.engine == "fused" and args.gemm == "split_bf16") else
                      "f32 (split-bf16x6 GEMMs, split-bf16x3 attention)" if (args.engine == "fused" and args.gemm == "split_bf16_strict") else
                      "bf16 products, f32 storage/accumulate (throughput mode, NOT parity-grade)" if (args.engine == "fused" and args.gemm == "bf16") else "f32"),
            "data": "synthetic",
            "config": {"workload": w["desc"], "slates_per_gpu": B, "slate_len": L, "global_batch": B * world,
                       "optimizer": "Adam lr=1e-3", "dropout": args.dropout, "slates": ("ragged (lognormal lengths)" + (", compact execution" if args.compact else "") if args.ragged else "dense"),
                       "arithmetic": ("fp32 storage and accumulation; ONE bf16 MFMA product per contraction in the dense projections and in attention (throughput mode)" if (args.engine == "fused" and args.gemm == "bf16") else "fp32 storage and accumulation; dense projections and attention contractions as fp32-accurate split-bf16 (3 bf16 MFMA products per fp32 product)" if (args.engine == "fused" and args.gemm != "hipblaslt") else "fp32 (hipBLASLt GEMMs, fp32 MFMA attention)"), "engine": args.engine, "gemm": args.gemm if args.engine == "fused" else "hipblaslt", "parallelism": "slate-sharded dp%d" % world,
                       "train_flops_per_item": fl_item},
            "model_tflops": round(value * fl_item / 1e12, 2),
            "model_mfma_frac_fp32": round(value * fl_item / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
            "algorithmic_hbm_frac": round(value * (4 * w["n_features"] + 8) / 1e9 / (PEAK_HBM_GBPS * world), 6),
            "last_loss": last_loss,
            "valid_items_per_s": (round(value * float((y != -1).float().mean().item()), 1) if args.ragged else None),
            "roofline": roof,
            "roofline_weight_gradient": wg_roof,
            "roofline_attention": att_roof,
            "roofline_loss_kernels": loss_roof,
            "kernel_times_us": {n: round(v["sec"] * 1e6, 1) for n, v in kern.items()},
        }
        if world == 1 and B != 64 and args.engine == "fused" and not args.no_side_pass:
            try:
                m64 = build_model(w, device, args.dropout)
                t64 = FusedTrainer(m64, w["loss"], w.get("loss_args", {}), 64, L, lr=1e-3, world_size=1, use_graph=True, gemm=args.gemm)
                for i in range(6):
                    t64.step(x[:64], y[:64], idx[:64])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(20):
                    t64.step(x[:64], y[:64], idx[:64])
                torch.cuda.synchronize()
                out["value_at_64_slates_per_gpu"] = round(20 * 64 * L / (time.perf_counter() - t0), 1)
            except Exception as e:      # never let the side measurement break the contract line
                out["value_at_64_slates_per_gpu"] = "failed: %r" % (e,)
        if world == 1 and w["N"] and args.engine == "fused" and not args.no_side_pass and not args.ragged and args.dropout == 0.0:
            # "a second number uses WEB30K-like lengths counting VALID items only" and the shipped configs train with
            # dropout 0.1 (reproducibility/configs/neuralndcg_web30k/approxndcg.json) -- both measured inside the default run, so the
            # driver's record carries them: (i) ragged slates (lognormal lengths, ~47 % of the slots valid) through the variable-length
            # step (FusedTrainer(compact=True): eager launches over the packed rows), valid items per second; (ii) the same dense
            # workload with every transformer dropout at 0.1 (masks generated in the kernels).
            def _side(trainer_, xs, ys, ids, lens=None, n=20):
                nb = xs.shape[0] // B
                def st(i):
                    j = (i % nb) * B
                    kw = dict(lengths=lens[j:j + B]) if lens is not None else {}
                    return trainer_.step(xs[j:j + B], ys[j:j + B], ids[j:j + B], **kw)
                for i in range(6):
                    st(i)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(n):
                    st(6 + i)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0_) / n
            try:
                xr, yr, ir = synth_batch(4 * B, L, w["n_features"], 4711, device, ragged=True)
                lens_r = (yr != -1).sum(1).cpu()
                frac = float((yr != -1).float().mean().item())
                mr = build_model(w, device, 0.0)
                tr_ = FusedTrainer(mr, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm=args.gemm, compact=True)
                sec = _side(tr_, xr, yr, ir, lens_r)
                out["valid_items_per_s"] = round(B * L * frac / sec, 1)
                out["ragged"] = {"valid_fraction": round(frac, 4), "ms_per_step": round(sec * 1e3, 4), "slots_per_s": round(B * L / sec, 1),
                                 "execution": "variable-length (compact=True): packed valid rows, per-slate extents in attention; eager launches"}
                del tr_, mr, xr, yr, ir
            except Exception as e:
                out["valid_items_per_s"] = "failed: %r" % (e,)
            try:
                md = build_model(w, device, 0.1)
                td = FusedTrainer(md, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm=args.gemm)
                out["value_dropout_0.1"] = round(B * L / _side(td, x, y, idx), 1)
                del td, md
            except Exception as e:
                out["value_dropout_0.1"] = "failed: %r" % (e,)
        if world == 1 and not w["N"] and args.engine == "fused" and not args.no_side_pass:
            try:           # the large-batch point ("and a large-batch point, e.g. 2048/GPU"): 8 slates per workgroup
                Bl = 2048
                xl, yl, il = synth_batch(2 * Bl, L, w["n_features"], 4242, device)
                ml = build_model(w, device, args.dropout)
                tl = FusedTrainer(ml, w["loss"], w.get("loss_args", {}), Bl, L, lr=1e-3, world_size=1, use_graph=True, gemm=args.gemm)
                for i in range(6):
                    tl.step(xl[(i % 2) * Bl:(i % 2 + 1) * Bl], yl[(i % 2) * Bl:(i % 2 + 1) * Bl], il[(i % 2) * Bl:(i % 2 + 1) * Bl])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(40):
                    tl.step(xl[(i % 2) * Bl:(i % 2 + 1) * Bl], yl[(i % 2) * Bl:(i % 2 + 1) * Bl], il[(i % 2) * Bl:(i % 2 + 1) * Bl])
                torch.cuda.synchronize()
                tsec = (time.perf_counter() - t0) / 40
                byl = float(4 * w["n_features"] + 8) * Bl * L
                trl, trd = (measure_hbm_traffic(0, 0, 0, script="fc_one.py", kernels=("ltrx_fc_listnet_kernel", "ltrx_fc_reduce_kernel"),
                                                env_extra=dict(FB=str(Bl), FL=str(L), FF=str(w["n_features"]), FH=str(w["fc_sizes"][0])))
                            if getattr(tl, "fcstep", False) else (None, "not the slate-resident step"))
                out["large_batch_2048_slates"] = {"value": round(Bl * L / tsec, 1), "unit": "slate-items/s", "us_per_step": round(tsec * 1e6, 1),
                                                  "algorithmic_bytes_per_step": byl, "achieved_gbps": round(byl / tsec / 1e9, 1),
                                                  "hbm_roofline_frac": round(byl / tsec / 1e9 / PEAK_HBM_GBPS, 4), "traffic": trl,
                                                  "traffic_over_algorithmic": (round(trl / byl, 3) if trl else None), "traffic_detail": trd}
                del tl, ml, xl, yl, il
            except Exception as e:
                out["large_batch_2048_slates"] = "failed: %r" % (e,)
            if w.get("fc_activation") is None:
                # opt-in specialisation for a LINEAR scorer (FC activation None, as this workload is specified): the two linear layers
                # evaluated as one matrix-vector product per slate with the exact rank-1 gradients (ltrx_fc_linear_listnet_step) --
                # fp32 FMAs, slate in registers, HBM-bound.  Reported beside `value`, never as `value`.
                try:
                    rec = {}
                    for Bc in (B, 2048):
                        xc, yc, ic = synth_batch(2 * Bc, L, w["n_features"], 777, device)
                        mc = build_model(w, device, args.dropout)
                        tc = FusedTrainer(mc, w["loss"], w.get("loss_args", {}), Bc, L, lr=1e-3, world_size=1, use_graph=False, gemm=args.gemm,
                                          fc_step="collapse")
                        assert tc.fcstep == "collapse"
                        for i in range(6):
                            tc.step(xc[(i % 2) * Bc:(i % 2 + 1) * Bc], yc[(i % 2) * Bc:(i % 2 + 1) * Bc], ic[(i % 2) * Bc:(i % 2 + 1) * Bc])
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for i in range(40):
                            tc.step(xc[(i % 2) * Bc:(i % 2 + 1) * Bc], yc[(i % 2) * Bc:(i % 2 + 1) * Bc], ic[(i % 2) * Bc:(i % 2 + 1) * Bc])
                        torch.cuda.synchronize()
                        tsec = (time.perf_counter() - t0) / 40
                        byc = float(4 * w["n_features"] + 8) * Bc * L
                        rec["slates_%d" % Bc] = {"value": round(Bc * L / tsec, 1), "us_per_step": round(tsec * 1e6, 1),
                                                 "achieved_gbps": round(byc / tsec / 1e9, 1), "hbm_roofline_frac": round(byc / tsec / 1e9 / PEAK_HBM_GBPS, 4)}
                        del tc, mc, xc, yc, ic
                    rec["what"] = ("FusedTrainer(fc_step='collapse'): score = x . (W1^T w_out) + c, dW1 = w_out (x) u -- exact algebra of a linear "
                                   "scorer, fp32 FMAs, no matrix cores; parity: tests/test_gpu_fcstep.py::test_fc_linear_listnet_step_*")
                    out["linear_scorer_collapse"] = rec
                except Exception as e:
                    out["linear_scorer_collapse"] = "failed: %r" % (e,)
            try:           # A/B: the same workload through the GEMM launch sequence (FusedTrainer(fc_step=False), hipGraph)
                mg = build_model(w, device, args.dropout)
                tg = FusedTrainer(mg, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm=args.gemm, fc_step=False)
                for i in range(6):
                    tg.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(20):
                    tg.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                out["value_gemm_launch_sequence"] = round(20 * B * L / (time.perf_counter() - t0), 1)
                del tg, mg
            except Exception as e:
                out["value_gemm_launch_sequence"] = "failed: %r" % (e,)
        out["comm"] = comm
        if world == 1 and not args.no_side_pass:
            try:          # what a caller that hands over HOST batches (the reference's DataLoader, train_utils.py:95) would add per step
                hx, hy, hi = (t[:B].cpu().pin_memory() for t in (x, y, idx))
                dx, dy, di = torch.empty_like(x[:B]), torch.empty_like(y[:B]), torch.empty_like(idx[:B])
                for _ in range(2):
                    dx.copy_(hx, non_blocking=True); dy.copy_(hy, non_blocking=True); di.copy_(hi, non_blocking=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    dx.copy_(hx, non_blocking=True); dy.copy_(hy, non_blocking=True); di.copy_(hi, non_blocking=True)
                torch.cuda.synchronize()
                h2d_ms = (time.perf_counter() - t0) / 10 * 1e3
                step_ms = dt / args.steps * 1e3
                out["host_batches"] = {"h2d_ms_per_batch": round(h2d_ms, 4), "bytes_per_batch": int(hx.numel() * 4 + hy.numel() * 4 + hi.numel() * 8),
                                       "value_if_copy_not_overlapped": round(B * L / ((step_ms + h2d_ms) * 1e-3), 1),
                                       "value_if_copy_overlapped": round(B * L / (max(step_ms, h2d_ms) * 1e-3), 1),
                                       "note": "pinned host memory -> HBM of one batch (features, labels, indices); `value` itself is measured with the batch resident in HBM"}
            except Exception as e:
                out["host_batches"] = "failed: %r" % (e,)
        if world == 1 and args.engine == "fused" and not args.no_side_pass and not args.ragged and L == 240 and w["n_features"] <= 136:
            # VERDICT r5 item 1 / SURVEY 8(d) "time the reference's own loop once": the drop-in path END TO END -- what an unmodified
            # main.py runs after install(fit=True): libsvm files -> device parse -> HBM-resident slates -> DeviceLoader -> fit().
            # (i) a WEB30K-shaped ragged job (2048 queries, lognormal lengths): valid items/s over the training pass, beside
            # `valid_items_per_s` (the same variable-length step fed from resident tensors); (ii) a dense job (every query L items):
            # slots/s beside `value`.  The same fit() fed by the reference's host loader is measured in the cpu_baseline leg.
            import shutil
            import tempfile
            e2e_dir = tempfile.mkdtemp(prefix="ltrx_e2e_")
            try:
                rag = end_to_end_main(w, B, L, device, os.path.join(e2e_dir, "ragged"), dense=False, n_queries=max(2048, 8 * B), epochs=3,
                                      gemm=args.gemm)
                den = end_to_end_main(w, B, L, device, os.path.join(e2e_dir, "dense"), dense=True, n_queries=2 * B, epochs=8, gemm=args.gemm)
                out["end_to_end_main_items_per_s"] = den["slots_per_s"]
                out["end_to_end_main"] = {
                    "dense": den, "ragged": rag, "fraction_of_value": round(den["slots_per_s"] / value, 4),
                    "ragged_fraction_of_valid_items_per_s": (round(rag["valid_items_per_s"] / out["valid_items_per_s"], 4)
                                                             if isinstance(out.get("valid_items_per_s"), float) else None),
                    "what": "main.py's sequence after allrank_amd.install(fit=True): seeds, load_libsvm_dataset (parsed on the GPU, resident in "
                            "HBM), create_data_loaders (DeviceLoader, the reference loader's batch order), make_model, Adam, fit(): rate over the "
                            "training pass of the steady epochs (loader + step + train metrics, wall clock inside fit)"}
            except Exception as e:
                out["end_to_end_main_items_per_s"] = "failed: %r" % (e,)
            finally:
                shutil.rmtree(e2e_dir, ignore_errors=True)
        if w["N"] and args.engine == "fused":
            try:                               # measured arithmetic error of the benchmarked GEMM (and of the alternatives)
                out["gemm_max_rel_err_vs_fp64"] = {g_: gemm_error_vs_fp64(w, B, L, device, g_) for g_ in
                                                   sorted({args.gemm, "split_bf16_strict", "hipblaslt", "bf16"})}
            except Exception as e:
                out["gemm_max_rel_err_vs_fp64"] = "failed: %r" % (e,)
        if world == 1 and args.engine == "fused" and w["N"] and not args.no_side_pass and not args.compact:
            for key, gm in (("value_strict_fp32", "split_bf16_strict"), ("value_hipblaslt_fp32", "hipblaslt")):
                if gm == args.gemm:
                    continue
                try:
                    m2 = build_model(w, device, args.dropout)
                    t2 = FusedTrainer(m2, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm=gm)
                    for i in range(4):
                        t2.step(x[:B], y[:B], idx[:B])
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for i in range(10):
                        t2.step(x[:B], y[:B], idx[:B])
                    torch.cuda.synchronize()
                    out[key] = round(10 * B * L / (time.perf_counter() - t0), 1)
                    del t2, m2
                except Exception as e:
                    out[key] = "failed: %r" % (e,)
        if world == 1 and args.engine == "fused" and w["N"] and not args.no_side_pass and not args.compact and args.gemm == "split_bf16":
            # SURVEY.md §8(d) "strict fp32 mode for parity, bf16 mode for throughput ... report both": the same step with ONE bf16
            # product per contraction (GEMMs + attention).  Its loss error is measured here against the parity arithmetic at
            # identical weights (same seed -> same initial model, first step of the same batch); it is NOT `value`.
            try:
                ma, mb = build_model(w, device, args.dropout), build_model(w, device, args.dropout)
                ta = FusedTrainer(ma, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm="split_bf16")
                tb = FusedTrainer(mb, w["loss"], w.get("loss_args", {}), B, L, lr=1e-3, world_size=1, use_graph=True, gemm="bf16")
                la, lb = float(ta.step(x[:B], y[:B], idx[:B])), float(tb.step(x[:B], y[:B], idx[:B]))
                sa, sb = ta.scores.clone(), tb.scores.clone()
                for i in range(4):
                    tb.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(10):
                    tb.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                out["throughput_mode_bf16"] = {
                    "value": round(10 * B * L / (time.perf_counter() - t0), 1), "unit": "slate-items/s",
                    "arithmetic": "one bf16 MFMA product per contraction (dense projections + attention), fp32 storage / accumulation / LayerNorm / softmax / loss / Adam",
                    "loss_abs_err_vs_parity_mode_step0": abs(la - lb), "loss_parity_mode_step0": la,
                    "score_max_abs_err_vs_parity_mode_step0": float((sa - sb).abs().max().item()),
                    "score_max_abs": float(sa.abs().max().item()),
                    "within_1e-5_parity": bool(abs(la - lb) <= 1e-5)}
                del ta, tb, ma, mb
            except Exception as e:
                out["throughput_mode_bf16"] = "failed: %r" % (e,)
        if world == 1 and args.engine == "fused" and w["N"] and not args.no_side_pass and not args.compact:
            # the PLAIN drop-in: allrank_amd.install() without fit=True -- the reference's own loss_batch (train_utils.py:18-29):
            # nn.Module forward, torch autograd, torch.optim.Adam -- around the same HIP kernels (ops.linear / feed_forward /
            # attention_packed / layer_norm / the fused loss)
            try:
                m3 = build_model(w, device, args.dropout)
                o3 = torch.optim.Adam(m3.parameters(), lr=1e-3)
                _lf3, _la3 = getattr(E, w["loss"]), w.get("loss_args", {})
                t3 = Trainer(m3, (lambda sc, yt: _lf3(sc, yt, **_la3)), o3, None, 1, None)
                for i in range(3):
                    t3.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(10):
                    t3.step(x[:B], y[:B], idx[:B])
                torch.cuda.synchronize()
                out["value_plain_dropin_autograd"] = round(10 * B * L / (time.perf_counter() - t0), 1)
                del t3, o3, m3
            except Exception as e:
                out["value_plain_dropin_autograd"] = "failed: %r" % (e,)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w, L)
            if args.engine == "fused" and not args.no_side_pass and not args.ragged and L == 240 and w["n_features"] <= 136:
                out["cpu_baseline"]["reference_loader"] = reference_loader_leg(w, B, L, device, args.gemm, out.get("end_to_end_main"))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
