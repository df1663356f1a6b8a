"""CPU-side checks of the drop-in boundary: libltrx.so builds for gfx950, loads, and exports every symbol that
include/ltrx.h declares; the Python surface mirrors the reference plugin signatures.  No kernel is launched."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    from allrank_amd import build
    return build.build(verbose=False)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ltrx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ltrx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(libpath):
    names = _declared_symbols()
    assert len(names) >= 20
    h = ctypes.CDLL(libpath)
    for n in names:
        assert hasattr(h, n), "libltrx.so does not export %s declared in include/ltrx.h" % n
    h.ltrx_version.restype = ctypes.c_int
    assert h.ltrx_version() == 130


def test_binding_table_matches_header(libpath):
    from allrank_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()
    assert _lib.lib().ltrx_version() == 130


def test_slate_length_limits_are_stated_once_and_reported(libpath):
    """the limits of include/ltrx.h, the constants the Python error message quotes, and the status code of an over-long slate
    (returned before any HIP call, so this runs without a GPU)"""
    from allrank_amd import _lib
    src = open(os.path.join(ROOT, "include", "ltrx.h")).read()
    assert int(re.search(r"#define LTRX_MAX_SLATE_LEN (\d+)", src).group(1)) == _lib.MAX_SLATE_LEN
    assert int(re.search(r"#define LTRX_MAX_METRIC_SLATE_LEN (\d+)", src).group(1)) == _lib.MAX_METRIC_SLATE_LEN
    lib = _lib.lib()
    fake = ctypes.c_void_p(4096)                       # never dereferenced: the shape check comes first
    ats = (ctypes.c_int * 1)(5)
    assert lib.ltrx_ndcg_at(fake, fake, 1, _lib.MAX_METRIC_SLATE_LEN + 1, ats, 1, -1.0, 1.0, fake, None, None, None, None) == -2
    assert int(re.search(r"#define LTRX_MAX_LONG_SLATE_LEN (\d+)", src).group(1)) == _lib.MAX_LONG_SLATE_LEN
    assert lib.ltrx_listnet_fwd_bwd(fake, fake, 1, _lib.MAX_LONG_SLATE_LEN + 1, 1e-10, -1.0, 1.0, fake, None, None, fake, None) == -2
    with pytest.raises(RuntimeError, match="LTRX_MAX_METRIC_SLATE_LEN = %d" % _lib.MAX_METRIC_SLATE_LEN):
        _lib.check(-2, "ndcg_at")


def test_workspace_queries_and_argument_validation_run_without_a_gpu(libpath):
    from allrank_amd import _lib
    lib = _lib.lib()
    assert lib.ltrx_listnet_workspace_bytes(64, 240) >= 64 * 4
    assert lib.ltrx_neuralndcg_workspace_bytes(64, 240, 50) >= 2 * 64 * 240 * 240 * 4
    assert lib.ltrx_mha_bwd_workspace_bytes(64, 240, 8, 64, 0) == 64 * 240 * 8 * 4
    # modes 1 / 2, LDS-resident backward: the dS exchange, B*h*256*256 floats
    assert lib.ltrx_mha_bwd_workspace_bytes(64, 240, 8, 64, 1) == 64 * 8 * 256 * 256 * 4
    assert lib.ltrx_mha_bwd_workspace_bytes(64, 240, 8, 128, 1) == 64 * 240 * 8 * 4
    # NULL pointers / bad shapes are rejected before any HIP call
    assert lib.ltrx_listnet_fwd_bwd(None, None, 1, 1, 1e-10, -1.0, 1.0, None, None, None, None, None) == -1
    assert lib.ltrx_mha_fwd(None, None, None, None, 1, 1, 1, 64, 64, None, 64, None, 0.0, 0, None, None, None, 1, None) == -1


def test_wgrad_workspace_covers_every_smaller_row_count(libpath):
    """a variable-length batch calls ltrx_gemm_tn with a different row count every step against a
    workspace sized once for B*L rows; the split plan is not monotone in the row count, so the size query must be an
    upper bound over all row counts <= M."""
    from allrank_amd import _lib
    lib = _lib.lib()
    for (M, NP, KP) in [(16 * 240, 2048, 512), (16 * 240, 512, 2048), (64 * 240, 1024, 256), (64 * 240, 256, 1024),
                        (100 * 50, 512, 512), (256 * 240, 1536, 512), (5000, 96, 136), (4096, 768, 256)]:
        have = lib.ltrx_gemm_tn_workspace_bytes(M, NP, KP)
        step = 1 if M <= 6000 else 7
        for m in list(range(1, M + 1, step)) + [M]:
            sp = lib.ltrx_gemm_tn_splits(m, NP, KP)
            need = (sp * NP * KP + 2 * sp * NP) * 4
            assert need <= have, (M, NP, KP, m, sp, need, have)


def test_grouped_wgrad_workspace_and_relu_bits_queries(libpath):
    """round 4 size queries, no GPU: the grouped weight-gradient workspace bounds the grouped plan of every row count <= M (at most
    256 / total_tiles splits, never more than m / 128) and every single-problem call it may fall back to; the padded-B form of
    ltrx_gemm_tn (KP not a multiple of 256, row stride covering the tile) is inside the single-problem bound; the one-bit ReLU mask
    exists exactly where the large-tile kernel is unconditional."""
    import ctypes
    from allrank_amd import _lib
    lib = _lib.lib()
    for (M, probs) in [(64 * 240, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]), (256 * 240, [(1536, 512), (512, 512), (2048, 512), (512, 2048)]),
                       (6176, [(768, 256), (256, 256), (512, 256), (256, 512)]), (4096, [(256, 256)])]:
        n = len(probs)
        NP = (ctypes.c_int * n)(*[a for a, _ in probs])
        KP = (ctypes.c_int * n)(*[b for _, b in probs])
        have = lib.ltrx_gemm_tn_group_workspace_bytes(n, M, NP, KP)
        tiles = sum((a // 256) * (b // 256) for a, b in probs)
        for m in range(2048, M + 1, 32):
            sp = max(1, min(256 // tiles, m // 128))
            mps = ((m + sp - 1) // sp + 31) // 32 * 32
            splits = (m + mps - 1) // mps
            need = sum((splits * a * b + splits * a + 4) * 4 for a, b in probs)
            assert need <= have, (M, m, splits, need, have)
        assert have >= max(lib.ltrx_gemm_tn_workspace_bytes(M, a, b) for a, b in probs)
    assert lib.ltrx_gemm_tn_group_workspace_bytes(0, 1024, None, None) == 0
    # padded-B weight gradient (F = 136 in rows of 256 floats): splits <= min(256 / tiles256, m / 128) slabs of NP x KP
    for (M, NP, KP) in [(64 * 240, 512, 136), (256 * 240, 512, 136), (4096, 256, 300)]:
        have = lib.ltrx_gemm_tn_workspace_bytes(M, NP, KP)
        tiles = (NP // 256) * ((KP + 255) // 256)
        for m in range(2048, M + 1, 32):
            sp = max(1, min(256 // tiles, m // 128))
            assert (sp * NP * KP + sp * NP + 4) * 4 <= have, (M, NP, KP, m)
    for (M, N, K, want) in [(61440, 2048, 512, True), (15360, 2048, 512, True), (16384, 1024, 256, True), (12000, 2048, 512, False),
                            (15360, 512, 512, False), (15360, 2000, 512, False), (1000, 2048, 512, False),
                            (15360, 1024, 144, False)]:      # d_model 144: K % 32 != 0 -> acts 1 / 2
        b = lib.ltrx_gemm_nt_relu_bits_bytes(M, N, K)
        assert (b > 0) == want and (not want or b == ((M + 255) // 256) * (N // 256) * 8192), (M, N, K, b)
    # argument validation before any HIP call
    assert lib.ltrx_reduce_group(17, None, None, None, None, None, None) == -1
    assert lib.ltrx_ingest_batch(None, None, 0, 0, 0, 0, -1.0, None, None, None, None) == -1
    assert lib.ltrx_weight_images(None, 0, None, None, None, None, None, 0, 0, None, 0, 0, 0, None, None, None) == -1


def test_grouped_wgrad_workgroup_map_is_a_permutation_that_keeps_groups_on_one_xcd(libpath):
    """the host-built workgroup -> (tile, split) table of ltrx_gemm_tn_group (no GPU): every (tile, split) exactly once, at most 256
    workgroups, and the tiles of a (problem, split) group on ONE XCD (workgroup w runs on XCD w % 8) wherever bin packing allows --
    15 of the 20 groups of the bench layer, all of them for shapes that divide evenly."""
    import ctypes
    from allrank_amd import _lib
    lib = _lib.lib()
    cases = [(64 * 240, [(1536, 512), (512, 512), (2048, 512), (512, 2048)], 15), (256 * 240, [(1536, 512), (512, 512), (2048, 512), (512, 2048)], 15),
             (8192, [(768, 256), (256, 256), (512, 256), (256, 512)], None), (4096, [(256, 256)], None), (6176, [(512, 512), (512, 512)], None)]
    for (M, probs, want_whole) in cases:
        n = len(probs)
        NP = (ctypes.c_int * n)(*[a for a, _ in probs])
        KP = (ctypes.c_int * n)(*[b for _, b in probs])
        t_out, s_out = (ctypes.c_ubyte * 256)(), (ctypes.c_ubyte * 256)()
        nwg = lib.ltrx_debug_tn_group_map(n, M, NP, KP, t_out, s_out)
        tiles = [(a // 256) * (b // 256) for a, b in probs]
        total = sum(tiles)
        assert 0 < nwg <= 256 and nwg % total == 0, (M, probs, nwg)
        splits = nwg // total
        pairs = sorted((t_out[w], s_out[w]) for w in range(nwg))
        assert pairs == sorted((t, s) for t in range(total) for s in range(splits)), (M, probs)
        starts = [sum(tiles[:p]) for p in range(n + 1)]
        whole = 0
        for p in range(n):
            for s in range(splits):
                xcds = {w % 8 for w in range(nwg) if s_out[w] == s and starts[p] <= t_out[w] < starts[p + 1]}
                whole += len(xcds) == 1
                assert len(xcds) <= max(2, (tiles[p] + 31) // 32 + 1)
        if want_whole is not None:
            assert whole >= want_whole, (M, whole)
    NPb, KPb = (ctypes.c_int * 1)(96), (ctypes.c_int * 1)(136)
    assert lib.ltrx_debug_tn_group_map(1, 4096, NPb, KPb, t_out, s_out) == 0


def test_loss_signatures_mirror_reference():
    """same parameter names and defaults as allrank/models/losses/*.py"""
    from allrank_amd import losses, metrics
    exp = {
        "listNet": ["y_pred", "y_true", "eps", "padded_value_indicator"],
        "listMLE": ["y_pred", "y_true", "eps", "padded_value_indicator"],
        "approxNDCGLoss": ["y_pred", "y_true", "eps", "padded_value_indicator", "alpha"],
        "lambdaLoss": ["y_pred", "y_true", "eps", "padded_value_indicator", "weighing_scheme", "k", "sigma", "mu",
                       "reduction", "reduction_log"],
        "neuralNDCG": ["y_pred", "y_true", "padded_value_indicator", "temperature", "powered_relevancies", "k",
                       "stochastic", "n_samples", "beta", "log_scores"],
        "neuralNDCG_transposed": ["y_pred", "y_true", "padded_value_indicator", "temperature", "powered_relevancies", "k",
                                  "stochastic", "n_samples", "beta", "log_scores", "max_iter", "tol"],
    }
    for name, params in exp.items():
        got = list(inspect.signature(getattr(losses, name)).parameters)
        assert got[:len(params)] == params, (name, got)
    assert list(inspect.signature(metrics.ndcg).parameters)[:6] == ["y_pred", "y_true", "ats", "gain_function",
                                                                    "padding_indicator", "filler_value"]
    d = inspect.signature(losses.lambdaLoss).parameters
    assert d["sigma"].default == 1. and d["mu"].default == 10. and d["reduction"].default == "sum" \
        and d["reduction_log"].default == "binary" and d["eps"].default == 1e-10 and d["padded_value_indicator"].default == -1


def test_product_path_refuses_cpu_tensors(libpath):
    import torch
    from allrank_amd import losses
    with pytest.raises(RuntimeError):
        losses.listNet(torch.zeros(2, 3), torch.zeros(2, 3))


def test_reference_error_behaviour():
    import torch
    from allrank_amd import losses
    x = torch.zeros(2, 3)
    with pytest.raises(ValueError):
        losses.lambdaLoss(x, x, reduction="median")             # lambdaLoss.py:79
    with pytest.raises(ValueError):
        losses.lambdaLoss(x, x, reduction_log="decimal")        # lambdaLoss.py:72
    with pytest.raises(KeyError):
        losses.lambdaLoss(x, x, weighing_scheme="nope")         # globals()[...] at lambdaLoss.py:61
