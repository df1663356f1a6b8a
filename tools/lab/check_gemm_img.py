"""-m gpu: the pre-split ("HL16 image") GEMM path of allrank_amd/csrc/ltrx_gemm_img.hip against fp64.

Same bar as the on-the-fly split kernels (tests/test_gpu_parity.py::test_split_bf16_gemm_matches_fp64): the error relative
to max sum_k |a||b| must be fp32-class (<= 4e-6; measured ~1e-6), for every epilogue the training step uses (bias, ReLU,
ReLU-backward mask read from an image, image output, dropout), with row counts that are not multiples of the 256-row tile
(bounds-checked buffer accesses instead of tail code) and for the weight-gradient kernel with its bias column sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from allrank_amd import _lib as LB
    return LB, LB.lib()


def to_image(x):
    """fp32 [R, K] device tensor -> HL16 image (float32-typed byte container of the same shape)"""
    LB, lib = _lib()
    R, K = x.shape
    img = torch.empty_like(x)
    LB.check(lib.ltrx_to_image(LB.ptr(x), x.stride(0), R, K, LB.ptr(img), K, LB.stream_of(x)), "to_image")
    return img


def decode_image(img):
    """HL16 image -> (hi + lo) as float64 numpy [R, K]"""
    R, K = img.shape
    raw = img.cpu().numpy().view(np.uint16).reshape(R, K // 16, 2, 16)          # [row][block][plane][16]
    f = (raw.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    return (f[:, :, 0, :] + f[:, :, 1, :]).reshape(R, K)


def test_image_round_trip():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((77, 160)) * np.exp(rng.uniform(-8, 8, (77, 160)))).astype(np.float32)
    img = to_image(torch.tensor(x, device=DEV))
    back = decode_image(img)
    assert np.abs(back - x).max() <= 2.0 ** -16 * np.abs(x).max()
    assert (np.abs(back - x) <= 2.0 ** -15 * np.abs(x) + 1e-38).all()
    # the hi plane is exactly torch's round-to-nearest bf16
    hi = (img.cpu().numpy().view(np.uint16).reshape(77, 10, 2, 16)[:, :, 0, :].astype(np.uint32) << 16).view(np.float32).reshape(77, 160)
    assert np.array_equal(hi, torch.tensor(x).bfloat16().float().numpy())


@pytest.mark.parametrize("M,N,K", [(1000, 512, 512), (4096, 256, 2048), (23040, 1536, 512), (61440, 2048, 512), (300, 256, 16), (32, 256, 48)])
def test_gemm_nt_img_matches_fp64(M, N, K):
    LB, lib = _lib()
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    At, Wt, bt = torch.tensor(A, device=DEV), torch.tensor(W, device=DEV), torch.tensor(bias, device=DEV)
    Ai, Wi = to_image(At), to_image(Wt)
    rows = np.unique(np.concatenate([np.arange(min(M, 300)), np.linspace(0, M - 1, min(M, 2000)).astype(int), np.arange(max(0, M - 300), M)]))
    ref = A[rows].astype(np.float64) @ W.astype(np.float64).T
    scale = float((np.abs(A[rows]).astype(np.float64) @ np.abs(W).astype(np.float64).T).max())
    tol = 4e-6 if K >= 64 else 1.2e-5        # 3 * 2^-18 per product worst case; it averages down only over a long contraction

    def run(act, c_img, bias_t, aux=None, p=0.0, seed=0):
        C = torch.full((M, N), float("nan"), device=DEV)
        LB.check(lib.ltrx_gemm_nt_img(LB.ptr(Ai), K, LB.ptr(Wi), K, LB.ptr(C), N, 1 if c_img else 0, M, N, K, LB.ptr(bias_t), act,
                                      LB.ptr(aux), N if aux is not None else 0, p, seed, None, LB.stream_of(At)), "gemm_nt_img")
        return C
    # bias, no activation, fp32 out
    C = run(0, False, bt)
    err = np.abs(C[torch.tensor(rows, device=DEV)].cpu().numpy() - (ref + bias)).max() / scale
    assert err < tol, ("plain", err)
    assert torch.isfinite(C).all()
    # ReLU + image out
    Ci = run(1, True, bt)
    got = decode_image(Ci)[rows]
    err = np.abs(got - np.maximum(ref + bias, 0)).max() / scale
    assert err < 2 * tol, ("relu image", err)          # + the 2^-17 of storing the result as hi + lo
    # ReLU-backward mask read from an image (no bias), fp32 and image out
    auxv = rng.standard_normal((M, N)).astype(np.float32)
    auxi = to_image(torch.tensor(auxv, device=DEV))
    C2 = run(2, False, None, aux=auxi)
    err = np.abs(C2[torch.tensor(rows, device=DEV)].cpu().numpy() - ref * (auxv[rows] > 0)).max() / scale
    assert err < tol, ("mask", err)
    # dropout in the epilogue: kept entries are value / (1 - p), the keep rate is 1 - p, the mask is a function of (seed, index)
    if M >= 1000:
        p = 0.25
        Cd = run(0, False, bt, p=p, seed=123)
        keep = (Cd != 0)
        rate = float(keep.float().mean().item())
        assert abs(rate - (1 - p)) < 0.01, rate
        assert torch.allclose(Cd[keep], (C / (1 - p))[keep], rtol=1e-6, atol=1e-7)
        assert torch.equal(Cd, run(0, False, bt, p=p, seed=123)) and not torch.equal(Cd, run(0, False, bt, p=p, seed=124))


@pytest.mark.parametrize("M,NP,KP", [(4096, 512, 256), (23040, 2048, 512), (61440, 512, 512), (1040, 256, 256), (61440, 1536, 512), (2048, 512, 2048)])
def test_gemm_tn_img_matches_fp64(M, NP, KP):
    LB, lib = _lib()
    rng = np.random.default_rng(M + NP)
    dY = rng.standard_normal((M, NP)).astype(np.float32)
    X = rng.standard_normal((M, KP)).astype(np.float32)
    dYt, Xt = torch.tensor(dY, device=DEV), torch.tensor(X, device=DEV)
    dYi, Xi = to_image(dYt), to_image(Xt)
    C = torch.full((NP, KP), float("nan"), device=DEV)
    gb = torch.full((NP,), float("nan"), device=DEV)
    ws = torch.empty(max(lib.ltrx_gemm_tn_workspace_bytes(M, NP, KP), 64), dtype=torch.uint8, device=DEV)
    LB.check(lib.ltrx_gemm_tn_img(LB.ptr(dYi), NP, LB.ptr(Xi), KP, LB.ptr(C), LB.ptr(gb), M, NP, KP, LB.ptr(ws), LB.stream_of(dYt)), "gemm_tn_img")
    ref = (dYt.double().t() @ Xt.double()).cpu().numpy()
    scale = float((dYt.abs().double().t() @ Xt.abs().double()).max().item())
    err = np.abs(C.cpu().numpy() - ref).max() / scale
    assert err < 4e-6, err
    bref = dY.astype(np.float64).sum(0)
    berr = np.abs(gb.cpu().numpy() - bref).max() / np.abs(dY).sum(0).max()
    assert berr < 4e-6, berr
    # without the bias output
    C2 = torch.empty((NP, KP), device=DEV)
    LB.check(lib.ltrx_gemm_tn_img(LB.ptr(dYi), NP, LB.ptr(Xi), KP, LB.ptr(C2), None, M, NP, KP, LB.ptr(ws), LB.stream_of(dYt)), "gemm_tn_img")
    assert torch.equal(C, C2)
