"""GPU parity of the head GEMM kernels (csrc/gemm.hip): forward / input-gradient form with fused bias and GroupNorm
statistics, and the weight-gradient form (hardware transpose reads), vs fp32 torch on the same bf16-rounded operands
(products of bf16 values are exact in f32: the difference left is f32 summation order, 1e-4 relative to the scale)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K,bias,out_f32", [
    (1024, 512, 256, True, 0), (1024, 256, 512, True, 0), (1000, 256, 832, True, 0), (640, 64, 256, True, 1),
    (384, 10, 256, True, 1), (384, 22, 256, False, 1), (300, 3, 256, True, 1), (512, 128, 272, True, 0),
    (256, 1024, 256, True, 0), (512, 256, 16, False, 0), (512, 256, 32, True, 0), (129, 96, 80, True, 1),
    # M >= 32768 and N >= 256 also run with the opt-in 256 x 256 tile (128 x 128 per wave): ragged M, N not a multiple of
    # the tile, short K
    (32768, 512, 256, True, 0), (32768 + 100, 256, 1280, True, 0), (40000, 320, 48, True, 1), (32768, 1024, 272, False, 0)])
def test_gemm_forward(dev, M, N, K, bias, out_f32):
    _check_forward(dev, M, N, K, bias, out_f32)


@pytest.mark.parametrize("M,N,K,bias,out_f32", [(32768, 512, 256, True, 0), (32768 + 100, 256, 1280, True, 0),
                                                (40000, 320, 48, True, 1), (32768, 1024, 272, False, 0)])
def test_gemm_forward_tile256(dev, monkeypatch, M, N, K, bias, out_f32):
    """The opt-in 256 x 256 workgroup tile (GCANET_GEMM_TILE=256; serves M >= 32768, N >= 256)."""
    monkeypatch.setenv("GCANET_GEMM_TILE", "256")
    _check_forward(dev, M, N, K, bias, out_f32)


def _check_forward(dev, M, N, K, bias, out_f32):
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=g)).to(dev)
    Np = (N + 31) // 32 * 32
    W = torch.zeros(Np, K, dtype=torch.bfloat16)
    W[:N] = _bf(torch.randn(N, K, generator=g) / K ** 0.5)
    W = W.to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    out = torch.empty(M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), out_f32, M, N, Np, K, None, None, 0, 0,
              _lib.stream_of(A))
    ref = A.float() @ W[:N].float().t()
    if bias:
        ref = ref + b
    tol = 1e-4 if out_f32 else 1e-2           # bf16 output: one rounding of the result (2^-9 relative)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * float(ref.abs().max()))


@pytest.mark.parametrize("B,Npts,N,K,G", [(2, 256, 512, 256, 8), (3, 128, 256, 512, 4), (2, 384, 128, 272, 4), (1, 256, 1024, 256, 8),
                                          (8, 4096, 512, 256, 8), (5, 8192, 256, 64, 4)])
def test_gemm_fused_groupnorm_statistics(dev, B, Npts, N, K, G):
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(N + K + G)
    M = B * Npts
    A = _bf(torch.randn(M, K, generator=g)).to(dev)
    W = _bf(torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gsum = torch.empty(B, G, 2, dtype=torch.float64, device=dev)
    ws = torch.empty(_lib.lib().gcn_gemm_stats_ws_bytes(M, N), dtype=torch.uint8, device=dev)
    _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), 0, M, N, N, K, _lib.ptr(gsum), _lib.ptr(ws),
              Npts, G, _lib.stream_of(A))
    y = (A.float() @ W.float().t() + b).double().view(B, Npts, G, N // G)
    np.testing.assert_allclose(gsum[..., 0].cpu().numpy(), y.sum((1, 3)).cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(gsum[..., 1].cpu().numpy(), (y * y).sum((1, 3)).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(2048, 256, 256), (4096, 512, 256), (1000, 64, 256), (3000, 16, 256), (2048, 256, 272),
                                   (2048, 128, 832), (700, 32, 256), (5000, 256, 1024), (1024, 8, 64), (65536, 64, 256),
                                   (40000, 512, 1280), (33, 16, 16)])
def test_gemm_weight_gradient(dev, M, N, K):
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    dY = _bf(torch.randn(M, N, generator=g)).to(dev)
    X = _bf(torch.randn(M, K, generator=g)).to(dev)
    dW = torch.empty(N, K, device=dev)
    ws = torch.empty(_lib.lib().gcn_gemm_wgrad_ws_bytes(M, N, K), dtype=torch.uint8, device=dev)
    _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dY), _lib.ptr(X), M, N, K, _lib.ptr(dW), None, _lib.ptr(ws), _lib.stream_of(X))
    ref = dY.float().t() @ X.float()
    np.testing.assert_allclose(dW.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    # with the bias gradient (column sums of dY) from the same pass, adjacent and separate accumulators
    raw = torch.empty(N * K + N, device=dev)
    _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dY), _lib.ptr(X), M, N, K, _lib.ptr(raw), _lib.ptr(raw[N * K:]), _lib.ptr(ws), _lib.stream_of(X))
    db2 = torch.full((N,), 7.0, device=dev)
    _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dY), _lib.ptr(X), M, N, K, _lib.ptr(dW), _lib.ptr(db2), _lib.ptr(ws), _lib.stream_of(X))
    dbr = dY.float().sum(0)
    for got in (raw[N * K:], db2):
        np.testing.assert_allclose(got.cpu().numpy(), dbr.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(dbr.abs().max()) + 1e-5)
    np.testing.assert_allclose(raw[:N * K].view(N, K).cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    # slices are added in a fixed order: a second run gives the same bits
    again = torch.empty_like(raw)
    _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dY), _lib.ptr(X), M, N, K, _lib.ptr(again), _lib.ptr(again[N * K:]), _lib.ptr(ws),
              _lib.stream_of(X))
    assert torch.equal(again, raw)
