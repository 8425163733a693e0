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


@pytest.mark.parametrize("M,N,K,ybf,xbf", [(65536, 10, 256, 1, 1), (65536, 22, 256, 1, 1), (65536, 3, 256, 1, 1),
                                           (65536, 30, 30, 1, 1), (65536, 3, 128, 0, 0), (1000, 32, 512, 0, 1),
                                           (777, 1, 1, 0, 0), (4099, 7, 20, 1, 0), (300, 16, 1024, 1, 1), (257, 5, 262, 0, 1)])
def test_wgrad_narrow(dev, M, N, K, ybf, xbf):
    """gcn_wgrad_narrow: dW = dY^T X and db = column sums for the narrow layers (N <= 32, any K <= 1024, bf16 / f32
    operands) vs f64 torch; fixed fold order -> a second run gives the same bits.  LinearPMFunction's backward goes
    through it for the 10-, 22-, 3-wide heads and KPAM's 30x30 layers (and, operands swapped, the K = 3 projection)."""
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    dY = torch.randn(M, N, generator=g)
    X = torch.randn(M, K, generator=g)
    dY = (dY.bfloat16() if ybf else dY).to(dev)
    X = (X.bfloat16() if xbf else X).to(dev)
    assert _lib.lib().gcn_wgrad_narrow_supported(M, N, K)
    ws = torch.empty(_lib.lib().gcn_wgrad_narrow_ws_bytes(M, N, K), dtype=torch.uint8, device=dev)
    raw = torch.full((N * K + N,), 7.0, device=dev)

    def run(out):
        _lib.call("gcn_wgrad_narrow", _lib.ptr(dY), ybf, _lib.ptr(X), xbf, M, N, K, _lib.ptr(out), _lib.ptr(out[N * K:]), _lib.ptr(ws),
                  _lib.stream_of(X))
    run(raw)
    ref = (dY.double().t() @ X.double()).cpu().numpy()
    dbr = dY.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(raw[:N * K].view(N, K).cpu().numpy(), ref, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()) + 1e-6)
    np.testing.assert_allclose(raw[N * K:].cpu().numpy(), dbr, rtol=1e-4, atol=2e-5 * float(np.abs(dbr).max()) + 1e-6)
    again = torch.empty_like(raw)
    run(again)
    assert torch.equal(again, raw)
    dW = torch.empty(N, K, device=dev)                          # without the bias gradient
    _lib.call("gcn_wgrad_narrow", _lib.ptr(dY), ybf, _lib.ptr(X), xbf, M, N, K, _lib.ptr(dW), None, _lib.ptr(ws), _lib.stream_of(X))
    assert torch.equal(dW.reshape(-1), raw[:N * K])


def test_linear_pm_narrow_layers_gradients(dev):
    """LinearPMFunction with narrow outputs / inputs under bf16 autocast and in f32: weight, bias and input gradients vs
    torch's own linear in f64 (the narrow weight-gradient kernel and its operand-swapped form sit behind these)."""
    from gcanet_amd.layers import linear_pm
    g = torch.Generator().manual_seed(5)
    for (M, N, K, auto) in ((8192, 10, 256, True), (8192, 22, 256, True), (8192, 3, 256, True), (8192, 30, 30, True),
                            (8192, 128, 3, False), (8192, 30, 30, False)):
        x = torch.randn(2, M // 2, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        b = torch.randn(N, generator=g) if K != 3 else None
        go = torch.randn(2, M // 2, N, generator=g)
        xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True) if b is not None else None
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=auto):
            y = linear_pm(xd, wd, bd)
        y.backward(go.to(dev).to(y.dtype))
        if auto:            # what the bf16 path sees: rounded operands, f32 accumulation
            xr, wr, gr = x.bfloat16().double(), w.bfloat16().double(), go.bfloat16().double()
        else:
            xr, wr, gr = x.double(), w.double(), go.double()
        dw = gr.reshape(-1, N).t() @ xr.reshape(-1, K)
        tol = 2e-2 if auto else 1e-4
        np.testing.assert_allclose(wd.grad.double().cpu().numpy(), dw.numpy(), rtol=tol, atol=tol * float(dw.abs().max()))
        if b is not None:
            db = gr.reshape(-1, N).sum(0)
            np.testing.assert_allclose(bd.grad.double().cpu().numpy(), db.numpy(), rtol=tol, atol=tol * float(db.abs().max()))
        dx = gr @ wr
        np.testing.assert_allclose(xd.grad.double().cpu().numpy(), dx.numpy(), rtol=tol, atol=tol * float(dx.abs().max()))
