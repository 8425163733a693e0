"""GPU parity: fused EdgeConv block (csrc/edgeconv.hip) vs the plain-torch fp32 oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_model as R

pytestmark = pytest.mark.gpu


def _bf16_round(t, t16=torch.bfloat16):
    return t.to(t16).to(torch.float32)


def _served_as(dtype, C):
    """The operand type dgcnn._edgeconv_dtype picks for the supported widths of CASES: the IEEE-half kernels
    (csrc/edgeconv_fwd_f16.hip, BASELINE configs[4] "fp16+MFMA") start at 33 input channels, narrower layers keep bf16."""
    return "bf16" if (dtype == "f16" and C <= 32) else dtype


def _reference(x, idx, w, gamma, beta, G, bf16):
    """fp32 oracle.  For the 16-bit MFMA paths (bf16 = True or a torch 16-bit type) the oracle sees the SAME rounded
    operands the kernel contracts ([x_j ; x_i] against [W1 | W2-W1]); products of 16-bit values are exact in f32,
    so the only difference left is f32 summation order."""
    if not bf16:
        return R.edgeconv_block(x, idx, w, gamma, beta, G)
    t16 = torch.bfloat16 if bf16 is True else bf16
    C = x.shape[1]
    xr = _bf16_round(x, t16)
    w1, w2 = _bf16_round(w[:, :C], t16), _bf16_round(w[:, C:] - w[:, :C], t16)
    # W1.(x_j - x_i) + W2.x_i == W1.x_j + (W2-W1).x_i  -> feed the oracle an equivalent weight
    w_eq = torch.cat([w1, w2 + w1], 1)
    return R.edgeconv_block(xr, idx, w_eq, gamma, beta, G)


CASES = [  # B, C, N, k, Cout, G
    (2, 16, 96, 8, 64, 2), (2, 64, 300, 20, 64, 2), (1, 64, 257, 64, 128, 2), (2, 6, 200, 16, 64, 2),
    (1, 128, 130, 64, 128, 2), (1, 3, 100, 30, 64, 2), (1, 32, 90, 80, 128, 4), (1, 128, 64, 33, 64, 2),
    (1, 256, 96, 32, 128, 2),
]
_T16 = {"bf16": torch.bfloat16, "f16": torch.float16}


@pytest.mark.parametrize("B,C,N,k,Cout,G", CASES)
@pytest.mark.parametrize("dtype", ["bf16", "f32", "f16"])
def test_edgeconv_forward(dev, B, C, N, k, Cout, G, dtype):
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(B * 1000 + C + N + k)
    x = torch.randn(B, C, N, generator=g)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)])
    w = torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5
    gamma = torch.randn(Cout, generator=g)       # mixed signs: exercises max- and min-routing
    beta = torch.randn(Cout, generator=g) * 0.1
    r = dgcnn.edgeconv_forward_raw(x.to(dev), idx.to(dev), w.to(dev), gamma.to(dev), beta.to(dev), G, dtype, need_arg=True)
    t16 = _T16.get(_served_as(dtype, C))
    ref = _reference(x, idx, w, gamma, beta, G, t16 if t16 is not None else False)
    # tolerance: fp32 features within 1e-4 (north star); 16-bit paths compared on identical rounded operands
    np.testing.assert_allclose(r["out"].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    # raw extremes + arg slots vs a direct evaluation
    xr, wr = (x, w) if dtype == "f32" else (_bf16_round(x, t16), None)
    ef = R.get_graph_feature(xr, idx=idx)                      # (B,2C,N,k)
    if dtype == "f32":
        y = torch.einsum("oc,bcnk->bnko", w, ef)
    else:
        w1, w2 = _bf16_round(w[:, :C], t16), _bf16_round(w[:, C:] - w[:, :C], t16)
        y = torch.einsum("oc,bcnk->bnko", torch.cat([w1, w2 + w1], 1), ef)
    np.testing.assert_allclose(r["ymax"].cpu().numpy(), y.max(2)[0].numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(r["ymin"].cpu().numpy(), y.min(2)[0].numpy(), rtol=1e-4, atol=1e-4)
    am = r["amax"].cpu().long()
    picked = torch.gather(y, 2, am.unsqueeze(2)).squeeze(2)
    np.testing.assert_allclose(picked.numpy(), y.max(2)[0].numpy(), rtol=1e-4, atol=1e-4)
    an = r["amin"].cpu().long()
    picked = torch.gather(y, 2, an.unsqueeze(2)).squeeze(2)
    np.testing.assert_allclose(picked.numpy(), y.min(2)[0].numpy(), rtol=1e-4, atol=1e-4)
    assert int(am.max()) < k and int(an.max()) < k
    # GroupNorm statistics
    cnt = (Cout // G) * N * k
    yg = y.permute(0, 3, 1, 2).reshape(B, G, -1).double()
    np.testing.assert_allclose(r["gsum"].cpu().numpy()[..., 0] / cnt, yg.mean(-1).numpy(), rtol=1e-4, atol=1e-5)


def test_edgeconv_forward_golden(dev, golden):
    """vs the output of the reference's own modules (tests/golden/make_golden.py, section 3)."""
    from gcanet_amd import dgcnn
    g = golden
    t = lambda a: torch.from_numpy(a).to(dev)
    r = dgcnn.edgeconv_forward_raw(t(g["ec_x"]), t(g["ec_idx"]), t(g["ec_w"]), t(g["ec_gamma"]), t(g["ec_beta"]), 2, "f32")
    np.testing.assert_allclose(r["out"].cpu().numpy(), g["ec_y"], rtol=1e-4, atol=1e-4)


def test_edgeconv_backward_golden(dev, golden):
    """Gradients vs the reference's autograd through its own modules (golden section 3)."""
    from gcanet_amd import dgcnn
    g = golden
    t = lambda a: torch.from_numpy(a).to(dev)
    x, w = t(g["ec_x"]).requires_grad_(), t(g["ec_w"]).requires_grad_()
    ga, be = t(g["ec_gamma"]).requires_grad_(), t(g["ec_beta"]).requires_grad_()
    y = dgcnn.edge_conv(x, t(g["ec_idx"]), w, ga, be, 2, "f32")
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["ec_y"], rtol=1e-4, atol=1e-4)
    (y * t(g["ec_gout"])).sum().backward()
    for got, key in ((x.grad, "ec_dx"), (w.grad, "ec_dw"), (ga.grad, "ec_dgamma"), (be.grad, "ec_dbeta")):
        ref = g[key]                                   # sums over up to N*k terms: atol relative to the tensor's scale
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max(), err_msg=key)


@pytest.mark.parametrize("B,C,N,k,Cout,G", [(2, 64, 300, 20, 64, 2), (1, 128, 200, 64, 128, 2), (2, 6, 128, 16, 64, 2)])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_edgeconv_backward_vs_oracle_autograd(dev, B, C, N, k, Cout, G, dtype):
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(7 + C + N)
    x = torch.randn(B, C, N, generator=g)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)])
    w = torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g) * 0.1
    gout = torch.randn(B, Cout, N, generator=g)
    if dtype == "bf16":   # feed both sides operands that are exactly representable in bf16
        x = _bf16_round(x)
        w1, wd = _bf16_round(w[:, :C]), _bf16_round(w[:, C:] - w[:, :C])
        w = torch.cat([w1, wd + w1], 1)
    leaves = [v.clone().requires_grad_() for v in (x, w, gamma, beta)]
    R.edgeconv_block(leaves[0], idx, leaves[1], leaves[2], leaves[3], G).mul(gout).sum().backward()
    dl = [v.clone().to(dev).requires_grad_() for v in (x, w, gamma, beta)]
    y = dgcnn.edge_conv(dl[0], idx.to(dev), dl[1], dl[2], dl[3], G, dtype)
    (y * gout.to(dev)).sum().backward()
    for a, b, name in zip(dl, leaves, ("dx", "dw", "dgamma", "dbeta")):
        ref = b.grad.numpy()
        scale = np.abs(ref).max()
        np.testing.assert_allclose(a.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * scale, err_msg=name)


def test_grouped_block_fwd_bwd(dev):
    """conv_normal-style block on a materialised 7-channel edge feature (M4:575-577,691-693)."""
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(3)
    B, N, k, F, Cout = 2, 150, 16, 7, 64
    ef = _bf16_round(torch.randn(B, N, k, F, generator=g))
    w = _bf16_round(torch.randn(Cout, F, generator=g) / F ** 0.5)
    gamma, beta = torch.randn(Cout, generator=g), torch.randn(Cout, generator=g) * 0.1
    gout = torch.randn(B, Cout, N, generator=g)
    leaves = [v.clone().requires_grad_() for v in (ef, w, gamma, beta)]
    R.grouped_block(leaves[0].permute(0, 3, 1, 2), leaves[1], leaves[2], leaves[3], 2).mul(gout).sum().backward()
    ref = R.grouped_block(ef.permute(0, 3, 1, 2), w, gamma, beta, 2)
    dl = [v.clone().to(dev).requires_grad_() for v in (ef, w, gamma, beta)]
    y = dgcnn.grouped_block(dl[0], dl[1], dl[2], dl[3], 2)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    (y * gout.to(dev)).sum().backward()
    for a, b, name in zip(dl, leaves, ("d_ef", "dw", "dgamma", "dbeta")):
        r = b.grad.numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * np.abs(r).max(), err_msg=name)


def test_graph_feature_functions_match_golden(dev, golden):
    from gcanet_amd import dgcnn
    t = lambda a: torch.from_numpy(a).to(dev)
    f = dgcnn.get_graph_feature(t(golden["knn_feat_x"]), idx=t(golden["knn_feat_idx_k8"]))
    np.testing.assert_array_equal(f.cpu().numpy(), golden["ggf_feat_out"])
    xp, ip = t(golden["knnpn_rand_x"]), t(golden["knnpn_rand_idx_k16"])
    np.testing.assert_array_equal(dgcnn.get_graph_feature_with_normals(xp, idx=ip).cpu().numpy(), golden["ggfn_out"])
    np.testing.assert_allclose(dgcnn.get_graph_feature_with_normals_g(xp, idx=ip).cpu().numpy(), golden["ggfng_out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("C,Cout,G", [(6, 96, 2), (20, 48, 3), (64, 256, 2), (160, 64, 2)])
def test_edge_conv_any_width_routes_to_the_exact_kernel(dev, C, Cout, G):
    """M4:493-505 takes arbitrary channel counts; widths the bf16 matrix-core kernel does not serve (Cout not in
    {64,128}, Cout/G % 32 != 0, C > 128) run on the exact f32 kernel behind the same `edge_conv(..., dtype="bf16")`
    call: forward and gradients vs the oracle at 1e-4."""
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(C * 7 + Cout)
    B, N, k = 2, 200, 12
    x = torch.randn(B, C, N, generator=g)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)])
    w = torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5
    ga, be = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    xd, wd, gd, bd = (t.clone().to(dev).requires_grad_(True) for t in (x, w, ga, be))
    out = dgcnn.edge_conv(xd, idx.to(dev), wd, gd, bd, groups=G, dtype="bf16")
    go = torch.randn(B, Cout, N, generator=g)
    out.backward(go.to(dev))
    xr, wr, gr, br = (t.clone().requires_grad_(True) for t in (x, w, ga, be))
    ref = R.edgeconv_block(xr, idx, wr, gr, br, G)
    ref.backward(go)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    for a, b, name in ((xd, xr, "dx"), (wd, wr, "dW"), (gd, gr, "dgamma"), (bd, br, "dbeta")):
        r = b.grad.numpy()
        np.testing.assert_allclose(a.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-4 * max(1e-6, float(np.abs(r).max())), err_msg=name)
