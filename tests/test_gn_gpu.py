"""GPU parity: point-major GroupNorm(+ReLU) (csrc/gn.hip) vs torch.nn.functional.group_norm in fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,C,G", [(2, 300, 64, 2), (1, 1000, 256, 4), (2, 513, 512, 8), (1, 128, 1024, 8), (2, 77, 128, 4), (1, 40, 3072, 8)])
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_group_norm_relu_fwd_bwd(dev, B, N, C, G, relu, dtype):
    from gcanet_amd.layers import GroupNormReLUFunction
    g = torch.Generator().manual_seed(B + N + C)
    x = (torch.randn(B, N, C, generator=g) * 2 + 0.5).to(dtype).float()
    gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g) * 0.3
    gy = torch.randn(B, N, C, generator=g).to(dtype).float()
    xs = [v.clone().requires_grad_() for v in (x, gamma, beta)]
    ref = F.group_norm(xs[0].permute(0, 2, 1), G, xs[1], xs[2], 1e-5)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 1)
    (ref * gy).sum().backward()
    xd = [x.to(dtype).to(dev).requires_grad_(), gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()]
    y = GroupNormReLUFunction.apply(xd[0], xd[1], xd[2], G, 1e-5, relu)
    assert y.dtype == dtype
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), ref.detach().numpy(), **tol)
    (y.float() * gy.to(dev)).sum().backward()
    for a, b, name in zip(xd, xs, ("dx", "dgamma", "dbeta")):
        r = b.grad.numpy()
        t2 = dict(rtol=1e-3, atol=1e-4 * max(1.0, np.abs(r).max())) if dtype == torch.float32 else \
            dict(rtol=5e-2, atol=2e-2 * max(1.0, np.abs(r).max()))
        np.testing.assert_allclose(a.grad.float().cpu().numpy(), r, err_msg=name, **t2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,C,G", [(2, 300, 64, 2), (3, 1000, 1024, 8), (1, 17, 256, 4), (2, 64, 2048, 1), (2, 40, 2048, 4), (1, 33, 3072, 8)])
def test_group_norm_relu_max_matches_two_step(dev, dtype, B, N, C, G):
    """Fused GN+ReLU+max-over-points equals group_norm_relu followed by max(dim=1), values and gradients."""
    from gcanet_amd import layers
    g = torch.Generator().manual_seed(B * N + C)
    x = torch.randn(B, N, C, generator=g).to(dev).to(dtype)
    gn = torch.nn.GroupNorm(G, C).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
    go = torch.randn(B, C, generator=g).to(dev).to(dtype)
    xa = x.clone().requires_grad_(True)
    a = layers.group_norm_relu_max(xa, gn)
    ga = torch.autograd.grad(a, (xa, gn.weight, gn.bias), go)
    xb = x.clone().requires_grad_(True)
    b = layers.group_norm_relu(xb, gn).max(dim=1)[0]
    gb = torch.autograd.grad(b, (xb, gn.weight, gn.bias), go)
    assert torch.equal(a, b)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    for u, v in zip(ga, gb):
        assert (u.float() - v.float()).abs().max().item() <= tol * max(v.float().abs().max().item(), 1.0)


def test_param_normalise_matches_torch_chain(dev):
    """csrc/heads.hip vs the reference's slice / norm / div / cat chain (M4:664-676), values and gradient."""
    from gcanet_amd import layers
    g = torch.Generator().manual_seed(4)
    p = torch.randn(3, 500, 22, generator=g).to(dev)
    p[0, 0, 4:7] = 0.0                                              # a zero triple: 0 / 1e-12
    go = torch.randn(3, 500, 22, generator=g).to(dev)
    unit = lambda v: v / (torch.norm(v, dim=-1, keepdim=True).repeat(1, 1, 3) + 1e-12)
    pa = p.clone().requires_grad_(True)
    a = layers.param_normalise(pa)
    (ga,) = torch.autograd.grad(a, pa, go)
    pb = p.clone().requires_grad_(True)
    b = torch.cat([pb[:, :, :4], unit(pb[:, :, 4:7]), pb[:, :, 7:8], unit(pb[:, :, 8:11]), pb[:, :, 11:15],
                   unit(pb[:, :, 15:18]), pb[:, :, 18:22]], dim=2)
    (gb,) = torch.autograd.grad(b, pb, go)
    np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    mask = torch.ones_like(p, dtype=torch.bool)
    mask[0, 0, 4:7] = False                                         # torch's norm backward is NaN at exactly zero
    np.testing.assert_allclose(ga[mask].cpu().numpy(), gb[mask].cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert torch.isfinite(ga).all()


@pytest.mark.parametrize("shape", [(2, 300, 64), (1, 5, 3), (3, 70, 200), (4096, 64)])
def test_row_normalise_matches_torch(dev, shape):
    """x / x.norm(dim=-1, keepdim=True) (the feature normalisation of cos_dist, M4:326-342) as one kernel each way
    (csrc/heads.hip) vs the torch expression in f64, forward and gradient."""
    from gcanet_amd.layers import row_normalise
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g, dtype=torch.float64)
    w = torch.randn(*shape, generator=g, dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    ref = xr / xr.norm(dim=-1, keepdim=True)
    (ref * w).sum().backward()
    xg = x.float().to(dev).requires_grad_(True)
    out = row_normalise(xg)
    (out * w.float().to(dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))


@pytest.mark.parametrize("shapes", [[(8, 8192, 10), (8, 8192, 22), (65536, 10), (65536, 3), (8, 8192, 64)],
                                    [(5,)], [(3, 7), (16385,), (2, 16384)]])
def test_sum_mean_squares_matches_torch(dev, shapes):
    """sum_t mean(v_t^2) over mixed f32 / bf16 tensors in one launch each way (csrc/heads.hip: the benchmark's synthetic
    objective) vs the per-tensor torch expression in f64: value, gradients, repeatability (fixed fold order), and a
    scaled incoming gradient."""
    from gcanet_amd.losses import sum_mean_squares
    g = torch.Generator().manual_seed(len(shapes))
    vs, refs = [], []
    for i, sh in enumerate(shapes):
        x = torch.randn(*sh, generator=g)
        if i % 2 == 1:
            x = x.bfloat16()
        vs.append(x.to(dev).requires_grad_(True))
        refs.append(x.double().requires_grad_(True))
    loss = sum_mean_squares(*vs)
    (loss * 3.0).backward()
    ref = sum(r.pow(2).mean() for r in refs)
    (ref * 3.0).backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * float(ref.detach())
    assert float(sum_mean_squares(*[v.detach() for v in vs])) == float(loss.detach())
    for v, r in zip(vs, refs):
        assert v.grad.dtype == v.dtype
        tol = 1e-2 if v.dtype == torch.bfloat16 else 1e-6
        np.testing.assert_allclose(v.grad.float().cpu().numpy(), r.grad.float().numpy(), rtol=tol, atol=1e-12)
