"""GPU parity of the key-point edge block of the offset module (csrc/edgeconv.hip: keyedge_fwd_kernel / keyedge_bwd_kernel,
M4:398-452): max_k LeakyReLU(GroupNorm(att[n,j] * (U[m_j] - V[n]))) against the materialised (B,N,k,Cout) torch form,
forward and all four gradients, at channel counts on both sides of the two-channels-per-lane path, odd and even k,
k = 1 and k > 64 (generic backward)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(att, kidx, U, V, gamma, beta, G, eps, slope):
    B, N, k = att.shape
    Cout = U.shape[2]
    Ug = torch.gather(U.unsqueeze(1).expand(-1, N, -1, -1), 2, kidx.unsqueeze(-1).expand(-1, -1, -1, Cout))   # (B,N,k,Cout)
    y = att.unsqueeze(-1) * (Ug - V.unsqueeze(2))
    y = torch.nn.functional.group_norm(y.permute(0, 3, 1, 2), G, gamma, beta, eps)                             # (B,Cout,N,k)
    return torch.nn.functional.leaky_relu(y, slope).amax(-1)                                                     # (B,Cout,N)


@pytest.mark.parametrize("B,N,k,NK,Cout,G", [(2, 200, 30, 120, 128, 4), (2, 130, 29, 50, 128, 2), (1, 96, 1, 8, 64, 2),
                                               (2, 100, 7, 33, 96, 3), (1, 80, 64, 64, 256, 8), (1, 90, 70, 90, 64, 2)])
def test_key_edge_block_matches_materialised(dev, B, N, k, NK, Cout, G):
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(B * N + k + Cout)
    att = torch.rand(B, N, k, generator=g, dtype=torch.float64) + 0.1
    kidx = torch.stack([torch.stack([torch.randperm(NK, generator=g)[:k] if k <= NK else torch.randint(0, NK, (k,), generator=g)
                                     for _ in range(N)]) for _ in range(B)])
    U = torch.randn(B, NK, Cout, generator=g, dtype=torch.float64)
    V = torch.randn(B, N, Cout, generator=g, dtype=torch.float64)
    gamma = torch.randn(Cout, generator=g, dtype=torch.float64)          # both signs: the routed forward keeps max or min per channel
    beta = torch.randn(Cout, generator=g, dtype=torch.float64) * 0.1
    w = torch.randn(B, Cout, N, generator=g, dtype=torch.float64)
    ref_in = [t.clone().requires_grad_(True) for t in (att, U, V, gamma, beta)]
    ref = _reference(ref_in[0], kidx, ref_in[1], ref_in[2], ref_in[3], ref_in[4], G, 1e-5, 0.2)
    (ref * w).sum().backward()
    got_in = [t.float().to(dev).requires_grad_(True) for t in (att, U, V, gamma, beta)]
    out = dgcnn.KeyEdgeBlockFunction.apply(got_in[0], kidx.to(dev), got_in[1], got_in[2], got_in[3], got_in[4], G, 1e-5, 0.2)
    (out * w.float().to(dev)).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    for name, a, b in zip(("att", "U", "V", "gamma", "beta"), got_in, ref_in):
        scale = float(b.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-3, atol=2e-4 * scale, err_msg=name)


@pytest.mark.parametrize("B,N,k,NK,Cout,G", [(2, 150, 30, 120, 128, 4), (1, 70, 9, 20, 64, 2), (1, 40, 80, 100, 96, 3)])
def test_key_edge_forward_routed_equals_unrouted_extremes(dev, B, N, k, NK, Cout, G):
    """gcn_keyedge_fwd with gamma_route keeps, per channel, exactly the extreme (value and position) the two-sided call
    returns in ymax/amax (gamma >= 0) or ymin/amin (gamma < 0), and the same GroupNorm sums."""
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(k + Cout)
    att = (torch.rand(B, N, k, generator=g) + 0.1).to(dev)
    kidx = torch.randint(0, NK, (B, N, k), generator=g).to(dev)
    U, V = torch.randn(B, NK, Cout, generator=g).to(dev), torch.randn(B, N, Cout, generator=g).to(dev)
    gamma = torch.randn(Cout, generator=g).to(dev)
    f = lambda: torch.empty(B, N, Cout, device=dev)
    u8 = lambda: torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
    ymax, ymin, amax, amin, gs = f(), f(), u8(), u8(), torch.empty(B, G, 2, dtype=torch.float64, device=dev)
    _lib.call("gcn_keyedge_fwd", _lib.ptr(att), _lib.ptr(kidx), _lib.ptr(U), _lib.ptr(V), B, N, k, NK, Cout, G, _lib.ptr(ymax),
              _lib.ptr(ymin), _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gs), None, _lib.stream_of(att))
    yr, ar, gr = f(), u8(), torch.empty(B, G, 2, dtype=torch.float64, device=dev)
    _lib.call("gcn_keyedge_fwd", _lib.ptr(att), _lib.ptr(kidx), _lib.ptr(U), _lib.ptr(V), B, N, k, NK, Cout, G, _lib.ptr(yr),
              None, _lib.ptr(ar), None, _lib.ptr(gr), _lib.ptr(gamma), _lib.stream_of(att))
    pos = (gamma >= 0).view(1, 1, Cout)
    assert torch.equal(yr, torch.where(pos, ymax, ymin))
    assert torch.equal(ar, torch.where(pos, amax, amin))
    np.testing.assert_allclose(gr.cpu().numpy(), gs.cpu().numpy(), rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError, match="routed mode"):
        _lib.call("gcn_keyedge_fwd", _lib.ptr(att), _lib.ptr(kidx), _lib.ptr(U), _lib.ptr(V), B, N, k, NK, Cout, G, _lib.ptr(yr),
                  None, _lib.ptr(ar), None, _lib.ptr(gr), None, _lib.stream_of(att))
