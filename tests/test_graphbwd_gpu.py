"""Direct C-ABI tests of the graph-aggregation / weight-gradient kernels behind the closed-form EdgeConv backward
(edgeconv.hip: reverse_sum_lds_kernel, neighbor_sum_kernel; graphbwd.hip: edge_wgrad_kernel) against plain torch
restatements.  The end-to-end gradient parity vs the reference's autograd lives in test_edgeconv_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(B, N, k, C, dev, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, C, generator=g).to(dev)
    idx = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)]).to(dev)
    return x, idx


@pytest.mark.parametrize("B,N,k,C", [(2, 256, 16, 64), (1, 1000, 30, 6), (3, 333, 7, 128), (2, 2048, 64, 64), (1, 96, 80, 13),
                                      (2, 2048, 24, 256)])
def test_reverse_and_neighbor_sum(dev, B, N, k, C):
    from gcanet_amd import _lib
    x, idx = _graph(B, N, k, C, dev, B * 100 + N + k + C)
    x[0, 0] *= 1e3                                    # wide dynamic range inside one tensor
    r = torch.empty_like(x)
    s = torch.empty_like(x)
    indeg = torch.empty(B, N, device=dev)
    ws = torch.empty(_lib.lib().gcn_reverse_sum_ws_bytes(B, N, C, k), dtype=torch.uint8, device=dev)
    _lib.call("gcn_reverse_sum", _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(r), _lib.ptr(indeg), _lib.ptr(ws),
              _lib.stream_of(x))
    _lib.call("gcn_neighbor_sum", _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(s), _lib.stream_of(x))
    xd = x.double()
    r_ref = torch.zeros_like(xd)
    deg_ref = torch.zeros(B, N, dtype=torch.float64, device=dev)
    for b in range(B):
        r_ref[b].index_add_(0, idx[b].reshape(-1), xd[b].repeat_interleave(k, 0))
        deg_ref[b].index_add_(0, idx[b].reshape(-1), torch.ones(N * k, dtype=torch.float64, device=dev))
    s_ref = torch.stack([xd[b][idx[b]].sum(1) for b in range(B)])
    assert torch.equal(indeg.double(), deg_ref)
    # fixed-point accumulation is exact up to one final rounding: <= 1 ulp of the f64 reference rounded to f32
    assert (r.double() - r_ref).abs().max().item() <= 2e-7 * r_ref.abs().max().item() + 1e-30
    assert (s.double() - s_ref).abs().max().item() <= 1e-5 * s_ref.abs().max().item()
    r2 = torch.empty_like(x)
    _lib.call("gcn_reverse_sum", _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(r2), None, _lib.ptr(ws), _lib.stream_of(x))
    assert torch.equal(r, r2)                         # bitwise reproducible (integer sums)


@pytest.mark.parametrize("B,N,k,C,kind", [(2, 1024, 16, 64, "one_hub"), (1, 2048, 64, 64, "hub_partition"), (2, 512, 20, 128, "one_hub"),
                                          (1, 4096, 32, 128, "random"), (3, 8192, 8, 64, "random"),
                                          (1, 16384, 16, 256, "random"), (2, 4096, 32, 256, "one_hub"),
                                          (1, 2048, 64, 256, "hub_partition"), (1, 16384, 20, 128, "random")])
def test_reverse_sum_staged_path_degenerate_graphs(dev, B, N, k, C, kind):
    """The stage + sort + gather form (csrc/rsum.hip; C in {64,128,256}) on graphs that overflow a staging segment (every edge
    to one node) or the LDS sort list (every edge into one partition): the overflow list and the accumulate-in-LDS path
    must give the same sums, in-degrees included, bitwise reproducibly."""
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(N + k + C)
    x = torch.randn(B, N, C, generator=g).to(dev)
    if kind == "one_hub":
        idx = torch.full((B, N, k), 5, dtype=torch.int64)
    elif kind == "hub_partition":
        idx = torch.randint(0, 100, (B, N, k), generator=g)
    else:
        idx = torch.randint(0, N, (B, N, k), generator=g)
    idx = idx.to(dev)
    outs = []
    for _ in range(2):
        r = torch.full_like(x, float("nan"))
        indeg = torch.empty(B, N, device=dev)
        ws = torch.empty(_lib.lib().gcn_reverse_sum_ws_bytes(B, N, C, k), dtype=torch.uint8, device=dev)
        _lib.call("gcn_reverse_sum", _lib.ptr(x), _lib.ptr(idx), B, N, C, k, _lib.ptr(r), _lib.ptr(indeg), _lib.ptr(ws),
                  _lib.stream_of(x))
        outs.append((r, indeg))
    xd = x.double()
    r_ref = torch.zeros_like(xd)
    deg_ref = torch.zeros(B, N, dtype=torch.float64, device=dev)
    for b in range(B):
        r_ref[b].index_add_(0, idx[b].reshape(-1), xd[b].repeat_interleave(k, 0))
        deg_ref[b].index_add_(0, idx[b].reshape(-1), torch.ones(N * k, dtype=torch.float64, device=dev))
    r, indeg = outs[0]
    assert torch.equal(indeg.double(), deg_ref)
    assert (r.double() - r_ref).abs().max().item() <= 2e-7 * r_ref.abs().max().item() + 1e-30
    assert torch.equal(outs[0][0], outs[1][0])


@pytest.mark.parametrize("B,N,C,Cout", [(2, 512, 64, 128), (3, 300, 6, 64), (1, 1024, 64, 64), (2, 130, 16, 128)])
def test_edge_wgrad(dev, B, N, C, Cout):
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(N + C + Cout)
    x, s = (torch.randn(B, N, C, generator=g).to(dev) for _ in range(2))
    dsp, d2 = (torch.randn(B, N, Cout, generator=g).to(dev) for _ in range(2))
    indeg = torch.randint(0, 9, (B, N), generator=g).float().to(dev)
    W = torch.randn(Cout, 2 * C, generator=g).to(dev)
    Ac, Bc = (torch.randn(B, Cout, generator=g).to(dev) for _ in range(2))
    dW = torch.empty(Cout, 2 * C, device=dev)
    ws = torch.empty(_lib.lib().gcn_edge_wgrad_ws_floats(B, C, Cout), device=dev)
    _lib.call("gcn_edge_wgrad", _lib.ptr(x), _lib.ptr(s), _lib.ptr(dsp), _lib.ptr(d2), _lib.ptr(indeg), _lib.ptr(W),
              _lib.ptr(Ac), _lib.ptr(Bc), B, N, C, Cout, _lib.ptr(dW), _lib.ptr(ws), _lib.stream_of(x))
    xd, sd, pd, qd, W_, A_, B_ = (t.double() for t in (x, s, dsp, d2, W, Ac, Bc))
    W1, Wd = W_[:, :C], W_[:, C:] - W_[:, :C]
    G11 = torch.einsum("bnc,bn,bnd->bcd", xd, indeg.double(), xd)
    G21 = torch.einsum("bnc,bnd->bcd", xd, sd)
    dW1 = torch.einsum("bno,bnc->oc", pd, xd) + torch.einsum("bo,bc->oc", A_, sd.sum(1)) \
        + torch.einsum("bo,boc->oc", B_, torch.einsum("oc,bcd->bod", W1, G11) + torch.einsum("oc,bcd->bod", Wd, G21))
    dWd = torch.einsum("bno,bnc->oc", qd, xd)
    ref = torch.cat([dW1 - dWd, dWd], 1)
    assert (dW.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_edge_wgrad_rejects_unsupported(dev):
    from gcanet_amd import _lib
    t = torch.zeros(8, device=dev)
    with pytest.raises(RuntimeError):
        _lib.call("gcn_edge_wgrad", _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t),
                  _lib.ptr(t), 1, 1, 32, 64, _lib.ptr(t), _lib.ptr(t), _lib.stream_of(t))


@pytest.mark.parametrize("B,N,k", [(2, 300, 16), (1, 257, 80), (2, 64, 64), (2, 200, 31), (1, 100, 1), (1, 300, 129)])
def test_normal_edge_block_matches_materialised(dev, B, N, k):
    """Fused normal-feature EdgeConv (csrc/normaledge.hip) vs the same block on the materialised (B,N,k,7) edge
    feature of get_graph_feature_with_normals_g (M4:164-205) through torch autograd, f32."""
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(N + k)
    xyz = torch.rand(B, N, 3, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
    pts = torch.cat([xyz, nrm], -1).to(dev)
    idx = dgcnn.knn_points_normals(pts.transpose(1, 2).contiguous(), k, k)
    W = (0.3 * torch.randn(64, 7, 1, 1, generator=g)).to(dev).requires_grad_(True)
    gn = torch.nn.GroupNorm(2, 64).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(64, generator=g))          # both signs: max- and min-routed channels
        gn.bias.copy_(torch.randn(64, generator=g))
    go = torch.randn(B, N, 64, generator=g).to(dev)
    out = dgcnn.normal_edge_block(pts, idx, W, gn.weight, gn.bias, 2, gn.eps, 0.2, pm_out=True)
    g1 = torch.autograd.grad(out, (W, gn.weight, gn.bias), go)
    ef = dgcnn.get_graph_feature_with_normals_g(pts.transpose(1, 2).contiguous(), k, k, idx)      # (B,7,N,k)
    y = torch.nn.functional.conv2d(ef, W)
    y = torch.nn.functional.leaky_relu(gn(y), 0.2).max(dim=-1)[0].permute(0, 2, 1)
    g2 = torch.autograd.grad(y, (W, gn.weight, gn.bias), go)
    assert (out - y).abs().max().item() < 1e-4
    for a, b in zip(g1, g2):
        assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1.0)


@pytest.mark.parametrize("B,N,k,Cout", [(2, 300, 16, 64), (1, 200, 33, 96), (1, 260, 200, 128)])
def test_normal_edge_forward_routed_equals_unrouted_extremes(dev, B, N, k, Cout):
    """gcn_normal_edge_fwd with gamma_route keeps exactly the extreme (value, position) the two-sided call returns in
    ymax/amax (gamma >= 0) or ymin/amin (gamma < 0), and the same GroupNorm sums."""
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(N + k)
    pts = torch.cat([torch.rand(B, N, 3, generator=g), torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)], -1).to(dev)
    idx = torch.randint(0, N, (B, N, k), generator=g).to(dev)
    W = (0.3 * torch.randn(Cout, 7, generator=g)).to(dev)
    gamma = torch.randn(Cout, generator=g).to(dev)
    G = 2 if Cout != 96 else 3
    f = lambda: torch.empty(B, N, Cout, device=dev)
    u8 = lambda: torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
    ymax, ymin, amax, amin, gs = f(), f(), u8(), u8(), torch.empty(B, G, 2, dtype=torch.float64, device=dev)
    _lib.call("gcn_normal_edge_fwd", _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(W), B, N, k, Cout, G, _lib.ptr(ymax), _lib.ptr(ymin),
              _lib.ptr(amax), _lib.ptr(amin), _lib.ptr(gs), None, _lib.stream_of(pts))
    yr, ar, gr = f(), u8(), torch.empty(B, G, 2, dtype=torch.float64, device=dev)
    _lib.call("gcn_normal_edge_fwd", _lib.ptr(pts), _lib.ptr(idx), _lib.ptr(W), B, N, k, Cout, G, _lib.ptr(yr), None,
              _lib.ptr(ar), None, _lib.ptr(gr), _lib.ptr(gamma), _lib.stream_of(pts))
    pos = (gamma >= 0).view(1, 1, Cout)
    assert torch.equal(yr, torch.where(pos, ymax, ymin))
    assert torch.equal(ar, torch.where(pos, amax, amin))
    np.testing.assert_allclose(gr.cpu().numpy(), gs.cpu().numpy(), rtol=1e-6, atol=1e-6)


def test_new_entry_points_handle_empty_and_reject_bad_shapes(dev):
    """Empty batches are a no-op (GCN_OK), impossible shapes raise RuntimeError (status int -> exception, never exit)."""
    from gcanet_amd import _lib
    t = torch.zeros(64, device=dev)
    i64 = torch.zeros(64, dtype=torch.int64, device=dev)
    u8 = torch.zeros(64, dtype=torch.uint8, device=dev)
    d = torch.zeros(64, dtype=torch.float64, device=dev)
    st = _lib.stream_of(t)
    _lib.call("gcn_reverse_sum", _lib.ptr(t), _lib.ptr(i64), 0, 8, 4, 2, _lib.ptr(t), _lib.ptr(t), _lib.ptr(u8), st)
    _lib.call("gcn_topk_rows", _lib.ptr(t), 0, 0, 8, 4, _lib.ptr(t), _lib.ptr(i64), st)
    _lib.call("gcn_normal_edge_fwd", _lib.ptr(t), _lib.ptr(i64), _lib.ptr(t), 0, 8, 4, 8, 2, _lib.ptr(t), _lib.ptr(t),
              _lib.ptr(u8), _lib.ptr(u8), _lib.ptr(d), None, st)
    _lib.call("gcn_param_normalise_fwd", _lib.ptr(t), 0, _lib.ptr(t), st)
    _lib.call("gcn_attention_fwd_bf16", _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, 0, 0, 4, 4, 32, 1.0, _lib.ptr(t),
              _lib.ptr(t), _lib.ptr(u8), st)
    with pytest.raises(RuntimeError):
        _lib.call("gcn_topk_rows", _lib.ptr(t), 0, 1, 200, 4, _lib.ptr(t), _lib.ptr(i64), st)            # NK > 128
    with pytest.raises(RuntimeError):
        _lib.call("gcn_attention_fwd_bf16", _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, 0, 1, 4, 4, 48, 1.0,
                  _lib.ptr(t), _lib.ptr(t), _lib.ptr(u8), st)                                               # head dim 48
    with pytest.raises(RuntimeError):
        _lib.call("gcn_normal_edge_fwd", _lib.ptr(t), _lib.ptr(i64), _lib.ptr(t), 1, 8, 300, 8, 2, _lib.ptr(t), _lib.ptr(t),
                  _lib.ptr(u8), _lib.ptr(u8), _lib.ptr(d), None, st)                                        # k > 256
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,N,k,Cout,kind", [(2, 1024, 16, 64, "random"), (3, 512, 20, 128, "random"), (1, 2048, 64, 128, "random"),
                                            (2, 1024, 8, 64, "one_hub"), (1, 512, 8, 128, "few_hubs"),
                                            (1, 4096, 8, 64, "one_hub"), (1, 8192, 4, 128, "few_hubs")])
def test_route_bwd_lds_scatter_matches_definition(dev, B, N, k, Cout, kind):
    """gcn_route_bwd's sparse scatter Dsp[b, idx[b,n,jsel[n,c]], c] += coef[b,n,c] through the partition-staged LDS sum
    (dsp_ws given) == the f32-atomics path (dsp_ws NULL) == an f64 index_add of the returned coef, including the
    degenerate graphs where every point selects the same neighbour (one accumulator takes N addends; at N >= 4096 a
    tile files more than DSP_CAP entries under one partition and the overflow list is used).  Fixed-point sums: bitwise reproducible run to run."""
    from gcanet_amd import _lib
    from gcanet_amd.layers import _acc_buffers
    g = torch.Generator().manual_seed(B * 1000 + N + Cout + k)
    G = 2
    dout = torch.randn(B, N, Cout, generator=g).to(dev)
    dout[0, 0] *= 300.0
    ymax = torch.randn(B, N, Cout, generator=g).to(dev)
    amax = torch.randint(0, k, (B, N, Cout), generator=g, dtype=torch.uint8).to(dev)
    gamma = (torch.rand(Cout, generator=g) + 0.5).to(dev)
    beta = torch.randn(Cout, generator=g).to(dev)
    mean_rstd = torch.stack([torch.randn(B, G, generator=g), torch.rand(B, G, generator=g) + 0.5], -1).to(dev).contiguous()
    if kind == "random":
        idx = torch.randint(0, N, (B, N, k), generator=g)
    elif kind == "one_hub":
        idx = torch.full((B, N, k), 7, dtype=torch.int64)
    else:
        idx = torch.randint(0, 3, (B, N, k), generator=g) * (N // 3)
    idx = idx.to(dev)

    def run(with_ws):
        coef = torch.empty(B, N, Cout, device=dev)
        dsp = torch.full((B, N, Cout), float("nan"), device=dev)
        Ac, Bc = torch.empty(B, Cout, device=dev), torch.empty(B, Cout, device=dev)
        part = torch.empty(_lib.lib().gcn_route_bwd_part_bytes(B, N, Cout, G), dtype=torch.uint8, device=dev)
        if with_ws:
            S, dg, db, ws = _acc_buffers(B * G * 2, Cout, dev, tail_bytes=_lib.lib().gcn_route_bwd_ws_bytes(B, N, Cout))
        else:
            (S, dg, db), ws = _acc_buffers(B * G * 2, Cout, dev), None
        _lib.call("gcn_route_bwd", _lib.ptr(dout), _lib.ptr(ymax), None, _lib.ptr(amax), None, _lib.ptr(gamma), _lib.ptr(beta),
                  _lib.ptr(mean_rstd), _lib.ptr(idx), B, N, k, Cout, G, 0.2, _lib.ptr(coef), None, None, _lib.ptr(dsp),
                  _lib.ptr(dg), _lib.ptr(db), _lib.ptr(S), float((Cout // G) * N * k), _lib.ptr(Ac), _lib.ptr(Bc),
                  _lib.ptr(ws), _lib.ptr(part) if with_ws else None, _lib.stream_of(dout))
        return coef, dsp, dg.clone(), db.clone(), S.clone()

    coef1, dsp1, dg1, db1, S1 = run(True)
    coef0, dsp0, dg0, db0, S0 = run(False)
    assert torch.equal(coef1, coef0)
    msel = torch.gather(idx, 2, amax.long())                                   # (B,N,Cout)
    ref = torch.zeros(B, N, Cout, dtype=torch.float64, device=dev)
    ref.scatter_add_(1, msel, coef1.double())
    scale = ref.abs().max().item()
    assert torch.isfinite(dsp1).all()
    assert (dsp1.double() - ref).abs().max().item() <= 1e-7 * scale + 1e-30     # one final rounding to f32
    assert (dsp0.double() - ref).abs().max().item() <= 1e-4 * scale + 1e-30     # f32 atomics in arbitrary order
    for a, b in ((dg1, dg0), (db1, db0), (S1.float(), S0.float())):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    assert torch.equal(run(True)[1], dsp1)                                     # bitwise reproducible
