"""Sparse convolutions + the instance tiny U-Net (gcanet_amd/sparseconv.py, csrc/sparseconv.hip) against their dense
definition (oracle/ref_model.py: conv3d / conv_transpose3d on the densified grid).  The reference takes these ops from
the un-vendored spconv package and has no fixtures for them: parity unpinned by reference data, pinned to the operator
definition.  fp32, tolerance 1e-4 (relative to the tensor's scale: the summation order differs from the dense conv)."""
import numpy as np
import pytest
import torch

from oracle import ref_model as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _scene(seed, batch, D, M, C):
    g = torch.Generator().manual_seed(seed)
    # thin random "surfaces": pick M distinct cells per sample near a few planes so that neighbourhoods are populated
    idx = []
    for b in range(batch):
        cells = torch.stack(torch.meshgrid(*[torch.arange(D)] * 3, indexing="ij"), -1).view(-1, 3)
        pl = torch.randn(3, generator=g)
        d = (cells.float() - D / 2) @ (pl / pl.norm())
        cand = cells[(d.abs() < 1.2)]
        sel = cand[torch.randperm(cand.shape[0], generator=g)[:M]]          # arbitrary order inside a sample,
        idx.append(torch.cat([torch.full((sel.shape[0], 1), b), sel], 1))   # samples contiguous: what clusters_voxelization
    idx = torch.cat(idx).int()                                              # gives and global_pool (M4:1358-1370) relies on
    feats = torch.randn(idx.shape[0], C, generator=g)
    return feats, idx


def _close(a, b, what):
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max()) / scale
    assert err < TOL, "%s: max rel-to-scale error %.3g" % (what, err)


@pytest.mark.parametrize("Cin,Cout", [(64, 64), (128, 64), (64, 128)])
def test_subm_conv_forward_backward(dev, Cin, Cout):
    from gcanet_amd import sparseconv as S
    batch, D = 3, 12
    feats, idx = _scene(0, batch, D, 150, Cin)
    conv = S.SubMConv3d(Cin, Cout, "k").to(dev)
    x = feats.to(dev).requires_grad_(True)
    y = conv(S.SparseConvTensor(x, idx.to(dev), [D] * 3, batch)).features
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(gy.to(dev))
    xr = feats.clone().requires_grad_(True)
    wr = conv.weight.detach().cpu().clone().requires_grad_(True)
    yr = R.subm_conv3(xr, idx, D, batch, wr)
    yr.backward(gy)
    _close(y.detach().cpu(), yr.detach(), "forward")
    _close(x.grad.cpu(), xr.grad, "input gradient")
    _close(conv.weight.grad.cpu(), wr.grad, "weight gradient")


def test_rule_tables(dev):
    from gcanet_amd import sparseconv as S
    batch, D = 2, 9                                    # odd extent: the last coarse cell is half empty
    feats, idx = _scene(1, batch, D, 120, 64)
    x = S.SparseConvTensor(feats.to(dev), idx.to(dev), [D] * 3, batch)
    nbr = S.subm_rules(x).cpu().numpy()
    pos = {tuple(r): i for i, r in enumerate(idx.tolist())}
    for i in range(0, idx.shape[0], 7):
        b, xx, yy, zz = idx[i].tolist()
        for k in range(27):
            want = pos.get((b, xx + k // 9 - 1, yy + (k // 3) % 3 - 1, zz + k % 3 - 1), -1)
            assert nbr[i, k] == want
    coords2, child, parent = S.coarse_rules(x)
    ref2 = R.coarse_sites(idx)
    np.testing.assert_array_equal(coords2.cpu().numpy(), ref2.numpy())
    child, parent = child.cpu().numpy(), parent.cpu().numpy()
    assert (child >= 0).sum() == idx.shape[0] and (parent >= 0).sum() == idx.shape[0]
    for i in range(idx.shape[0]):
        b, xx, yy, zz = idx[i].tolist()
        k = (xx & 1) * 4 + (yy & 1) * 2 + (zz & 1)
        o = parent[i, k]
        assert child[o, k] == i and ref2[o].tolist() == [b, xx // 2, yy // 2, zz // 2]


def test_strided_and_inverse_conv(dev):
    from gcanet_amd import sparseconv as S
    batch, D = 3, 12
    feats, idx = _scene(2, batch, D, 160, 64)
    down, up = S.SparseConv3d(64, 128, "s").to(dev), S.SparseInverseConv3d(128, 64, "s").to(dev)
    x = feats.to(dev).requires_grad_(True)
    mid = down(S.SparseConvTensor(x, idx.to(dev), [D] * 3, batch))
    out = up(mid)
    assert torch.equal(out.indices.cpu(), idx)
    g = torch.randn(out.features.shape, generator=torch.Generator().manual_seed(6))
    out.features.backward(g.to(dev))
    xr = feats.clone().requires_grad_(True)
    wd = down.weight.detach().cpu().clone().requires_grad_(True)
    wu = up.weight.detach().cpu().clone().requires_grad_(True)
    f2, idx2 = R.strided_conv2(xr, idx, D, batch, wd)
    outr = R.inverse_conv2(f2, idx2, D, batch, wu, idx)
    outr.backward(g)
    np.testing.assert_array_equal(mid.indices.cpu().numpy(), idx2.numpy())
    _close(mid.features.detach().cpu(), f2.detach(), "strided forward")
    _close(out.features.detach().cpu(), outr.detach(), "inverse forward")
    _close(x.grad.cpu(), xr.grad, "input gradient")
    _close(down.weight.grad.cpu(), wd.grad, "strided weight gradient")
    _close(up.weight.grad.cpu(), wu.grad, "inverse weight gradient")


def test_instance_head_matches_dense_unet(dev):
    """forward_instance (M4:1379-1392): tiny U-Net + output layer + heads, fwd and parameter gradients."""
    from gcanet_amd import sparseconv as S
    torch.manual_seed(0)
    batch, D = 4, 16
    feats, idx = _scene(3, batch, D, 220, 64)
    head = S.InstanceHead(64, 10).to(dev)
    inst_map = torch.randint(0, idx.shape[0], (500,), generator=torch.Generator().manual_seed(7))
    x = feats.to(dev).requires_grad_(True)
    bidx, cls, iou, mask = head(S.SparseConvTensor(x, idx.to(dev), [D] * 3, batch), inst_map.to(dev))
    loss = cls.pow(2).mean() + iou.pow(2).mean() + mask.pow(2).mean()
    loss.backward()
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in head.state_dict().items()}
    xr = feats.clone().requires_grad_(True)
    f = R.tiny_unet(sd, xr, idx, D, batch)
    f = torch.relu(torch.nn.functional.batch_norm(f, None, None, sd["tiny_unet_outputlayer.0.weight"],
                                                   sd["tiny_unet_outputlayer.0.bias"], True, 0.1, 1e-4))
    lin = lambda t, p: t @ sd[p + ".weight"].t() + sd[p + ".bias"]
    maskr = lin(torch.relu(lin(f, "mask_linear.0")), "mask_linear.2")[inst_map]
    pooled = torch.stack([f[idx[:, 0] == b].mean(0) for b in range(batch)])
    clsr, iour = lin(pooled, "cls_linear"), lin(pooled, "iou_score_linear")
    lossr = clsr.pow(2).mean() + iour.pow(2).mean() + maskr.pow(2).mean()
    lossr.backward()
    np.testing.assert_array_equal(bidx.cpu().numpy(), idx[:, 0][inst_map].numpy())
    _close(cls.detach().cpu(), clsr.detach(), "cls scores")
    _close(iou.detach().cpu(), iour.detach(), "iou scores")
    _close(mask.detach().cpu(), maskr.detach(), "mask scores")
    _close(x.grad.cpu(), xr.grad, "input gradient")
    for name, p in head.named_parameters():
        if sd[name].grad is not None:
            scale = float(sd[name].grad.abs().max()) + 1e-12
            err = float((p.grad.cpu() - sd[name].grad).abs().max()) / scale
            assert err < 2e-3, "%s gradient: %.3g" % (name, err)      # through 14 BatchNorms: batch statistics amplify rounding


def test_empty_and_tiny_sparse_tensors(dev):
    """M = 0 rows and a single isolated voxel (the degenerate tensors clusters_voxelization can hand over)."""
    from gcanet_amd import sparseconv as S
    conv = S.SubMConv3d(64, 64, "k").to(dev)
    x = S.SparseConvTensor(torch.zeros(0, 64, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), [8] * 3, 1)
    assert conv(x).features.shape == (0, 64)
    f = torch.randn(1, 64, device=dev)
    y = conv(S.SparseConvTensor(f, torch.tensor([[0, 3, 4, 5]], dtype=torch.int32, device=dev), [8] * 3, 1)).features
    _close(y.detach().cpu(), (f @ conv.weight[13]).detach().cpu(), "isolated voxel = centre tap only")


@pytest.mark.parametrize("C,relu", [(64, True), (128, True), (64, False)])
def test_batch_norm_relu_matches_torch(dev, C, relu):
    """Fused BatchNorm1d(+ReLU) (csrc/sparseconv.hip bn_* kernels) vs nn.BatchNorm1d in training mode: output, input and
    parameter gradients, running statistics."""
    from gcanet_amd.sparseconv import batch_norm_relu
    g = torch.Generator().manual_seed(11)
    M = 3001
    x = torch.randn(M, C, generator=g) * 2 + 0.5
    gy = torch.randn(M, C, generator=g)
    bn_a, bn_b = torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1), torch.nn.BatchNorm1d(C, eps=1e-4, momentum=0.1)
    with torch.no_grad():
        bn_a.weight.copy_(torch.randn(C, generator=g)); bn_a.bias.copy_(torch.randn(C, generator=g))
    bn_b.load_state_dict(bn_a.state_dict())
    bn_a = bn_a.to(dev)
    xa = x.to(dev).requires_grad_(True)
    ya = batch_norm_relu(xa, bn_a, relu=relu)
    ya.backward(gy.to(dev))
    xb = x.clone().requires_grad_(True)
    yb = bn_b(xb)
    yb = torch.relu(yb) if relu else yb
    yb.backward(gy)
    _close(ya.detach().cpu(), yb.detach(), "output")
    _close(xa.grad.cpu(), xb.grad, "input gradient")
    _close(bn_a.weight.grad.cpu(), bn_b.weight.grad, "dgamma")
    _close(bn_a.bias.grad.cpu(), bn_b.bias.grad, "dbeta")
    _close(bn_a.running_mean.cpu(), bn_b.running_mean, "running mean")
    _close(bn_a.running_var.cpu(), bn_b.running_var, "running var")
    assert int(bn_a.num_batches_tracked) == 1


@pytest.mark.parametrize("M,Cin,Cout", [(70001, 64, 128), (3000, 128, 64)])
def test_gather_gemm_and_wgrad_on_random_rule_tables(dev, M, Cin, Cout):
    """The two kernels on a synthetic rule table (30 % filled, random sources) against a float64 gather + matmul on the
    same device: the large case runs unsplit (>= 512 workgroups), the small one with the offsets split 9 ways."""
    from gcanet_amd import _lib
    K = 27
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, Cin, device=dev, generator=g)
    W = torch.randn(K, Cin, Cout, device=dev, generator=g) / (K * Cin) ** 0.5
    dy = torch.randn(M, Cout, device=dev, generator=g)
    src = torch.randint(0, M, (M, K), device=dev, generator=g)
    rule = torch.where(torch.rand(M, K, device=dev, generator=g) < 0.3, src, torch.full_like(src, -1)).int().contiguous()
    out = torch.empty(M, Cout, device=dev)
    n_ws = _lib.lib().gcn_sparse_gather_gemm_ws_floats(M, K, Cout)
    assert (n_ws == 0) == (M > 60000)
    ws = torch.empty(max(n_ws, 1), device=dev)
    st = _lib.stream_of(x)
    _lib.call("gcn_sparse_gather_gemm", M, K, Cin, Cout, _lib.ptr(x), _lib.ptr(rule), _lib.ptr(W), 0, 0, _lib.ptr(out), _lib.ptr(ws), st)
    dW = torch.empty_like(W)
    _lib.call("gcn_sparse_wgrad", M, K, Cin, Cout, _lib.ptr(x), _lib.ptr(rule.t().contiguous()), _lib.ptr(dy), _lib.ptr(dW), st)
    ref = torch.zeros(M, Cout, dtype=torch.float64, device=dev)
    refw = torch.zeros(K, Cin, Cout, dtype=torch.float64, device=dev)
    for k in range(K):
        ok = rule[:, k] >= 0
        xs = x[rule[ok, k].long()].double()
        ref[ok] += xs @ W[k].double()
        refw[k] = xs.t() @ dy[ok].double()
    _close(out.cpu().double(), ref.cpu(), "gather-GEMM")
    _close(dW.cpu().double(), refw.cpu(), "weight gradient")
