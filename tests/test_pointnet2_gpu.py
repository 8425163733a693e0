"""GPU parity: pointnet2_ops drop-in (csrc/pointnet2.hip) vs the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _u():
    from gcanet_amd.pointnet2_ops import pointnet2_utils as u
    return u


@pytest.mark.parametrize("b,n,m,r,ns", [(2, 500, 100, 0.2, 16), (1, 64, 64, 0.05, 8), (3, 1000, 7, 0.5, 64),
                                         (2, 130, 50, 1e-6, 4), (1, 10, 10, 10.0, 32)])
def test_ball_query(dev, b, n, m, r, ns):
    rng = np.random.default_rng(n + m)
    xyz = rng.random((b, n, 3)).astype(np.float32)
    new = xyz[:, :m].copy() if m <= n else rng.random((b, m, 3)).astype(np.float32)
    if r < 1e-3:
        new += 1.0  # nothing in range -> all-zero rows (ball_query.cpp:19-21)
    out = _u().ball_query(r, ns, torch.from_numpy(xyz).to(dev), torch.from_numpy(new).to(dev))
    assert out.dtype == torch.int32 and out.shape == (b, m, ns)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.ball_query(r, ns, xyz, new))


@pytest.mark.parametrize("b,c,n,np_,ns", [(2, 5, 100, 40, 3), (1, 64, 2048, 2048, 16), (2, 3, 33, 7, 5), (1, 130, 50, 50, 64)])
def test_group_points_fwd_bwd(dev, b, c, n, np_, ns):
    rng = np.random.default_rng(c + n)
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, np_, ns)).astype(np.int32)
    f = torch.from_numpy(pts).to(dev).requires_grad_()
    out = _u().grouping_operation(f, torch.from_numpy(idx).to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.group_points(pts, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.group_points_grad(go, idx, n), rtol=1e-4, atol=1e-4)


def test_group_points_grad_large_rows_use_atomic_path(dev):
    rng = np.random.default_rng(0)
    b, c, n, np_, ns = 1, 2, 20000, 300, 4
    idx = rng.integers(0, n, (b, np_, ns)).astype(np.int32)
    go = rng.standard_normal((b, c, np_, ns)).astype(np.float32)
    f = torch.zeros(b, c, n, device=dev, requires_grad=True)
    _u().grouping_operation(f, torch.from_numpy(idx).to(dev)).backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.group_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)


def test_gather_fwd_bwd(dev):
    rng = np.random.default_rng(4)
    b, c, n, m = 2, 9, 300, 77
    pts = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    f = torch.from_numpy(pts).to(dev).requires_grad_()
    out = _u().gather_operation(f, torch.from_numpy(idx).to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.gather_points(pts, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.gather_points_grad(go, idx, n), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("b,n,m", [(2, 1000, 64), (1, 2048, 512), (2, 33, 33), (1, 5000, 20), (1, 700, 1)])
def test_fps(dev, b, n, m):
    rng = np.random.default_rng(n)
    xyz = rng.random((b, n, 3)).astype(np.float32)
    xyz[:, 5] = 0.0           # exercises the |p|^2 <= 1e-3 skip (sampling_gpu.cu:100-101)
    out = _u().furthest_point_sample(torch.from_numpy(xyz).to(dev), m)
    assert out.dtype == torch.int32
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.furthest_point_sampling(xyz, m))


def test_fps_ties_follow_reference_tree_order(dev):
    # lattice -> many exact ties; winner must match the reference's block reduction order
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
    xyz = (g[None].astype(np.float32) + 1.0) / 8.0
    out = _u().furthest_point_sample(torch.from_numpy(xyz).to(dev), 40)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.furthest_point_sampling(xyz, 40))


def test_three_nn_and_interpolate(dev):
    rng = np.random.default_rng(9)
    b, n, m, c = 2, 400, 90, 6
    unk = rng.random((b, n, 3)).astype(np.float32)
    kn = rng.random((b, m, 3)).astype(np.float32)
    dist, idx = _u().three_nn(torch.from_numpy(unk).to(dev), torch.from_numpy(kn).to(dev))
    d2, io = oracle.three_nn(unk, kn)
    np.testing.assert_array_equal(idx.cpu().numpy(), io)
    np.testing.assert_array_equal(dist.cpu().numpy(), np.sqrt(d2))
    w = rng.random((b, n, 3)).astype(np.float32)
    feats = rng.standard_normal((b, c, m)).astype(np.float32)
    f = torch.from_numpy(feats).to(dev).requires_grad_()
    out = _u().three_interpolate(f, idx, torch.from_numpy(w).to(dev))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.three_interpolate(feats, io, w))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_allclose(f.grad.cpu().numpy(), oracle.three_interpolate_grad(go, io, w, m), rtol=1e-4, atol=1e-5)


def test_three_nn_fewer_than_three_known(dev):
    unk = np.random.default_rng(1).random((1, 5, 3)).astype(np.float32)
    kn = np.random.default_rng(2).random((1, 2, 3)).astype(np.float32)
    dist, idx = _u().three_nn(torch.from_numpy(unk).to(dev), torch.from_numpy(kn).to(dev))
    d2, io = oracle.three_nn(unk, kn)
    np.testing.assert_array_equal(idx.cpu().numpy(), io)
    assert np.isinf(dist.cpu().numpy()[..., 2]).all() and np.isinf(d2[..., 2]).all()


def test_query_and_group_module(dev):
    rng = np.random.default_rng(11)
    xyz = rng.random((2, 200, 3)).astype(np.float32)
    feats = rng.standard_normal((2, 4, 200)).astype(np.float32)
    qg = _u().QueryAndGroup(0.3, 8)
    out = qg(torch.from_numpy(xyz).to(dev), torch.from_numpy(xyz[:, :50].copy()).to(dev), torch.from_numpy(feats).to(dev))
    idx = oracle.ball_query(0.3, 8, xyz, xyz[:, :50].copy())
    gx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - xyz[:, :50].transpose(0, 2, 1)[..., None]
    exp = np.concatenate([gx, oracle.group_points(feats, idx)], 1)
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


def test_dtype_and_contiguity_errors(dev):
    u = _u()
    f = torch.rand(1, 4, 10, device=dev)
    idx = torch.zeros(1, 3, 2, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError, match="int"):
        u.grouping_operation(f, idx)
    with pytest.raises(RuntimeError, match="contiguous"):
        u.grouping_operation(f.transpose(1, 2), idx.int())
    with pytest.raises(RuntimeError):
        u.grouping_operation(f.cpu(), idx.int().cpu())
