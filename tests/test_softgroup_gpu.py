"""GPU parity: softgroup.ops drop-in (csrc/softgroup.hip) vs the oracle.  The reference holds no
tests or vectors for these ops ("parity unpinned"): the oracle is the C restatement of SG/src."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _ops():
    from gcanet_amd.softgroup import ops
    return ops


def _cloud(rng, sizes):
    xyz = np.concatenate([rng.random((s, 3)) for s in sizes]).astype(np.float32)
    bidx = np.concatenate([np.full(s, i) for i, s in enumerate(sizes)]).astype(np.int32)
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    return xyz, bidx, offs


@pytest.mark.parametrize("sizes,radius,mean_active", [((300, 200), 0.15, 50), ((500,), 0.3, 10), ((64, 1, 129), 0.5, 300)])
def test_ball_query_easy_csr_exact(dev, sizes, radius, mean_active):
    rng = np.random.default_rng(sum(sizes))
    xyz, bidx, offs = _cloud(rng, sizes)
    idx, sl = _ops().ball_query_easy(torch.from_numpy(xyz).to(dev), torch.from_numpy(bidx).to(dev),
                                     torch.from_numpy(offs).to(dev), radius, mean_active)
    io, slo = oracle.ballquery_batch_p(xyz, bidx, offs, radius, mean_active)
    assert idx.dtype == torch.int32 and sl.shape == (xyz.shape[0], 2)
    np.testing.assert_array_equal(sl.cpu().numpy(), slo)     # deterministic point-order CSR
    np.testing.assert_array_equal(idx.cpu().numpy(), io)


def test_ball_query_with_adjacency(dev):
    rng = np.random.default_rng(8)
    xyz, bidx, offs = _cloud(rng, (250, 150))
    n = xyz.shape[0]
    a1 = rng.random((n, n)).astype(np.float32)
    a2 = rng.random((n, n)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    idx, sl = _ops().ball_query(t(xyz), t(bidx), t(offs), t(a1), 0.3, t(a2), 0.2, 0.4, 20)
    io, slo = oracle.ballquery_batch_p(xyz, bidx, offs, 0.4, 20, a1, 0.3, a2, 0.2)
    np.testing.assert_array_equal(sl.cpu().numpy(), slo)
    np.testing.assert_array_equal(idx.cpu().numpy(), io)


def test_ball_query_per_point_cap(dev):
    # 1500 coincident points: every point has 1500 hits -> capped at 1000 (bfs_cluster_easy.cu:43)
    xyz = np.zeros((1500, 3), np.float32)
    bidx = np.zeros(1500, np.int32)
    offs = np.array([0, 1500], np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    idx, sl = _ops().ball_query_easy(t(xyz), t(bidx), t(offs), 0.1, 100)
    io, slo = oracle.ballquery_batch_p(xyz, bidx, offs, 0.1, 100)
    assert (slo[:, 1] == 1000).all()
    np.testing.assert_array_equal(sl.cpu().numpy(), slo)
    np.testing.assert_array_equal(idx.cpu().numpy(), io)


@pytest.mark.parametrize("mode", [3, 4])
def test_voxelization_fwd_bwd(dev, mode):
    ops = _ops()
    rng = np.random.default_rng(mode)
    N, C = 3000, 67
    coords = np.concatenate([rng.integers(0, 2, (N, 1)), rng.integers(0, 9, (N, 3))], 1).astype(np.int64)
    feats = rng.standard_normal((N, C)).astype(np.float32)
    oc, im, om = ops.voxelization_idx(torch.from_numpy(coords), 2, mode)
    f = torch.from_numpy(feats).to(dev).requires_grad_()
    out = ops.voxelization(f, om.to(dev), mode)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.voxelization(feats, om.numpy(), mode))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), oracle.voxelization_bp(go, om.numpy(), N, mode))
    # round trip: scatter the voxel means back to points (property used by clusters_voxelization)
    np.testing.assert_allclose(out.detach().cpu().numpy()[im.numpy().astype(np.int64)].mean(), feats.mean(), atol=0.05)


def _segments(rng, P, maxlen):
    lens = rng.integers(1, maxlen, P)
    return np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)


@pytest.mark.parametrize("C", [3, 64, 130])
def test_segment_ops(dev, C):
    ops = _ops()
    rng = np.random.default_rng(C)
    offs = _segments(rng, 37, 200)
    S = int(offs[-1])
    inp = rng.standard_normal((S, C)).astype(np.float32)
    ti, to = torch.from_numpy(inp).to(dev), torch.from_numpy(offs).to(dev)
    np.testing.assert_array_equal(ops.sec_mean(ti, to).cpu().numpy(), oracle.sec_op("mean", inp, offs))
    np.testing.assert_array_equal(ops.sec_min(ti, to).cpu().numpy(), oracle.sec_op("min", inp, offs))
    np.testing.assert_array_equal(ops.sec_max(ti, to).cpu().numpy(), oracle.sec_op("max", inp, offs))
    f = ti.clone().requires_grad_()
    out = ops.global_avg_pool(f, to)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), oracle.global_avg_pool(inp, offs))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    np.testing.assert_array_equal(f.grad.cpu().numpy(), oracle.global_avg_pool_bp(go, offs, S))


def test_mask_iou_and_label(dev):
    ops = _ops()
    rng = np.random.default_rng(21)
    N, nI, P = 5000, 23, 40
    labels = rng.integers(0, nI, N).astype(np.int64)
    labels[rng.random(N) < 0.1] = -100
    pointnum = np.bincount(labels[labels >= 0], minlength=nI).astype(np.int32)
    cls = rng.integers(0, 7, nI).astype(np.int64)
    cls[3] = -100
    offs = _segments(rng, P, 400)
    pidx = rng.integers(0, N, int(offs[-1])).astype(np.int32)
    mask = rng.random(int(offs[-1])).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    iou_c = ops.get_mask_iou_on_cluster(t(pidx), t(offs), t(labels), t(pointnum))
    np.testing.assert_array_equal(iou_c.cpu().numpy(), oracle.get_mask_iou(pidx, offs, labels, pointnum))
    iou_p = ops.get_mask_iou_on_pred(t(pidx), t(offs), t(labels), t(pointnum), t(mask))
    np.testing.assert_array_equal(iou_p.cpu().numpy(), oracle.get_mask_iou(pidx, offs, labels, pointnum, mask))
    for thr in (0.0, 0.05, 0.5):
        ml = ops.get_mask_label(t(pidx), t(offs), t(labels), t(cls), t(pointnum), iou_c, thr)
        np.testing.assert_array_equal(ml.cpu().numpy(),
                                      oracle.get_mask_label(pidx, offs, labels, cls, pointnum, iou_c.cpu().numpy(), thr))


def test_full_size_softgroup_path_properties(dev):
    """cfg4 shape (N=100 000): voxelize + ball query + aggregate, checked through properties."""
    ops = _ops()
    g = torch.Generator().manual_seed(1234)
    N = 100000
    xyz = torch.rand(N, 3, generator=g)
    coords = torch.cat([torch.zeros(N, 1, dtype=torch.int64), (xyz * 128).floor().long()], 1)
    oc, im, om = ops.voxelization_idx(coords, 1, 4)
    assert (om[:, 0].sum() == N) and im.max() == oc.shape[0] - 1
    feats = torch.rand(N, 16, generator=g).to(dev)
    vox = ops.voxelization(feats, om.to(dev), 4)
    # mean pooling conserves the count-weighted sum
    np.testing.assert_allclose((vox * om[:, :1].to(dev)).sum(0).cpu().numpy(), feats.sum(0).cpu().numpy(), rtol=1e-4)
    bidx = torch.zeros(N, dtype=torch.int32, device=dev)
    offs = torch.tensor([0, N], dtype=torch.int32, device=dev)
    idx, sl = ops.ball_query_easy(xyz.to(dev), bidx, offs, 0.03, 32)
    sl_c = sl.cpu().numpy()
    assert (np.diff(sl_c[:, 0]) == sl_c[:-1, 1]).all() and sl_c[-1].sum() == idx.numel()
    # symmetry of the neighbour relation on a sample, self always included
    ic = idx.cpu().numpy()
    for p in range(0, N, 9973):
        nb = ic[sl_c[p, 0]:sl_c[p, 0] + sl_c[p, 1]]
        assert p in nb and (np.diff(nb) > 0).all()
        q = int(nb[-1])
        assert p in ic[sl_c[q, 0]:sl_c[q, 0] + sl_c[q, 1]]


@pytest.mark.parametrize("sizes,radius,mean_active", [((3000, 2500), 0.05, 40), ((4096,), 0.11, 20), ((2048, 1, 700), 0.02, 8),
                                                      ((5000,), 0.4, 64)])
def test_ball_query_grid_path_matches_oracle(dev, sizes, radius, mean_active):
    """n >= 2048 takes the uniform-grid candidate search: same CSR, bit for bit, incl. the truncation at
    n*meanActive (last case: ~1100 neighbours per point, over the 1000 cap and over the 1024-candidate grid limit)."""
    rng = np.random.default_rng(sum(sizes) + 1)
    xyz, bidx, offs = _cloud(rng, sizes)
    xyz[: sizes[0] // 2] *= np.float32(0.5)                 # uneven density
    idx, sl = _ops().ball_query_easy(torch.from_numpy(xyz).to(dev), torch.from_numpy(bidx).to(dev),
                                     torch.from_numpy(offs).to(dev), radius, mean_active)
    io, slo = oracle.ballquery_batch_p(xyz, bidx, offs, radius, mean_active)
    np.testing.assert_array_equal(sl.cpu().numpy(), slo)
    np.testing.assert_array_equal(idx.cpu().numpy(), io)


def test_ball_query_grid_equals_bruteforce_full_size(dev):
    """cfg4 (N = 100 000, radius 0.03): grid path vs the brute-force kernel through the C ABI (grid_ws = NULL)."""
    import ctypes as C
    from gcanet_amd import _lib
    g = torch.Generator().manual_seed(7)
    N, mean_active = 100000, 32
    xyz = torch.rand(N, 3, generator=g).to(dev)
    bidx = torch.cat([torch.zeros(60000), torch.ones(40000)]).to(torch.int32).to(dev)
    offs = torch.tensor([0, 60000, N], dtype=torch.int32, device=dev)
    res = []
    for use_grid in (True, False):
        idx = torch.zeros(N * mean_active, dtype=torch.int32, device=dev)
        sl = torch.zeros(N, 2, dtype=torch.int32, device=dev)
        cw = torch.empty(N + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(_lib.lib().gcn_ballquery_grid_ws_bytes(N), dtype=torch.uint8, device=dev) if use_grid else None
        total = C.c_int(0)
        _lib.call("gcn_ballquery_batch_p", N, mean_active, 0.03, _lib.ptr(xyz), _lib.ptr(bidx), _lib.ptr(offs), None, 0.0,
                  None, 0.0, _lib.ptr(idx), _lib.ptr(sl), _lib.ptr(cw), 2, _lib.ptr(ws), C.addressof(total),
                  _lib.stream_of(xyz))
        res.append((idx[: total.value].clone(), sl.clone(), total.value))
    assert res[0][2] == res[1][2] and res[0][2] > N
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][0], res[1][0])


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("ncol,N,span", [(4, 5000, 12), (3, 777, 4), (4, 1, 3), (4, 4096, 60000)])
def test_voxelization_idx_device_matches_oracle(dev, mode, ncol, N, span):
    """Device voxelize_idx (csrc/voxelize_dev.hip, sort based) vs the oracle's insertion-ordered hash
    (voxelize.cpp:11-165): identical voxel numbering, rule rows and coordinates for every mode."""
    rng = np.random.default_rng(N + span + mode)
    coords = rng.integers(0, span, size=(N, ncol)).astype(np.int64)
    if ncol == 4:
        coords[:, 0] = rng.integers(0, 3, size=N)
    oc, im, om = _ops().voxelization_idx(torch.from_numpy(coords).to(dev), 3, mode)
    assert oc.is_cuda and im.is_cuda and om.is_cuda
    oco, imo, omo = oracle.voxelization_idx(coords, 3, mode)
    np.testing.assert_array_equal(im.cpu().numpy(), imo)
    np.testing.assert_array_equal(om.cpu().numpy(), omo)
    np.testing.assert_array_equal(oc.cpu().numpy(), oco)


def test_voxelization_idx_device_full_size_equals_host(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(99)
    N = 100000
    coords = torch.cat([torch.randint(0, 2, (N, 1), generator=g), (torch.rand(N, 3, generator=g) * 128).floor().long()], 1)
    h = ops.voxelization_idx(coords, 2, 4)
    d = ops.voxelization_idx(coords.to(dev), 2, 4)
    for a, b in zip(h, d):
        assert torch.equal(a, b.cpu())


def test_voxelization_idx_device_rejects_out_of_range(dev):
    coords = torch.tensor([[0, 1, 2, 70000]], dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):
        _ops().voxelization_idx(coords, 1, 4)


@pytest.mark.parametrize("class_id,threshold", [(0, 5.0), (4, 0.01)])
def test_bfs_cluster_device_matches_oracle(dev, class_id, threshold):
    """bfs_cluster (K16) on CUDA neighbour lists: device components + the reference's BFS member order
    (csrc/cluster_dev.hip) against the CPU restatement of bfs_cluster.cpp:48-143."""
    ops = _ops()
    rng = np.random.default_rng(31)
    centers = rng.random((12, 3)).astype(np.float32)
    which = rng.integers(0, 12, 4000)
    t_ = rng.random((4000, 1)).astype(np.float32)
    dirs = rng.standard_normal((12, 3)).astype(np.float32)
    xyz = (centers[which] + 0.2 * t_ * dirs[which] / np.linalg.norm(dirs[which], axis=1, keepdims=True)
           + 0.003 * rng.standard_normal((4000, 3))).astype(np.float32)          # thin rods: many BFS levels
    bidx = np.zeros(4000, np.int32)
    offs = np.array([0, 4000], np.int32)
    t = lambda a: torch.from_numpy(a).to(dev)
    idx, sl = ops.ball_query_easy(t(xyz), t(bidx), t(offs), 0.02, 50)
    mean = np.array([-1., -1., 3917., 12056., 2303., 8331., 3948., 3166., 5629., 11719.], np.float32)
    ci, co = ops.bfs_cluster(torch.from_numpy(mean), idx, sl, threshold, class_id)
    rci, rco = oracle.bfs_cluster(mean, idx.cpu().numpy(), sl.cpu().numpy(), threshold, class_id)
    assert rco.size > 3 and not ci.is_cuda
    np.testing.assert_array_equal(co.numpy(), rco)
    np.testing.assert_array_equal(ci.numpy(), rci)


def test_octree_ball_query_matches_oracle_order(dev):
    """octree_ball_query (functions.py:127-157): neighbour lists in the reference's order -- active leaves of the fixed
    3-level octree breadth first, ascending index inside a leaf -- vs the CPU restatement of octree_ball_query.cpp:19-165
    + .cu:56-126 (oracle/gcanet_oracle.c).  Parity unpinned by reference data (the reference holds no vectors)."""
    import oracle
    from gcanet_amd.softgroup.ops import functions as SGF
    rng = np.random.default_rng(7)
    for n, radius in ((3000, 0.06), (500, 0.2), (4096, 0.03)):
        c = rng.random((n, 3)).astype(np.float32)
        c[: n // 4] *= 0.25                                           # a dense corner: long lists, several leaves each
        idx_o, sl_o, leaf_o = oracle.octree_ball_query(c, 40, radius)
        idx_d, sl_d = SGF.octree_ball_query(torch.from_numpy(c), 40, radius)
        np.testing.assert_array_equal(SGF._octree_leaf(torch.from_numpy(c).to(dev)).cpu().numpy(), leaf_o)
        np.testing.assert_array_equal(sl_d.cpu().numpy(), sl_o)
        np.testing.assert_array_equal(idx_d.cpu().numpy(), idx_o)
        assert int(sl_o[:, 1].max()) < 1000                           # below the cap, where the two truncation rules agree
