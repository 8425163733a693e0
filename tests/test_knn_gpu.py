"""GPU parity: fused HIP kNN (csrc/knn.hip, via the C ABI) vs the oracle -- indices BIT-EXACT."""
import numpy as np
import pytest
import torch

import oracle
from util import knn_rows_equivalent, pn_metric64, sqdist64

pytestmark = pytest.mark.gpu


def _knn_mod(k, t):
    from gcanet_amd.knn_cuda import KNN
    return KNN(k, transpose_mode=t)


# shapes of the reference's own test (KNN_CUDA/tests/test_knn_cuda.py:59-87) + edge cases
CASES = [
    # (dim, nr, nq, k)
    (5, 1000, 50, 2), (5, 1000, 50, 10), (5, 1000, 50, 400), (5, 10, 50, 2), (5, 30, 50, 10),
    (5, 30001, 50, 10), (3, 777, 333, 64), (3, 64, 64, 64), (3, 65, 1, 65), (7, 129, 130, 128),
    (3, 600, 40, 257), (64, 300, 100, 16), (1, 100, 33, 5),
]


@pytest.mark.parametrize("dim,nr,nq,k", CASES)
def test_knn_cuda_matches_oracle_exactly(dev, dim, nr, nq, k):
    rng = np.random.default_rng(dim * 1000 + nr + k)
    ref = rng.random((2, dim, nr)).astype(np.float32)
    qry = rng.random((2, dim, nq)).astype(np.float32)
    D, I = _knn_mod(k, False)(torch.from_numpy(ref).to(dev), torch.from_numpy(qry).to(dev))
    assert D.shape == (2, k, nq) and I.shape == (2, k, nq) and I.dtype == torch.int64 and D.dtype == torch.float32
    Do, Io = oracle.KNN_forward(ref, qry, k, False)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)          # same fmaf chain + sqrt -> identical bits


def test_knn_cuda_transpose_mode_and_ties(dev):
    # integer grid -> massive exact ties: lowest index must win (knn.cu:125-131)
    rng = np.random.default_rng(1)
    ref = rng.integers(0, 4, (2, 500, 3)).astype(np.float32)
    qry = rng.integers(0, 4, (2, 70, 3)).astype(np.float32)
    D, I = _knn_mod(20, True)(torch.from_numpy(ref).to(dev), torch.from_numpy(qry).to(dev))
    assert D.shape == (2, 70, 20)
    Do, Io = oracle.KNN_forward(ref, qry, 20, True)
    np.testing.assert_array_equal(I.cpu().numpy(), Io)
    np.testing.assert_array_equal(D.cpu().numpy(), Do)
    # stable order inside each list
    Ic = I.cpu().numpy()
    Dc = D.cpu().numpy()
    same = Dc[..., 1:] == Dc[..., :-1]
    assert (Ic[..., 1:][same] > Ic[..., :-1][same]).all()


def test_knn_cuda_kdtree_property(dev):
    """reference test property: distances == sklearn KDTree to 3 decimals (test_knn_cuda.py:32-47)."""
    from sklearn.neighbors import KDTree
    rng = np.random.default_rng(2)
    ref = rng.random((1, 1000, 5)).astype(np.float32)
    qry = rng.random((1, 50, 5)).astype(np.float32)
    D, I = _knn_mod(10, True)(torch.from_numpy(ref).to(dev), torch.from_numpy(qry).to(dev))
    dk, ik = KDTree(ref[0]).query(qry[0], k=10)
    np.testing.assert_almost_equal(D[0].cpu().numpy(), dk, decimal=3)
    np.testing.assert_array_equal(I[0].cpu().numpy(), ik)


def test_knn_cuda_errors(dev):
    from gcanet_amd.knn_cuda import KNN
    x = torch.rand(1, 3, 10, device=dev)
    with pytest.raises(RuntimeError, match="k"):
        KNN(11)(x, x)
    with pytest.raises(RuntimeError):
        KNN(2)(x.cpu(), x.cpu())


# ------------------------------------------------------------------ in-model kNN
def _knn_model(x, k1, k2, metric):
    from gcanet_amd import dgcnn
    fn = dgcnn.knn if metric == 0 else dgcnn.knn_points_normals
    return fn(x, k1, k2)


@pytest.mark.parametrize("C,N,k1,k2,metric", [
    (3, 2048, 16, 16, 0), (3, 1000, 4, 16, 0), (64, 700, 20, 20, 0), (128, 513, 64, 64, 0), (64, 2048, 64, 64, 0),
    (32, 256, 8, 16, 0), (128, 1028, 33, 33, 0), (64, 64, 64, 64, 0),
    (6, 2048, 16, 16, 1), (6, 1024, 16, 64, 1), (6, 3072, 64, 64, 1), (6, 999, 80, 80, 1), (3, 300, 100, 200, 0), (9, 64, 64, 64, 0)])
def test_knn_model_matches_oracle_exactly(dev, C, N, k1, k2, metric):
    g = torch.Generator().manual_seed(1234 + C + N)
    x = torch.rand(2, C, N, generator=g)
    if metric == 1:
        x[:, 3:6] = torch.nn.functional.normalize(torch.randn(2, 3, N, generator=g), dim=1)
    idx = _knn_model(x.to(dev), k1, k2, metric)
    ref = oracle.knn_model(x.numpy(), k1, k2, metric)
    assert idx.dtype == torch.int64 and tuple(idx.shape) == ref.shape
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)


def test_knn_model_vs_reference_golden(dev, golden):
    """HIP kernel vs indices produced by the reference's own knn()/knn_points_normals() (tie-aware)."""
    for key, metric in (("knn_grid", 0), ("knn_rand", 0), ("knnpn_grid", 1), ("knnpn_rand", 1)):
        x = golden[key + "_x"]
        ref = golden[key + "_idx_k16"]
        idx = _knn_model(torch.from_numpy(x).to(dev), 16, 16, metric).cpu().numpy()
        for b in range(x.shape[0]):
            fn = sqdist64(x[b]) if metric == 0 else pn_metric64(x[b])
            tol = dict(rtol=0, atol=0) if "grid" in key else dict(rtol=1e-6, atol=1e-9)
            ident, tie, bad = knn_rows_equivalent(idx[b], ref[b], fn, **tol)
            assert bad == 0 and ident >= 0.98 * idx.shape[1], (key, ident, tie, bad)


def test_knn_full_size_properties(dev):
    """BASELINE size (N=8192, k=64): size-independent properties instead of an oracle run."""
    g = torch.Generator().manual_seed(1234)
    x = torch.rand(2, 3, 8192, generator=g).to(dev)
    from gcanet_amd import dgcnn
    idx = dgcnn.knn(x, 64, 64)
    assert idx.shape == (2, 8192, 64)
    # (1) self is the nearest neighbour; (2) rows have no duplicates; (3) distances ascend;
    # (4) the 64th distance is a valid threshold: exactly >= 64 points within it (checked on a sample)
    assert (idx[:, :, 0] == torch.arange(8192, device=dev)).all()
    s = torch.sort(idx, dim=-1)[0]
    assert (s[..., 1:] != s[..., :-1]).all()
    xt = x.transpose(1, 2)
    d = ((xt.unsqueeze(2) - torch.gather(xt.unsqueeze(1).expand(-1, 8192, -1, -1), 2,
                                         idx.unsqueeze(-1).expand(-1, -1, -1, 3))) ** 2).sum(-1)
    assert (d[..., 1:] >= d[..., :-1] - 1e-6).all()
    rows = torch.arange(0, 8192, 97, device=dev)
    full = ((xt[0, rows].unsqueeze(1) - xt[0].unsqueeze(0)) ** 2).sum(-1)
    kth = torch.sort(full, dim=1)[0][:, 63]
    np.testing.assert_allclose(d[0, rows, 63].cpu().numpy(), kth.cpu().numpy(), rtol=1e-5, atol=1e-7)
    # KNN_CUDA signature on the same cloud agrees with the in-model kNN as a set
    from gcanet_amd.knn_cuda import KNN
    _, I = KNN(64, transpose_mode=False)(x, x)
    a = torch.sort(I.permute(0, 2, 1), dim=-1)[0]
    assert (a == s).float().mean() > 0.999


def test_search_knn_known_answer_on_gpu(dev):
    """The reference's only golden table (models/search_knn.py:183-243) through the drop-in SoftProjection."""
    import json
    import os
    from gcanet_amd.search_knn import SoftProjection
    ka = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "search_knn_known_answer.json")))
    t = lambda a: torch.tensor(a, dtype=torch.float32).t().unsqueeze(0).contiguous().to(dev)
    pc, qc, pf = t(ka["point_cloud"]), t(ka["query_cloud"]), t(ka["point_features"])
    sp = SoftProjection(3, initial_temperature=1.0).to(dev)
    prop = sp.propagate(pc, pf, qc)[0].t().detach().cpu().numpy()
    np.testing.assert_allclose(prop, np.asarray(ka["expected_features_nn_3"]), atol=1.5e-3)
    sp._temperature.data = torch.tensor(0.1, device=dev)      # sigma := 0.1**2 (search_knn.py:279)
    proj = sp.project(qc, pc)[0].t().detach().cpu().numpy()
    np.testing.assert_allclose(proj, np.asarray(ka["expected_nn_cloud"]), atol=1.5e-3)


@pytest.mark.parametrize("shape,k", [((3, 500, 120), 30), ((1000, 128), 64), ((7, 5), 3), ((2, 33, 64), 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_topk_rows_matches_torch(dev, shape, k, dtype):
    """csrc/knn.hip:topk_rows_kernel vs torch.topk: identical sorted values; indices identical where values are
    distinct, and always consistent (x[idx] == values, no duplicates); ties resolve to the lower column."""
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(sum(shape) + k)
    x = torch.randn(*shape, generator=g).to(dev).to(dtype)          # bf16: plenty of exact ties
    x.requires_grad_(True)
    v, i = dgcnn.topk_rows(x, k)
    tv, ti = torch.topk(x.detach(), k, dim=-1, largest=True)
    assert v.dtype == x.dtype and i.dtype == torch.int64
    assert torch.equal(v.detach(), tv)
    assert torch.equal(torch.gather(x.detach(), -1, i), tv)
    si = i.sort(-1)[0]
    assert (si[..., 1:] != si[..., :-1]).all()
    if dtype == torch.float32:
        assert torch.equal(i, ti)
    # ties: among equal values the lower column comes first
    same = v.detach()[..., 1:] == v.detach()[..., :-1]
    assert (i[..., 1:][same] > i[..., :-1][same]).all()
    go = torch.randn(*v.shape, generator=g).to(dev).to(dtype)
    (gx,) = torch.autograd.grad(v, x, go)
    ref = torch.zeros_like(x).scatter_(-1, i, go)
    assert torch.equal(gx, ref)


def _knn_both(x, k, metric):
    """gcn_knn_model through the pruned (tile_ws) and the brute-force (tile_ws = NULL) paths: (idx, val) each."""
    from gcanet_amd import _lib
    B, C, N = x.shape
    outs = []
    for tiled in (True, False):
        idx = torch.empty(B, N, k, dtype=torch.int64, device=x.device)
        val = torch.empty(B, N, k, dtype=torch.float32, device=x.device)
        xx = torch.empty(B, N, dtype=torch.float32, device=x.device)
        ws = torch.empty(_lib.lib().gcn_knn_tiles_ws_bytes(B, C, N), dtype=torch.uint8, device=x.device) if tiled else None
        _lib.call("gcn_knn_model", _lib.ptr(x), B, C, N, k, k, metric, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(xx), _lib.ptr(ws),
                  _lib.stream_of(x))
        outs.append((idx, val))
    return outs


@pytest.mark.parametrize("kind", ["uniform", "clustered", "grid_ties", "plane", "duplicates"])
@pytest.mark.parametrize("B,N,k,metric", [(2, 2048, 64, 0), (3, 1000, 20, 0), (2, 4096, 64, 1), (2, 2000, 64, 1), (1, 777, 33, 1)])
def test_knn_tiles_bitexact_vs_bruteforce(dev, kind, B, N, k, metric):
    """Morton-tiled pruned kNN (and, for the normal metric at N % 1024 == 0, the threshold + filter + re-rank path of
    knn_normal.hip, which the same workspace switches on) == brute-force kernel, indices AND distances, on benign and
    adversarial clouds:
    clusters (uneven density), an integer grid (masses of exact ties), a plane (degenerate boxes), duplicated points."""
    g = torch.Generator().manual_seed(N * 7 + k + metric)
    if kind == "uniform":
        p = torch.rand(B, N, 3, generator=g)
    elif kind == "clustered":
        c = torch.rand(B, 8, 3, generator=g)
        p = c[:, torch.randint(0, 8, (N,), generator=g)] + 0.02 * torch.randn(B, N, 3, generator=g)
        p[:, : N // 10] = torch.rand(B, N // 10, 3, generator=g) * 5.0
    elif kind == "grid_ties":
        p = torch.randint(0, 12, (B, N, 3), generator=g).float() / 4.0
    elif kind == "plane":
        p = torch.rand(B, N, 3, generator=g)
        p[..., 2] = 0.25
    else:
        p = torch.rand(B, N // 4 + 1, 3, generator=g).repeat(1, 4, 1)[:, :N]
    if metric == 1:
        nrm = torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=-1)
        if kind == "grid_ties":
            nrm = torch.nn.functional.normalize(torch.randint(-1, 2, (B, N, 3), generator=g).float() + 0.001, dim=-1)
        p = torch.cat([p, nrm], -1)
    x = p.transpose(1, 2).contiguous().to(dev)
    (i1, v1), (i0, v0) = _knn_both(x, k, metric)
    assert torch.equal(i1, i0)
    assert torch.equal(v1, v0)


def test_knn_tiles_unnormalised_normals_fall_back_to_full_scan(dev):
    """|n| > 1.22 makes the normal factor's lower bound non-positive: pruning must switch itself off, results unchanged."""
    g = torch.Generator().manual_seed(3)
    p = torch.cat([torch.rand(1, 1024, 3, generator=g), 3.0 * torch.randn(1, 1024, 3, generator=g)], -1)
    x = p.transpose(1, 2).contiguous().to(dev)
    (i1, v1), (i0, v0) = _knn_both(x, 32, 1)
    assert torch.equal(i1, i0) and torch.equal(v1, v0)


@pytest.mark.parametrize("transpose_mode", [False, True])
@pytest.mark.parametrize("kind", ["uniform", "grid_ties"])
def test_knn_cuda_self_query_tiles_equals_bruteforce(dev, transpose_mode, kind):
    """KNN_CUDA module on a cloud against itself takes the pruned kernel; a clone of the query (different pointer)
    takes the brute-force one: identical D and I in both layouts."""
    from gcanet_amd.knn_cuda import KNN
    g = torch.Generator().manual_seed(21)
    B, N, k = 2, 3000, 40
    p = torch.rand(B, N, 3, generator=g) if kind == "uniform" else torch.randint(0, 10, (B, N, 3), generator=g).float() / 3.0
    x = (p if transpose_mode else p.transpose(1, 2)).contiguous().to(dev)
    m = KNN(k, transpose_mode=transpose_mode)
    d1, i1 = m(x, x)
    d0, i0 = m(x, x.clone())
    assert torch.equal(i1, i0) and torch.equal(d1, d0)


@pytest.mark.parametrize("B,C,N,k1,k2", [(2, 64, 600, 80, 80), (1, 32, 260, 20, 100), (1, 128, 512, 128, 128), (2, 64, 1000, 65, 65)])
def test_feature_knn_k_up_to_128_on_matrix_cores(dev, B, C, N, k1, k2):
    """64 < k <= 128 (the reference's default k = 80, M4:544-550) also runs on the f32 MFMA kernel (two list registers
    per query): bit-exact indices vs the oracle, incl. the dilated pick and an integer grid full of ties."""
    from gcanet_amd import dgcnn
    g = torch.Generator().manual_seed(C + N + k2)
    for kind in ("gauss", "ties"):
        x = torch.randn(B, C, N, generator=g) if kind == "gauss" else torch.randint(0, 3, (B, C, N), generator=g).float()
        idx = dgcnn.knn(x.to(dev), k1, k2).cpu().numpy()
        ref = oracle.knn_model(x.numpy(), k1, k2, metric=0)
        np.testing.assert_array_equal(idx, ref)


# ------------------------------------------------------------------ feature-space kNN: bf16 prefilter + exact re-rank
def _feature_cloud(kind, B, C, N, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, N, generator=g)
    if kind == "relu":                   # post-activation-like: mostly positive, a common offset
        x = torch.nn.functional.leaky_relu(x + 0.5, 0.2)
    elif kind == "offset":               # tiny spread around a large mean: the reference's own f32 noise decides
        x = x * 0.05 + 4.0               # the order -> the a-posteriori check fails and the exhaustive path runs
    elif kind == "offset2":              # the same, harsher: the reference's f32 error bound exceeds every neighbour distance
        x = x * 0.01 + 8.0
    elif kind == "dups":                 # exact duplicates: exact key ties straddling the k-th place
        x[:, :, N // 2:] = x[:, :, :N // 2]
    elif kind == "clusters":             # tight clusters of ~40 points: the k-th neighbour is far beyond the first few
        centres = torch.randn(B, C, N // 40 + 1, generator=g) * 3
        x = centres[:, :, torch.arange(N) // 40] + 0.01 * x
    elif kind == "blobs":                # 64 blobs stored in ROTATION (point i belongs to blob i % 64, as bench.blob_clouds
        centres = torch.randn(B, C, 64, generator=g)      # lays its clouds out): a fixed-phase strided sample meets the
        x = centres[:, :, torch.arange(N) % 64] + 0.03 * x   # same 8 blobs for every query (round 2: every query flagged)
    elif kind == "flat":                 # 64 patches of near-identical features (a CAD part's flat faces), same layout
        x = torch.randn(B, C, 64, generator=g)[:, :, torch.arange(N) % 64] + 1e-4 * x
    return x


@pytest.mark.parametrize("kind,C,N,k1,k2", [
    ("normal", 64, 1024, 16, 16), ("normal", 32, 2048, 20, 20), ("relu", 64, 4096, 33, 33), ("normal", 128, 2048, 64, 64),
    ("normal", 64, 2048, 8, 64), ("dups", 64, 2048, 64, 64), ("dups", 32, 1024, 7, 7), ("offset", 64, 1024, 16, 16),
    ("clusters", 64, 2048, 64, 64), ("normal", 64, 1152, 10, 10),
    # round 3: any 1024 <= N <= 16384 (candidate rows padded to a multiple of 128) and 64 < k <= 128 -- the reference's own
    # default shape is N = 7000, k = 80 (option_new.py:58, M4:544-550) -- and clouds stored cluster by cluster in rotation
    ("normal", 64, 1100, 80, 80), ("relu", 64, 7000, 80, 80), ("clusters", 32, 1500, 100, 100), ("dups", 64, 1300, 128, 128),
    ("normal", 128, 2000, 20, 100), ("blobs", 64, 4096, 64, 64), ("blobs", 64, 7000, 80, 80), ("flat", 64, 4096, 64, 64),
    ("flat", 32, 3000, 80, 80)])
def test_knn_feature_prefilter_matches_exact_kernel_and_oracle(dev, kind, C, N, k1, k2):
    """csrc/knn_filter.hip (bf16 matrix-core prefilter + exact f32 re-rank + exhaustive fallback) returns the SAME
    indices as the exact matrix-core kernel (csrc/knn.hip) and as the CPU oracle, whatever the data does to the
    prefilter: duplicates (exact ties -> lowest index), clusters, and clouds whose order is decided by the reference's
    own f32 rounding (every query then takes the exhaustive path)."""
    import ctypes
    from gcanet_amd import _lib, dgcnn
    x = _feature_cloud(kind, 2, C, N, 11 + C + N)
    xd = x.to(dev)
    assert _lib.lib().gcn_knn_feature_supported(2, N, C, k2) == 1
    st = {}
    new = dgcnn.knn_feature_pm(xd.transpose(1, 2).contiguous(), k1, k2, stats=st)
    old = dgcnn._knn_model(xd, k1, k2, 0)
    assert torch.equal(new, old)
    np.testing.assert_array_equal(new.cpu().numpy(), oracle.knn_model(x.numpy(), k1, k2, 0))
    assert torch.equal(dgcnn.knn(xd, k1, k2), old)                      # the drop-in entry takes the same path
    print("%s C=%d N=%d k=%d: %d of %d queries flagged, %.0f candidates per query" % (kind, C, N, k2, st["flagged"], 2 * N, st["candidates"] / (2 * N)))
    if kind in ("normal", "relu", "blobs", "flat", "clusters"):
        # (a handful of queries per 10^4 miss the proof by a hair -- a sample order statistic at the low end of its
        # spread -- and take the exhaustive stage; "clusters": the members of a cluster whose k-th neighbour sits in
        # another cluster right at the threshold)
        limit = 4 + N // 500 if kind != "clusters" else 2 * N // 50
        assert st["flagged"] <= limit, "the prefilter should prove (nearly) every query of data it can resolve"
        assert st["candidates"] / (2 * N) < 6 * k2 + 64
    if kind == "offset":
        assert st["flagged"] > 0, "this cloud is meant to exercise the exhaustive path"


@pytest.mark.parametrize("kind,C,N,k1,k2,long_list", [
    ("offset2", 64, 4096, 16, 16, None),     # every query flagged, 8192 > the default list limit -> matrix-core search
    ("offset2", 32, 2048, 20, 20, 0), ("offset2", 128, 1024, 64, 64, 0), ("halfoffset", 64, 2048, 8, 64, 0),
    ("normal", 64, 1024, 16, 16, 0),         # nothing flagged: the gated kernels are no-ops
    ("offset2", 64, 8192, 64, 64, None),     # the size bench.py runs
    ("offset", 64, 2048, 64, 64, None),      # a few hundred flagged: the sliced short-list stage
    ("offset2", 64, 1500, 80, 80, None), ("halfoffset", 32, 1100, 20, 100, None)])   # k > 64 / ragged N: the VALU selection kernel
def test_knn_feature_long_fallback_list_runs_on_the_matrix_cores(dev, monkeypatch, kind, C, N, k1, k2, long_list):
    """Clouds whose neighbours sit closer than the bf16 prefilter resolves put (nearly) every query on the fallback
    list; beyond KNNF_LONG_LIST entries the list is searched by the exact f32 matrix-core kernel in its flagged mode
    (waves without a flagged query idle, unflagged results are left alone).  GCANET_KNN_LONG_LIST=0 forces that kernel
    for any non-empty list, so partly flagged clouds exercise the per-wave skip and the per-query write mask."""
    import ctypes
    from gcanet_amd import _lib, dgcnn
    if long_list is not None:
        monkeypatch.setenv("GCANET_KNN_LONG_LIST", str(long_list))
    if kind == "halfoffset":                 # every other block of 24 points comes from the unresolvable cloud
        x = _feature_cloud("normal", 2, C, N, 5 + C + N)
        y = _feature_cloud("offset2", 2, C, N, 6 + C + N)
        sel = (torch.arange(N) // 24) % 2 == 0
        x[:, :, sel] = y[:, :, sel] - 8.0 + 0.5
    else:
        x = _feature_cloud(kind, 2, C, N, 11 + C + N)
    xd = x.to(dev)
    st = {}
    new = dgcnn.knn_feature_pm(xd.transpose(1, 2).contiguous(), k1, k2, stats=st)
    old = dgcnn._knn_model(xd, k1, k2, 0)                   # the exact kernel, unflagged (itself checked against the oracle)
    assert torch.equal(new, old)
    if N <= 4096:
        np.testing.assert_array_equal(new.cpu().numpy(), oracle.knn_model(x.numpy(), k1, k2, 0))
    print("%s C=%d N=%d k=%d: %d of %d queries flagged" % (kind, C, N, k2, st["flagged"], 2 * N))
    if kind == "offset2":
        assert st["flagged"] > (4096 if long_list is None and N >= 4096 else N)
    if kind == "offset":
        assert 0 < st["flagged"] <= 4096
    if kind == "halfoffset":
        assert 0 < st["flagged"] < 2 * N
    if kind == "normal":
        assert st["flagged"] <= 2


@pytest.mark.parametrize("B,C,N,k1,k2,metric", [
    (1, 64, 1024, 1, 1, 0),          # smallest served cloud, k = 1 (the query itself)
    (3, 32, 1025, 5, 5, 0),          # one row into the padding tile, three clouds (not a multiple of the 8 XCDs)
    (1, 128, 16384, 128, 128, 0),    # largest served cloud, largest k
    (1, 64, 16383, 16, 128, 0),      # ragged at the top end, dilated pick
    (2, 6, 1025, 128, 128, 1), (1, 6, 16383, 80, 80, 1), (3, 3, 5000, 1, 1, 0), (2, 3, 9000, 100, 100, 0)])
def test_knn_filters_size_and_k_limits(dev, B, C, N, k1, k2, metric):
    """The edges of what the threshold / filter / re-rank paths accept (1024 <= N <= 16384, k <= 128, any batch):
    bit-identical to the C oracle, feature space (csrc/knn_filter.hip) and xyz / xyz+normal (csrc/knn_normal.hip)."""
    from gcanet_amd import _lib, dgcnn
    g = torch.Generator().manual_seed(B * 1000 + N + k2)
    x = torch.randn(B, C, N, generator=g)
    if metric == 1:
        x[:, :3] = torch.rand(B, 3, N, generator=g)
        x[:, 3:] = torch.nn.functional.normalize(x[:, 3:], dim=1)
    elif C == 3:
        x = torch.rand(B, 3, N, generator=g)
    if C >= 32:
        assert _lib.lib().gcn_knn_feature_supported(B, N, C, k2) == 1
    else:
        assert _lib.lib().gcn_knn_normal_supported(B, N, k2) == 1
    fn = dgcnn.knn_points_normals if metric == 1 else dgcnn.knn
    idx = fn(x.to(dev), k1, k2).cpu().numpy()
    np.testing.assert_array_equal(idx, oracle.knn_model(x.numpy(), k1, k2, metric))


def test_knn_feature_unsupported_shapes_use_the_exact_kernel(dev):
    from gcanet_amd import _lib, dgcnn
    lib = _lib.lib()
    assert lib.gcn_knn_feature_supported(2, 1000, 64, 16) == 0           # N < 1024
    assert lib.gcn_knn_feature_supported(2, 2048, 64, 129) == 0          # k > 128
    assert lib.gcn_knn_feature_supported(2, 2048, 48, 16) == 0           # channel count
    assert lib.gcn_knn_feature_supported(3, 7000, 64, 80) == 1           # the reference's default shape (option_new.py:58)
    assert dgcnn.knn_feature_pm(torch.randn(2, 1000, 64, device=dev), 16, 16) is None
    with pytest.raises(RuntimeError, match="unsupported shape"):
        idx = torch.empty(2, 1000, 16, dtype=torch.int64, device=dev)
        ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
        _lib.call("gcn_knn_feature", _lib.ptr(torch.randn(2, 1000, 64, device=dev)), 2, 1000, 64, 16, 16, _lib.ptr(idx),
                  _lib.ptr(ws), _lib.stream_of(idx))
