"""Pin the oracle: CPU restatement vs (a) the reference's own known-answer table
(models/search_knn.py:183-243) and (b) golden vectors produced by running the reference's
importable Python (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_model as R
from util import knn_rows_equivalent, pn_metric64, sqdist64

HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------- known-answer table
def _soft_projection(point_cloud, query_cloud, k, sigma, feats=None):
    """SoftProjection.propagate / project (search_knn.py:96-174) on top of the oracle's
    KNN_CUDA + group_points restatements.  point_cloud (1,3,P), query_cloud (1,3,Q)."""
    _, idx = oracle.KNN_forward(point_cloud, query_cloud, k)          # (1,k,Q)
    idx = np.ascontiguousarray(idx.transpose(0, 2, 1)).astype(np.int32)
    gp = oracle.group_points(point_cloud, idx)                        # (1,3,Q,k)
    d = ((gp - query_cloud[..., None]) ** 2).sum(1, keepdims=True) / sigma
    w = np.exp(-d - (-d).max(-1, keepdims=True))
    w = w / w.sum(-1, keepdims=True)
    src = gp if feats is None else oracle.group_points(feats, idx)
    return (src * w).sum(-1)


def test_search_knn_known_answer():
    ka = json.load(open(os.path.join(HERE, "golden", "search_knn_known_answer.json")))
    pc = np.asarray(ka["point_cloud"], np.float32).T[None]
    qc = np.asarray(ka["query_cloud"], np.float32).T[None]
    pf = np.asarray(ka["point_features"], np.float32).T[None]
    prop = _soft_projection(pc, qc, 3, 1.0, pf)[0].T                  # (Q, 5)
    np.testing.assert_allclose(prop, np.asarray(ka["expected_features_nn_3"]), atol=1.5e-3)
    proj = _soft_projection(qc, pc, 3, ka["project_sigma"])[0].T      # roles swapped (search_knn.py:279-281)
    np.testing.assert_allclose(proj, np.asarray(ka["expected_nn_cloud"]), atol=1.5e-3)
    prop1 = _soft_projection(pc, qc, 1, 1.0, pf)[0].T
    np.testing.assert_allclose(prop1, np.asarray(ka["expected_features_nn_1"]), atol=1e-6)


def test_knn_cuda_oracle_vs_kdtree_property():
    """The reference's own test property (KNN_CUDA/tests/test_knn_cuda.py:32-47): distances equal
    sklearn KDTree's to 3 decimals."""
    from sklearn.neighbors import KDTree
    rng = np.random.default_rng(0)
    for (nr, nq, k) in ((1000, 50, 10), (30, 50, 2), (3001, 20, 400)):
        ref = rng.random((nr, 5)).astype(np.float32)
        qry = rng.random((nq, 5)).astype(np.float32)
        d, i = oracle.knn_cuda(ref.T.copy(), qry.T.copy(), k)
        dk, ik = KDTree(ref).query(qry, k=k)
        np.testing.assert_almost_equal(d.T, dk, decimal=3)


# ----------------------------------------------------------------- in-model kNN vs imported reference
@pytest.mark.parametrize("key,k1,k2,metric,exact", [
    ("knn_grid", 16, 16, 0, True), ("knn_rand", 16, 16, 0, False), ("knn_feat", 8, 8, 0, False),
    ("knnpn_grid", 16, 16, 1, True), ("knnpn_rand", 16, 16, 1, False)])
def test_knn_model_vs_reference(golden, key, k1, k2, metric, exact):
    x = golden[key + "_x"]
    ref = golden[key + "_idx_k%d" % k1]
    idx = oracle.knn_model(x, k1, k2, metric)
    assert idx.shape == ref.shape and idx.dtype == np.int64
    for b in range(x.shape[0]):
        fn = sqdist64(x[b]) if metric == 0 else pn_metric64(x[b])
        # grid inputs: every summation order is exact, so differing rows must be exact ties
        tol = dict(rtol=0, atol=0) if exact else dict(rtol=1e-6, atol=1e-9)
        ident, tie, bad = knn_rows_equivalent(idx[b], ref[b], fn, **tol)
        assert bad == 0, (key, b, ident, tie, bad)
        assert ident >= 0.98 * idx.shape[1]


def test_knn_model_dilated_pick(golden):
    idx = oracle.knn_model(golden["knn_grid_x"], 4, 16, 0)
    full = oracle.knn_model(golden["knn_grid_x"], 16, 16, 0)
    np.testing.assert_array_equal(idx, full[:, :, ::4])
    ref = golden["knn_grid_idx_k4_16"]
    assert (idx == ref).all(-1).mean() > 0.98


# ----------------------------------------------------------------- graph features (bit-exact gathers)
def test_graph_features_bitexact(golden):
    x = torch.from_numpy(golden["knn_feat_x"])
    f = R.get_graph_feature(x, idx=torch.from_numpy(golden["knn_feat_idx_k8"]))
    np.testing.assert_array_equal(f.numpy(), golden["ggf_feat_out"])
    xp = torch.from_numpy(golden["knnpn_rand_x"])
    idxp = torch.from_numpy(golden["knnpn_rand_idx_k16"])
    np.testing.assert_array_equal(R.get_graph_feature_with_normals(xp, idx=idxp).numpy(), golden["ggfn_out"])
    np.testing.assert_array_equal(R.get_graph_feature_with_normals_g(xp, idx=idxp).numpy(), golden["ggfng_out"])


def test_edgeconv_block_fwd_bwd(golden):
    g = golden
    x = torch.from_numpy(g["ec_x"]).requires_grad_()
    w = torch.from_numpy(g["ec_w"]).requires_grad_()
    ga = torch.from_numpy(g["ec_gamma"]).requires_grad_()
    be = torch.from_numpy(g["ec_beta"]).requires_grad_()
    y = R.edgeconv_block(x, torch.from_numpy(g["ec_idx"]), w, ga, be, 2)
    np.testing.assert_allclose(y.detach().numpy(), g["ec_y"], rtol=1e-5, atol=1e-5)
    (y * torch.from_numpy(g["ec_gout"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["ec_dx"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(w.grad.numpy(), g["ec_dw"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ga.grad.numpy(), g["ec_dgamma"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(be.grad.numpy(), g["ec_dbeta"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("mode", [5, 0])
def test_dgcnn_encoder(golden, mode):
    p = "enc%d_" % mode
    sd = {k[len(p) + 3:]: golden[k] for k in golden.files if k.startswith(p + "sd_")}
    x = torch.from_numpy(golden[p + "x"])
    idxs = [golden[p + "idx%d" % i] for i in (1, 2, 3)]
    x4, xf, _ = R.dgcnn_encoder(x, sd, 8, mode, idxs=idxs)
    np.testing.assert_allclose(xf.numpy(), golden[p + "xf"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(x4.numpy(), golden[p + "x4"], rtol=1e-4, atol=1e-5)
    # and with the oracle's own kNN (ties -> lowest index): layer 1 lists agree up to exact ties
    _, _, used = R.dgcnn_encoder(x, sd, 8, mode)
    fn = pn_metric64 if mode == 5 else sqdist64
    for b in range(x.shape[0]):
        ident, tie, bad = knn_rows_equivalent(used[0][b].numpy(), idxs[0][b], fn(golden[p + "x"][b]))
        assert bad == 0


def test_offset_module_pieces(golden):
    g = golden
    np.testing.assert_allclose(R.compute_batch_adjacency_matrix(torch.from_numpy(g["adj_x"])).numpy(), g["adj_out"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(R.cos_dist(torch.from_numpy(g["cos_a"]), torch.from_numpy(g["cos_b"])).numpy(),
                               g["cos_out"], rtol=1e-5, atol=1e-6)
    sd = {k[7:]: g[k] for k in g.files if k.startswith("off_sd_")}
    pts = torch.from_numpy(g["off_points"]).requires_grad_()
    feat = torch.from_numpy(g["off_feat"]).requires_grad_()
    emb = torch.from_numpy(g["off_emb"]).requires_grad_()
    o = R.offset_pred_module(pts, feat, emb, sd)
    np.testing.assert_allclose(o.detach().numpy(), g["off_out"], rtol=1e-4, atol=1e-4)
    (o * torch.from_numpy(g["off_gout"])).sum().backward()
    np.testing.assert_allclose(pts.grad.numpy(), g["off_dpoints"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(feat.grad.numpy(), g["off_dfeat"], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(emb.grad.numpy(), g["off_demb"], rtol=1e-3, atol=2e-4)


# ----------------------------------------------------------------- attention stacks
def test_transformer(att_golden):
    g = att_golden
    sd = {k[6:]: g[k] for k in g.files if k.startswith("tr_sd_")}
    x = torch.from_numpy(g["tr_x"]).requires_grad_()
    y = R.transformer(x, sd, depth=2, heads=4)
    np.testing.assert_allclose(y.detach().numpy(), g["tr_y"], rtol=1e-4, atol=1e-5)
    (y * torch.from_numpy(g["tr_gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["tr_dx"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag,kw", [("qd", dict(iter_pred=False, attn_mask=False)),
                                    ("qdi", dict(iter_pred=True, attn_mask=True))])
def test_query_decoder(att_golden, tag, kw):
    g = att_golden
    sd = {k[len(tag) + 4:]: g[k] for k in g.files if k.startswith(tag + "_sd_")}
    offs = [int(v) for v in g[tag + "_offsets"]]
    with torch.no_grad():
        o = R.query_decoder(torch.from_numpy(g[tag + "_x"]), offs, sd, num_layer=2, nhead=4, **kw)
    for k_ in ("labels", "scores", "parameters"):
        np.testing.assert_allclose(o[k_].numpy(), g[tag + "_" + k_], rtol=1e-4, atol=1e-4)
    for i, m in enumerate(o["masks"]):
        np.testing.assert_allclose(m.numpy(), g[tag + "_mask%d" % i], rtol=1e-4, atol=1e-4)
    if kw["iter_pred"]:
        for li, aux in enumerate(o["aux_outputs"]):
            np.testing.assert_allclose(aux["labels"].numpy(), g[tag + "_aux%d_labels" % li], rtol=1e-4, atol=1e-4)


def test_transformer_masked_rows_follow_the_reference():
    """Padded tokens: the reference's finite mask fill makes a fully masked query row attend uniformly
    (transformer.py:57-67); golden from the reference's own module (tests/golden/make_golden_tr_mask.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transformer_mask_golden.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd_")}
    y = R.transformer(torch.from_numpy(g["x"]), sd, depth=2, heads=4, mask=torch.from_numpy(g["mask"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-5, atol=1e-5)


def test_octree_ball_query_restatement_is_a_radius_query():
    """CPU: the octree restatement (octree_ball_query.cpp/.cu) returns, per point, exactly the in-radius set of the
    brute-force restatement (the octree only prunes), ordered by (leaf, index)."""
    import oracle
    rng = np.random.default_rng(0)
    c = rng.random((1500, 3)).astype(np.float32)
    idx, sl, leaf = oracle.octree_ball_query(c, 50, 0.08)
    bi, bo = np.zeros(1500, np.int32), np.array([0, 1500], np.int32)
    i2, s2 = oracle.ballquery_batch_p(c, bi, bo, 0.08, 50)[:2]
    for p in range(1500):
        a = idx[sl[p, 0]:sl[p, 0] + sl[p, 1]]
        assert set(a) == set(i2[s2[p, 0]:s2[p, 0] + s2[p, 1]])
        key = leaf[a].astype(np.int64) * 1500 + a
        assert (np.diff(key) > 0).all()
