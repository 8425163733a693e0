#!/usr/bin/env python3
"""Generate golden vectors from the reference's own Python code (runs ONLY in the
build container, where /root/reference exists; the .npz/.json outputs are committed,
the reference never travels).

Sources executed:
  * models/sppnet.py      -- imported as a module (pure torch; twin of M4:30-161,455-534)
  * models/transformer.py, models/query_decoder.py -- imported as modules
  * models/dgcnn-hais-concat-direct-4.py -- NOT importable (needs spconv and a missing
    models/backbone.py).  Five self-contained top-level definitions are compiled from
    its source text, unmodified, into a scratch namespace: get_graph_feature_with_normals_g,
    compute_batch_adjacency_matrix, cos_dist, KPAM, OFFSET_PRED_MODULE (+ knn_points_normals
    which the first one calls).
The reference hard-codes torch.device('cuda') in get_graph_feature*; torch.device is
rebound to return the CPU device while those functions run.

Usage:  python tests/golden/make_golden.py      (writes next to this file)
"""
import ast
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

_real_device = torch.device


class _CpuDevice:
    def __enter__(self):
        torch.device = lambda *a, **k: _real_device("cpu")

    def __exit__(self, *a):
        torch.device = _real_device


def sd_to_np(module, prefix=""):
    return {prefix + k: v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grid_cloud(g, B, N, levels=1024):
    return torch.randint(0, levels, (B, 3, N), generator=g).float() / levels


def axis_normals(g, B, N):
    """Exactly representable unit normals (+-e_x, +-e_y, +-e_z)."""
    ax = torch.randint(0, 3, (B, N), generator=g)
    sg = torch.randint(0, 2, (B, N), generator=g).float() * 2 - 1
    n = torch.zeros(B, 3, N)
    n.scatter_(1, ax.unsqueeze(1), sg.unsqueeze(1))
    return n


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)
    sppnet = importlib.import_module("models.sppnet")
    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- 1. kNN index fixtures -------------------------------------------------
    x_grid = grid_cloud(g, 2, 256)
    out["knn_grid_x"] = x_grid.numpy()
    out["knn_grid_idx_k16"] = sppnet.knn(x_grid, 16, 16).numpy()
    out["knn_grid_idx_k4_16"] = sppnet.knn(x_grid, 4, 16).numpy()       # dilated pick
    x_rand = torch.rand(2, 3, 256, generator=g)
    out["knn_rand_x"] = x_rand.numpy()
    out["knn_rand_idx_k16"] = sppnet.knn(x_rand, 16, 16).numpy()
    x_feat = torch.randn(2, 64, 200, generator=g)
    out["knn_feat_x"] = x_feat.numpy()
    out["knn_feat_idx_k8"] = sppnet.knn(x_feat, 8, 8).numpy()
    pn_grid = torch.cat([grid_cloud(g, 2, 256), axis_normals(g, 2, 256)], 1)
    out["knnpn_grid_x"] = pn_grid.numpy()
    out["knnpn_grid_idx_k16"] = sppnet.knn_points_normals(pn_grid, 16, 16).numpy()
    nrm = torch.nn.functional.normalize(torch.randn(2, 3, 256, generator=g), dim=1)
    pn_rand = torch.cat([torch.rand(2, 3, 256, generator=g), nrm], 1)
    out["knnpn_rand_x"] = pn_rand.numpy()
    out["knnpn_rand_idx_k16"] = sppnet.knn_points_normals(pn_rand, 16, 16).numpy()

    # ---- 2. graph features with a GIVEN idx (pure gather: bit-exact) ------------
    with _CpuDevice():
        idx = out["knn_feat_idx_k8"]
        f = sppnet.get_graph_feature(x_feat, k1=8, k2=8, idx=torch.from_numpy(idx))
        out["ggf_feat_out"] = f.numpy()
        idxp = out["knnpn_rand_idx_k16"]
        f = sppnet.get_graph_feature_with_normals(pn_rand, k1=16, k2=16, idx=torch.from_numpy(idxp))
        out["ggfn_out"] = f.numpy()

    # ---- 3. EdgeConv block (graph feature -> conv -> GN -> LeakyReLU -> max) -----
    with _CpuDevice():
        torch.manual_seed(1)
        B, Cin, Cout, N, k = 2, 16, 32, 96, 8
        conv = nn.Sequential(nn.Conv2d(2 * Cin, Cout, 1, bias=False), nn.GroupNorm(2, Cout),
                             nn.LeakyReLU(negative_slope=0.2))
        with torch.no_grad():
            conv[1].weight.copy_(torch.randn(Cout))        # mixed-sign gamma exercises min AND max routing
            conv[1].bias.copy_(torch.randn(Cout) * 0.1)
        x = torch.randn(B, Cin, N, generator=g, requires_grad=True)
        idx = sppnet.knn(x.detach(), k, k)
        ef = sppnet.get_graph_feature(x, k1=k, k2=k, idx=idx)
        y = conv(ef).max(dim=-1)[0]
        gout = torch.randn(y.shape, generator=g)
        (y * gout).sum().backward()
        out.update({"ec_x": x.detach().numpy(), "ec_idx": idx.numpy(), "ec_w": conv[0].weight.detach().numpy()[:, :, 0, 0],
                    "ec_gamma": conv[1].weight.detach().numpy(), "ec_beta": conv[1].bias.detach().numpy(),
                    "ec_y": y.detach().numpy(), "ec_gout": gout.numpy(), "ec_dx": x.grad.numpy(),
                    "ec_dw": conv[0].weight.grad.numpy()[:, :, 0, 0], "ec_dgamma": conv[1].weight.grad.numpy(),
                    "ec_dbeta": conv[1].bias.grad.numpy()})

    # ---- 4. DGCNNEncoderGn mode 5 and mode 0, per-layer idx recorded ------------
    with _CpuDevice():
        for mode, cin in ((5, 6), (0, 3)):
            torch.manual_seed(2 + mode)
            enc = sppnet.DGCNNEncoderGn(mode=mode, input_channels=cin, nn_nb=8)
            N = 128
            if mode == 5:
                xin = torch.cat([grid_cloud(g, 2, N), axis_normals(g, 2, N)], 1)
            else:
                xin = grid_cloud(g, 2, N)
            x4, xf = enc(xin)
            # re-derive the per-layer neighbour lists the reference used
            if mode == 5:
                i1 = sppnet.knn_points_normals(xin, 8, 8)
            else:
                i1 = sppnet.knn(xin, 8, 8)
            i2 = sppnet.knn(xf[:, 0:64], 8, 8)
            i3 = sppnet.knn(xf[:, 64:128], 8, 8)
            p = "enc%d_" % mode
            out.update(sd_to_np(enc, p + "sd_"))
            out.update({p + "x": xin.numpy(), p + "x4": x4.detach().numpy(), p + "xf": xf.detach().numpy(),
                        p + "idx1": i1.numpy(), p + "idx2": i2.numpy(), p + "idx3": i3.numpy()})

    # ---- 5. M4-only definitions, compiled from source text ---------------------
    src = open(os.path.join(REF, "models", "dgcnn-hais-concat-direct-4.py")).read()
    tree = ast.parse(src)
    want = {"knn_points_normals", "get_graph_feature_with_normals_g", "compute_batch_adjacency_matrix",
            "cos_dist", "KPAM", "OFFSET_PRED_MODULE"}
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in body} == want
    ns = {"torch": torch, "nn": nn, "np": np, "F": torch.nn.functional}
    exec(compile(ast.Module(body=body, type_ignores=[]), "M4-extract", "exec"), ns)
    with _CpuDevice():
        f = ns["get_graph_feature_with_normals_g"](pn_rand, k1=16, k2=16, idx=torch.from_numpy(idxp))
        out["ggfng_out"] = f.numpy()
        pts = torch.rand(2, 40, 6, generator=g)
        out["adj_x"] = pts.numpy()
        out["adj_out"] = ns["compute_batch_adjacency_matrix"](pts).numpy()
        a = torch.randn(2, 50, 16, generator=g)
        b = torch.randn(2, 7, 16, generator=g)
        out["cos_a"], out["cos_b"] = a.numpy(), b.numpy()
        out["cos_out"] = ns["cos_dist"](a, b).numpy()
        torch.manual_seed(7)
        off = ns["OFFSET_PRED_MODULE"](nn_nb=30, sampling_ratio=120)
        with torch.no_grad():
            off.bn1.weight.copy_(torch.randn(128))
            off.bn1.bias.copy_(torch.randn(128) * 0.1)
        N = 160
        points = torch.rand(2, N, 3, generator=g, requires_grad=True)
        feat = torch.randn(2, N, 128, generator=g, requires_grad=True)
        emb = torch.randn(2, N, 64, generator=g, requires_grad=True)
        o = off(points, feat, emb)
        go = torch.randn(o.shape, generator=g)
        (o * go).sum().backward()
        out.update(sd_to_np(off, "off_sd_"))
        out.update({"off_points": points.detach().numpy(), "off_feat": feat.detach().numpy(),
                    "off_emb": emb.detach().numpy(), "off_out": o.detach().numpy(), "off_gout": go.numpy(),
                    "off_dpoints": points.grad.numpy(), "off_dfeat": feat.grad.numpy(), "off_demb": emb.grad.numpy()})

    np.savez_compressed(os.path.join(OUT, "dgcnn_golden.npz"), **out)

    # ---- 6. attention stacks ---------------------------------------------------
    att = {}
    tr = importlib.import_module("models.transformer")
    torch.manual_seed(11)
    T = tr.Transformer(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=64, dropout=0.0)
    x = torch.randn(2, 50, 32, generator=g, requires_grad=True)
    y = T(x)
    gy = torch.randn(y.shape, generator=g)
    (y * gy).sum().backward()
    att.update(sd_to_np(T, "tr_sd_"))
    att.update({"tr_x": x.detach().numpy(), "tr_y": y.detach().numpy(), "tr_gy": gy.numpy(), "tr_dx": x.grad.numpy()})
    for n_, p_ in T.named_parameters():
        att["tr_grad_" + n_] = p_.grad.numpy().copy()

    qd = importlib.import_module("models.query_decoder")
    for tag, kw in (("qd", dict(iter_pred=False, attn_mask=False)), ("qdi", dict(iter_pred=True, attn_mask=True, pe=True))):
        torch.manual_seed(13)
        Q = qd.QueryDecoder(num_layer=2, num_query=10, num_class=5, in_channel=16, d_model=32, nhead=4,
                            hidden_dim=64, **kw)
        Q.eval()
        x = torch.randn(70, 16, generator=g)
        offs = [0, 30, 70]
        with torch.no_grad():
            o = Q(x, offs)
        att.update(sd_to_np(Q, tag + "_sd_"))
        att[tag + "_x"] = x.numpy()
        att[tag + "_offsets"] = np.array(offs, np.int64)
        for k_ in ("labels", "scores", "parameters"):
            att[tag + "_" + k_] = o[k_].numpy()
        for i_, m_ in enumerate(o["masks"]):
            att[tag + "_mask%d" % i_] = m_.numpy()
        if "aux_outputs" in o:
            for li, aux in enumerate(o["aux_outputs"]):
                att[tag + "_aux%d_labels" % li] = aux["labels"].numpy()
                att[tag + "_aux%d_mask0" % li] = aux["masks"][0].numpy()
    np.savez_compressed(os.path.join(OUT, "attention_golden.npz"), **att)

    # ---- 7. known-answer table transcribed from models/search_knn.py:183-243 ----
    ka = {
        "source": "models/search_knn.py:183-243 (__main__ demo; 3-decimal expected values, k=3 / k=1)",
        "k": 3,
        "query_cloud": [[1, 0, 0], [0, 1, 0], [0, 0, 1], [5, 4, 4], [4, 5, 4], [4, 4, 5], [8, 7, 7], [7, 8, 7], [7, 7, 8]],
        "point_cloud": [[0, 0, 0], [1, 0, 0], [2, 0, 0], [5, 5, 5], [7, 7, 8], [7, 7, 8.5]],
        "point_features": [[1, 2, 3, 4, 5], [6, 7, 8, 9, 10], [11, 12, 13, 14, 15], [16, 17, 18, 19, 20],
                           [21, 22, 23, 24, 25], [26, 27, 28, 29, 30]],
        "expected_nn_cloud": [[0.333, 0.333, 0.333], [1, 0, 0], [1, 0, 0], [4.333, 4.333, 4.333], [7, 7, 8], [7, 7, 8]],
        "expected_features_nn_1": [[6, 7, 8, 9, 10], [1, 2, 3, 4, 5], [1, 2, 3, 4, 5], [16, 17, 18, 19, 20],
                                   [16, 17, 18, 19, 20], [16, 17, 18, 19, 20], [21, 22, 23, 24, 25],
                                   [21, 22, 23, 24, 25], [21, 22, 23, 24, 25]],
        "expected_features_nn_3": [[6.0, 7.0, 8.0, 9.0, 10.0], [2.459, 3.459, 4.459, 5.459, 6.459],
                                   [2.459, 3.459, 4.459, 5.459, 6.459], [16.0, 17.0, 18.0, 19.0, 20.0],
                                   [16.0, 17.0, 18.0, 19.0, 20.0], [16.0, 17.0, 18.0, 19.0, 20.0],
                                   [22.113, 23.113, 24.113, 25.113, 26.113], [22.113, 23.113, 24.113, 25.113, 26.113],
                                   [23.189, 24.189, 25.189, 26.189, 27.189]],
        "project_sigma": 0.01,
        "note": "propagate(point_cloud, point_features, query_cloud) at temperature 1.0 -> expected_features_nn_3; "
                "then sigma := 0.1**2 and project(query_cloud, point_cloud) -> expected_nn_cloud",
    }
    with open(os.path.join(OUT, "search_knn_known_answer.json"), "w") as f:
        json.dump(ka, f, indent=1)
    for fn in ("dgcnn_golden.npz", "attention_golden.npz", "search_knn_known_answer.json"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)), "bytes")


if __name__ == "__main__":
    main()
