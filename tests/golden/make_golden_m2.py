#!/usr/bin/env python3
"""Golden vectors for the offset module of the reference's variant M2 (models/dgcnn-hais-concat-direct-2.py:296-462),
produced by running the reference's OWN source text (build container only; the .npz is committed).

M2 is not importable (spconv, missing models/backbone.py), so -- as make_golden.py does for M4 -- the three
self-contained definitions `inst_and_seg_dist`, `KPAM`, `OFFSET_PRED_MODULE` are compiled unmodified from the file
into a scratch namespace.  The two native ops they call, `group_points` (models/search_knn.py:23-39 -> KNN_CUDA +
pointnet2 grouping_operation) and `grouping_operation`, are CUDA-only; the namespace binds them to the CPU oracle's
restatement (oracle.KNN_forward = knn.cu:29-183; a torch gather for group_points_gpu.cu:8-28 so that autograd gives
the scatter-add gradient of :43-64).

Usage:  python tests/golden/make_golden_m2.py
"""
import ast
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
import oracle  # noqa: E402


def grouping_operation(features, idx):
    """(B,C,N), (B,npoint,nsample) int32 -> (B,C,npoint,nsample): out[b,c,j,s] = features[b,c,idx[b,j,s]]."""
    B, C, N = features.shape
    _, P, S = idx.shape
    flat = idx.long().view(B, 1, P * S).expand(-1, C, -1)
    return torch.gather(features, 2, flat).view(B, C, P, S)


def group_points(group_size, point_cloud, query_cloud, point_features=None):
    """models/search_knn.py:23-39 on the oracle's KNN (indices 0-based int64 (B,k,Q), ties -> lowest index)."""
    _, I = oracle.KNN_forward(point_cloud.detach().numpy(), query_cloud.detach().numpy(), group_size, False)
    idx = torch.from_numpy(I).permute(0, 2, 1).type(torch.int32).contiguous()
    gp = grouping_operation(point_cloud, idx)
    gf = None if point_features is None else grouping_operation(point_features, idx)
    return gp, gf, idx


def main():
    torch.set_num_threads(1)
    src = open(os.path.join(REF, "models", "dgcnn-hais-concat-direct-2.py")).read()
    want = {"inst_and_seg_dist", "KPAM", "OFFSET_PRED_MODULE"}
    body = [n for n in ast.parse(src).body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in body} == want
    ns = {"torch": torch, "nn": nn, "np": np, "F": torch.nn.functional, "group_points": group_points,
          "grouping_operation": grouping_operation}
    exec(compile(ast.Module(body=body, type_ignores=[]), "M2-extract", "exec"), ns)
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(21)
    off = ns["OFFSET_PRED_MODULE"](nn_nb=60, sampling_ratio=120)
    with torch.no_grad():
        off.bn1.weight.copy_(torch.randn(128))             # mixed-sign gains: max- and min-routing
        off.bn1.bias.copy_(torch.randn(128) * 0.1)
    B, N = 2, 200
    points = torch.rand(B, N, 3, generator=g, requires_grad=True)
    feat = torch.randn(B, N, 128, generator=g, requires_grad=True)
    sem = torch.randn(B, N, 10, generator=g, requires_grad=True)
    ins = torch.randn(B, N, 64, generator=g, requires_grad=True)
    o = off(points, feat, sem, ins, None)
    go = torch.randn(o.shape, generator=g)
    (o * go).sum().backward()
    out = {"sd_" + k: v.detach().numpy().copy() for k, v in off.state_dict().items()}
    out.update({"points": points.detach().numpy(), "feat": feat.detach().numpy(), "sem": sem.detach().numpy(),
                "ins": ins.detach().numpy(), "out": o.detach().numpy(), "gout": go.numpy(),
                "dpoints": points.grad.numpy(), "dfeat": feat.grad.numpy(), "dins": ins.grad.numpy(),
                "dsem_is_none": np.array(sem.grad is None or float(sem.grad.abs().max()) == 0.0)})
    for n_, p_ in off.named_parameters():
        if p_.grad is not None:
            out["grad_" + n_] = p_.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "m2_offset_golden.npz"), **out)
    print("m2_offset_golden.npz", os.path.getsize(os.path.join(OUT, "m2_offset_golden.npz")), "bytes; out", o.shape,
          "grads:", sorted(k for k in out if k.startswith("grad_")))


if __name__ == "__main__":
    main()
