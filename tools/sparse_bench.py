#!/usr/bin/env python3
"""Instance head (tiny U-Net of sparse convolutions, M4:611-616,1379-1392) forward+backward on synthetic proposals.
usage: sparse_bench.py [proposals voxels_per_proposal D reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib, sparseconv as S  # noqa: E402


def scene(P, V, D, C, seed=0):
    """P proposals, ~V voxels each on a thin random surface patch inside a D^3 grid (what a voxelised primitive looks like)."""
    g = torch.Generator().manual_seed(seed)
    idx = []
    cells = torch.stack(torch.meshgrid(*[torch.arange(D)] * 3, indexing="ij"), -1).view(-1, 3)
    for b in range(P):
        n = torch.randn(3, generator=g)
        d = (cells.float() - D / 2) @ (n / n.norm())
        cand = cells[d.abs() < 0.8]
        sel = cand[torch.randperm(cand.shape[0], generator=g)[:V]]
        idx.append(torch.cat([torch.full((sel.shape[0], 1), b), sel], 1))
    idx = torch.cat(idx).int()
    return torch.randn(idx.shape[0], C, generator=g), idx


def main():
    P, V, D, reps = (int(a) for a in (sys.argv[1:5] + ["200", "1000", "64", "5"][len(sys.argv) - 1:]))
    dev = torch.device("cuda")
    feats, idx = scene(P, V, D, 64)
    feats, idx = feats.to(dev), idx.to(dev)
    M = idx.shape[0]
    head = S.InstanceHead(64, 10).to(dev)
    inst_map = torch.randint(0, M, (P * 1500,), device=dev)

    def step():
        x = feats.clone().requires_grad_(True)
        _, cls, iou, mask = head(S.SparseConvTensor(x, idx, [D] * 3, P), inst_map)
        (cls.pow(2).mean() + iou.pow(2).mean() + mask.pow(2).mean()).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.enable_timing(False)
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nb = S.subm_rules(S.SparseConvTensor(feats, idx, [D] * 3, P))
    fill = float((nb >= 0).float().mean())
    # 12 submanifold convs: 4 x (64->64), 4 x (128->128) on the coarse sites, 1 x (128->64), 3 x (64->64); dense-equivalent
    # work 2*M*27*Cin*Cout each (counting absent neighbours), x3 for forward + both gradients
    print("proposals %d  voxels %d  grid %d^3  neighbour fill %.2f  |  fwd+bwd %.2f ms" % (P, M, D, fill, dt * 1e3))


if __name__ == "__main__":
    main()
