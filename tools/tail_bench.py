#!/usr/bin/env python3
"""The stages after the hot path (M4:737-777) on a synthetic scene with real proposals: forward_grouping (device) ->
clusters_voxelization -> sparse instance head forward+backward.  usage: tail_bench.py [B N P reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd.grouping import clusters_voxelization, forward_grouping_device  # noqa: E402
from gcanet_amd.sparseconv import InstanceHead, SparseConvTensor  # noqa: E402
from grouping_bench import scene  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / reps * 1e3


def main():
    B, N, P, reps = (int(a) for a in (sys.argv[1:5] + ["8", "8192", "10", "5"][len(sys.argv) - 1:]))
    args = scene(0, B, N, P, 40, 0.3)
    kw = dict(semantic_classes=P, radius=0.03, similarity_threshold_inst=0.989, similarity_threshold_para=0.0,
              mean_active=300, min_npoint=50)
    (pi, po), t_group = timed(lambda: forward_grouping_device(*args, **kw), reps)
    if po.shape[0] > 201:
        po = po[:201]
        pi = pi[:int(po[-1])]
    feats = args[6].reshape(B * N, -1).clone().requires_grad_(True)
    coords = args[3]
    r = (torch.full((3,), 0.5), torch.full((3,), 0.5))
    vox, t_vox = timed(lambda: clusters_voxelization(pi, po, feats, coords, scale=64, spatial_shape=64, rand_quantize=True, rand=r), reps)
    vf, vc, shape, nb, inst_map = vox
    head = InstanceHead(64, P).cuda()
    im = inst_map.cuda()

    def head_step():
        x = vf.detach().clone().requires_grad_(True)
        _, cls, iou, mask = head(SparseConvTensor(x, vc, shape, nb), im)
        (cls.pow(2).mean() + iou.pow(2).mean() + mask.pow(2).mean()).backward()

    _, t_head = timed(head_step, reps)
    print("B=%d N=%d: %d proposals, %d members, %d voxels | grouping %.2f ms, voxelisation %.2f ms, instance head fwd+bwd %.2f ms"
          % (B, N, po.shape[0] - 1, pi.shape[0], vf.shape[0], t_group, t_vox, t_head))


if __name__ == "__main__":
    main()
