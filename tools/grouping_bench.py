#!/usr/bin/env python3
"""forward_grouping: literal per-(cloud, class) path vs the fused device path, B clouds x N points, P classes.
usage: grouping_bench.py [B N P reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd.grouping import forward_grouping, forward_grouping_device  # noqa: E402


def scene(seed, B, N, P, nblob, extent):
    rng = np.random.default_rng(seed)
    centers = rng.random((B, nblob, 3))
    which = rng.integers(0, nblob, (B, N))
    dirs = rng.standard_normal((B, nblob, 2, 3))
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    uv = rng.random((B, N, 2)) * np.array([extent, extent / 2])
    ar = np.arange(B)[:, None]
    xyz = centers[ar, which] + 0.004 * rng.standard_normal((B, N, 3)) + np.einsum("bnk,bnkd->bnd", uv, dirs[ar, which])
    sem = rng.standard_normal((B * N, P)) * 0.3 + 6 * np.eye(P)[(which % P).reshape(-1)]
    off = 0.001 * rng.standard_normal((B * N, 3))
    par = rng.standard_normal((B, N, 22)) * 0.01
    feat = np.eye(64)[which % 64] + 0.01 * rng.standard_normal((B, N, 64))
    f = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    return f(sem), f(off), torch.arange(B).repeat_interleave(N).cuda(), f(xyz.reshape(-1, 3)), torch.zeros(B, N, P), f(par), f(feat)


def main():
    B, N, P, reps = (int(a) for a in (sys.argv[1:5] + ["8", "8192", "10", "3"][len(sys.argv) - 1:]))
    args = scene(0, B, N, P, 40, 0.3)
    kw = dict(semantic_classes=P, radius=0.03, similarity_threshold_inst=0.989, similarity_threshold_para=0.0,
              mean_active=300, min_npoint=50)
    for name, fn in (("device", forward_grouping_device), ("literal", forward_grouping)):
        fn(*args, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            pi, po = fn(*args, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-8s %8.2f ms   clusters %d  members %d" % (name, dt * 1e3, po.numel() - 1, pi.shape[0]), flush=True)
        if name == "device":
            dpi, dpo = pi, po
    print("identical:", bool(torch.equal(dpi, pi) and torch.equal(dpo, po)))


if __name__ == "__main__":
    main()
