"""forward_grouping_device with similarity thresholds <= 0: the row-free decision of csrc/softgroup.hip:
ballquery_sim_kernel against the same kernel forced to evaluate every similarity from the rows (GCANET_BQ_EXACT=1).
Random blob scenes, feature sets from well separated to identical up to 1e-7, far from the origin, mixed scales."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd.grouping import forward_grouping_device  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(cases):
    B, N, P = int(rng.integers(1, 4)), int(rng.integers(800, 4000)), int(rng.integers(2, 5))
    nb = int(rng.integers(2, 12))
    cen = rng.random((B, nb, 3))
    which = rng.integers(0, nb, (B, N))
    xyz = (cen[np.arange(B)[:, None], which] + 10.0 ** rng.uniform(-3, -1.7) * rng.standard_normal((B, N, 3))).astype(np.float32)
    sem = (rng.standard_normal((B * N, P)) * 0.3 + 6 * np.eye(P)[(which % P).reshape(-1)]).astype(np.float32)
    off = (0.001 * rng.standard_normal((B * N, 3))).astype(np.float32)
    bidx = np.repeat(np.arange(B), N).astype(np.int64)

    def feats(C):
        base = rng.standard_normal((B, nb, C))[np.arange(B)[:, None], which] * 10.0 ** rng.uniform(-2, 1)
        f = base * (rng.random() < 0.7) + 10.0 ** rng.uniform(-7, 0) * rng.standard_normal((B, N, C))
        if rng.random() < 0.4:
            f = f + 10.0 ** rng.uniform(0, 3)
        return f.astype(np.float32)

    par, feat = feats(22), feats(int(rng.choice([8, 16, 64])))
    thr_i, thr_p = float(rng.choice([0.0, -0.3, 0.9])), float(rng.choice([0.0, -1.0]))
    t = lambda a: torch.from_numpy(a).to(dev)
    args = (t(sem), t(off), t(bidx), t(xyz.reshape(-1, 3)), torch.zeros(B, N, P), t(par), t(feat))
    kw = dict(semantic_classes=P, radius=0.03, similarity_threshold_inst=thr_i, similarity_threshold_para=thr_p,
              mean_active=300, min_npoint=20)
    os.environ["GCANET_BQ_EXACT"] = "0"
    a = forward_grouping_device(*args, **kw)
    os.environ["GCANET_BQ_EXACT"] = "1"
    b = forward_grouping_device(*args, **kw)
    if not (torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
        bad += 1
        print("MISMATCH case %d: thr %.1f/%.1f proposals %d vs %d" % (it, thr_i, thr_p, a[1].numel(), b[1].numel()), flush=True)
print("cases %d, mismatches %d" % (cases, bad))
