"""Random shapes and distributions through the feature-space kNN filter (csrc/knn_filter.hip) and the 3-D filter
(csrc/knn_normal.hip) against the exhaustive exact kernels of csrc/knn.hip: identical index lists expected."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import _lib, dgcnn  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0


def cloud(B, N, C):
    kind = rng.choice(["uniform", "blobs", "tight", "offset", "dups", "relu", "scaled"])
    x = rng.standard_normal((B, N, C))
    if kind == "blobs":
        nb = int(rng.integers(2, 100))
        x = rng.standard_normal((B, nb, C))[:, rng.integers(0, nb, N)] + 10.0 ** rng.uniform(-3, -0.5) * x
    elif kind == "tight":
        x = rng.standard_normal((B, 1, C)) + 10.0 ** rng.uniform(-5, -2) * x
    elif kind == "offset":
        x = x * 10.0 ** rng.uniform(-2, 0) + rng.uniform(1, 20)
    elif kind == "dups":
        x[:, rng.integers(0, N, N // 3)] = x[:, rng.integers(0, N, N // 3)]
    elif kind == "relu":
        x = np.maximum(x, 0) * (rng.random((1, 1, C)) < 0.7)
    elif kind == "scaled":
        x = x * 10.0 ** rng.uniform(-4, 3)
    return kind, torch.from_numpy(x.astype(np.float32)).to(dev)


for it in range(cases):
    B = int(rng.integers(1, 4))
    N = int(rng.choice([rng.integers(1024, 3000), rng.integers(3000, 9000), 4096, 7000, 8192]))
    if rng.random() < 0.6:
        C = int(rng.choice([32, 64, 128]))
        k2 = int(rng.choice([rng.integers(1, 129), 16, 64, 80, 128]))
        k1 = k2 if rng.random() < 0.7 else max(1, k2 // 2)
        kind, x = cloud(B, N, C)
        got = dgcnn.knn_feature_pm(x, k1, k2)
        if got is None:
            continue
        old = os.environ.get("GCANET_KNN_FILTER")
        ref = torch.empty_like(got)
        xx = torch.empty(B, N, device=dev)
        xc = x.transpose(1, 2).contiguous()
        with _lib.on_device(x):
            _lib.call("gcn_knn_model", _lib.ptr(xc), B, C, N, k1, k2, 0, _lib.ptr(ref), None, _lib.ptr(xx), None, _lib.stream_of(x))
        what = "feature C=%d" % C
    else:
        k2 = int(rng.choice([rng.integers(1, 129), 16, 64, 80]))
        k1 = k2
        metric = int(rng.integers(0, 2))
        C = 6 if metric == 1 else 3
        kind = "cloud"
        p = rng.random((B, 3, N))
        if rng.random() < 0.4:
            p = np.round(p * 64) / 64                        # many exact ties
        nrm = rng.standard_normal((B, 3, N))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        x = torch.from_numpy((np.concatenate([p, nrm], 1) if metric == 1 else p).astype(np.float32)).to(dev)
        got = dgcnn._knn_model(x, k1, k2, metric)
        ref = torch.empty_like(got)
        xx = torch.empty(B, N, device=dev)
        with _lib.on_device(x):                              # tile_ws = NULL: the exhaustive select kernel
            _lib.call("gcn_knn_model", _lib.ptr(x), B, C, N, k1, k2, metric, _lib.ptr(ref), None, _lib.ptr(xx), None, _lib.stream_of(x))
        what = "3-D metric %d" % metric
    if not torch.equal(got, ref):
        bad += 1
        d = (got != ref).any(-1).sum().item()
        print("MISMATCH case %d: %s %s B=%d N=%d k=%d/%d: %d rows differ" % (it, what, kind, B, N, k1, k2, d), flush=True)
print("cases %d, mismatches %d" % (cases, bad))
