"""Feature-space kNN (csrc/knn_filter.hip) on clouds the bf16 prefilter cannot resolve: time of the whole entry point
when (nearly) every query needs the exhaustive search.  python tools/knn_fallback_bench.py [C N B k]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib, dgcnn  # noqa: E402

C, N, B, k = [int(v) for v in (sys.argv[1:5] + ["64", "8192", "8", "64"][len(sys.argv) - 1:])]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
clouds = {
    "uniform": torch.randn(B, N, C, generator=g),
    "flat patches": torch.randn(B, 64, C, generator=g)[:, torch.arange(N) % 64] + 1e-4 * torch.randn(B, N, C, generator=g),
    "offset": torch.randn(B, N, C, generator=g) * 0.05 + 4.0,
    "blobs": torch.randn(B, 64, C, generator=g)[:, torch.arange(N) % 64] + 0.03 * torch.randn(B, N, C, generator=g),
}
for name, x in clouds.items():
    x = x.to(dev)
    for _ in range(2):
        idx = dgcnn.knn_feature_pm(x, k, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        idx = dgcnn.knn_feature_pm(x, k, k)
    e1.record()
    torch.cuda.synchronize()
    _st = {}
    dgcnn.knn_feature_pm(x, k, k, stats=_st)
    fl, ca = ctypes.c_long(_st["flagged"]), ctypes.c_long(_st["candidates"])
    same = torch.equal(idx, dgcnn._knn_model(x.transpose(1, 2).contiguous(), k, k, 0))
    print("%-14s %7.3f ms  flagged %6d of %d  identical to the exact kernel: %s" % (name, e0.elapsed_time(e1) / 5, fl.value, B * N, same))
