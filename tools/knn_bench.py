"""Feature-space kNN: prefilter path (csrc/knn_filter.hip) vs the exact matrix-core kernel (knn.hip), same inputs.
python tools/knn_bench.py [C] [N] [B] [k] [kind]   kind: normal | relu (post-activation-like) | offset (large mean)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib, dgcnn  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k = int(sys.argv[4]) if len(sys.argv) > 4 else 64
kind = sys.argv[5] if len(sys.argv) > 5 else "normal"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(C + N)
x = torch.randn(B, C, N, generator=g)
if kind == "relu":
    x = torch.nn.functional.leaky_relu(x + 0.5, 0.2)
elif kind == "offset":
    x = x * 0.1 + 5.0
x = x.to(dev)
x_pm = x.transpose(1, 2).contiguous()


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, r


ms_old, idx_old = timed(lambda: dgcnn._knn_model(x, k, k, 0))
ms_new, idx_new = timed(lambda: dgcnn.knn_feature_pm(x_pm, k, k))
same = torch.equal(idx_old, idx_new)
_st = {}
dgcnn.knn_feature_pm(x_pm, k, k, stats=_st)
fl, ca = ctypes.c_long(_st["flagged"]), ctypes.c_long(_st["candidates"])
print("C=%d N=%d B=%d k=%d %s: exact kernel %.3f ms, prefilter path %.3f ms (%.2fx), identical=%s, fallback queries %d, "
      "candidates/query %.1f" % (C, N, B, k, kind, ms_old, ms_new, ms_old / ms_new, same, fl.value, ca.value / (B * N)))
if not same:
    bad = (idx_old != idx_new).any(-1)
    print("  rows differing:", int(bad.sum()), "first:", bad.nonzero()[:5].tolist())
    sys.exit(1)
