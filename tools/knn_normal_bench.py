"""knn_points_normals: threshold + filter + re-rank (csrc/knn_normal.hip) vs the exhaustive kernel, same inputs.
python tools/knn_normal_bench.py [N] [B] [k] [kind]   kind: random (incoherent normals) | smooth (normals = f(position))"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
k = int(sys.argv[3]) if len(sys.argv) > 3 else 64
kind = sys.argv[4] if len(sys.argv) > 4 else "random"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(N + k)
p = torch.rand(B, N, 3, generator=g)
n = torch.randn(B, N, 3, generator=g) if kind == "random" else torch.sin(3.0 * p) + 0.1
x = torch.cat([p, torch.nn.functional.normalize(n, dim=-1)], -1).transpose(1, 2).contiguous().to(dev)
lib = _lib.lib()


def run(ws):
    idx = torch.empty(B, N, k, dtype=torch.int64, device=dev)
    val = torch.empty(B, N, k, dtype=torch.float32, device=dev)
    xx = torch.empty(B, N, dtype=torch.float32, device=dev)
    _lib.call("gcn_knn_model", _lib.ptr(x), B, 6, N, k, k, 1, _lib.ptr(idx), _lib.ptr(val), _lib.ptr(xx), _lib.ptr(ws),
              _lib.stream_of(x))
    return idx, val


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


assert lib.gcn_knn_normal_supported(B, N, k), "shape not served by the filter path"
ws = torch.empty(lib.gcn_knn_tiles_ws_bytes(B, 6, N), dtype=torch.uint8, device=dev)
ms_old, (i0, v0) = timed(lambda: run(None))
ms_new, (i1, v1) = timed(lambda: run(ws))
same = torch.equal(i0, i1) and torch.equal(v0, v1)
n = B * N
off = ((n * 32 + 255) // 256 * 256) + ((n * 4 + 255) // 256 * 256)
flag = ws[off:off + n]
bm = ws[off + (n + 255) // 256 * 256:][: n * (N // 8)]
bits = sum(int(((bm >> s) & 1).sum()) for s in range(8))
print("N=%d B=%d k=%d %s: exhaustive %.3f ms, filter path %.3f ms (%.2fx), identical=%s, fallback queries %d, "
      "candidates/query %.1f" % (N, B, k, kind, ms_old, ms_new, ms_old / ms_new, same, int(flag.sum()), bits / n))
sys.exit(0 if same else 1)
