"""Exhaustive vs filtered segment diameter (csrc/softgroup.hip: seg_diameter_kernel vs csrc/segdiam.hip) at the literal
forward_train shapes: 8 clouds x 8192 points, blob features, one or ten classes per cloud."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib  # noqa: E402


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


def main():
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    for name, sizes, C in (("8 x 8192, C=64", [8192] * 8, 64), ("80 x ~820, C=64", [820] * 80, 64),
                           ("8 x 8192, C=32", [8192] * 8, 32), ("1 x 65536, C=64", [65536], 64),
                           ("8 x 8192 uniform, C=64", [8192] * 8, 64), ("8 x 8192 tight, C=64", [8192] * 8, 64),
                           ("16 x 4096, C=64", [4096] * 16, 64), ("40 x 1640, C=64", [1640] * 40, 64)):
        parts = []
        for m in sizes:
            if "uniform" in name:
                parts.append(rng.standard_normal((m, C)))
            else:
                cen = rng.standard_normal((6, C)) * 2.0
                parts.append(cen[rng.integers(0, 6, m)] + (0.002 if "tight" in name else 0.1) * rng.standard_normal((m, C)))
        f = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(dev)
        n, S = f.shape[0], len(sizes)
        offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
        cls = torch.zeros(S, dtype=torch.int32, device=dev)
        xx, tiles = torch.empty(n, device=dev), torch.empty(S + 1, dtype=torch.int32, device=dev)
        ref, got = torch.empty(S, device=dev), torch.empty(S, device=dev)
        ws = torch.empty(_lib.lib().gcn_segment_diameter2_ws_bytes(n, C, S), dtype=torch.uint8, device=dev)
        st = _lib.stream_of(f)
        t0 = timed(lambda: _lib.call("gcn_segment_diameter2", n, C, _lib.ptr(f), _lib.ptr(offs), _lib.ptr(cls), S, _lib.ptr(xx),
                                     _lib.ptr(tiles), _lib.ptr(ref), st))
        t1 = timed(lambda: _lib.call("gcn_segment_diameter2_filtered", n, C, _lib.ptr(f), _lib.ptr(offs), _lib.ptr(cls), S,
                                     _lib.ptr(ws), _lib.ptr(got), st))
        ncand = int(ws[((4 * S + 255) // 256) * 256:][:4].view(torch.int32)[0])
        pairs = 4 * sum(((m + 63) // 64) * ((m + 63) // 64 + 1) // 2 for m in sizes)
        print("%-26s exhaustive %.3f ms   filtered %.3f ms   32x32 blocks rechecked %d of %d   identical %s"
              % (name, t0, t1, ncand, pairs, torch.equal(ref.view(torch.int32), got.view(torch.int32))), flush=True)


if __name__ == "__main__":
    main()
