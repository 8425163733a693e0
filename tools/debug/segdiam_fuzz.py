"""Random configurations of csrc/segdiam.hip against the exhaustive kernel (same bits expected): widths, segment sizes
around the routing limits, offsets far from the origin, cluster tightness over six decades, outliers, huge and tiny scales."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for it in range(cases):
    C = int(rng.choice([16, 32, 48, 64, 96, 128]))
    S = int(rng.integers(1, 6))
    sizes = [int(rng.choice([rng.integers(1, 300), rng.integers(2000, 2200), rng.integers(2049, 7000)])) for _ in range(S)]
    cls = [(-1 if rng.random() < 0.15 else i) for i in range(S)]
    parts = []
    for m in sizes:
        nb = int(rng.integers(1, 7))
        cen = rng.standard_normal((nb, C)) * 10.0 ** rng.uniform(-2, 2)
        tight = 10.0 ** rng.uniform(-6, 0.5)
        f = cen[rng.integers(0, nb, m)] + tight * rng.standard_normal((m, C))
        if rng.random() < 0.3:
            f = f + 10.0 ** rng.uniform(0, 3)                 # far from the origin
        if rng.random() < 0.2:
            f[rng.integers(0, m)] += 10.0 ** rng.uniform(0, 2)   # one outlier
        if rng.random() < 0.1:
            f = f * 10.0 ** rng.uniform(-8, 6)
        parts.append(f)
    f = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(dev)
    n = f.shape[0]
    offs = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    clst = torch.tensor(cls, dtype=torch.int32, device=dev)
    xx, tiles = torch.empty(n, device=dev), torch.empty(S + 1, dtype=torch.int32, device=dev)
    ref, got = torch.empty(S, device=dev), torch.empty(S, device=dev)
    st = _lib.stream_of(f)
    _lib.call("gcn_segment_diameter2", n, C, _lib.ptr(f), _lib.ptr(offs), _lib.ptr(clst), S, _lib.ptr(xx), _lib.ptr(tiles), _lib.ptr(ref), st)
    ws = torch.empty(_lib.lib().gcn_segment_diameter2_ws_bytes(n, C, S), dtype=torch.uint8, device=dev)
    _lib.call("gcn_segment_diameter2_filtered", n, C, _lib.ptr(f), _lib.ptr(offs), _lib.ptr(clst), S, _lib.ptr(ws), _lib.ptr(got), st)
    if not torch.equal(ref.view(torch.int32), got.view(torch.int32)):
        bad += 1
        print("MISMATCH case", it, "C", C, "sizes", sizes, "cls", cls, ref.tolist(), got.tolist(), flush=True)
print("cases %d, mismatches %d" % (cases, bad))
