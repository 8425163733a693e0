#!/usr/bin/env python3
"""Stage-by-stage run of the device grouping path with a synchronise + progress line after every entry point."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib
from gcanet_amd.grouping import _pad16

def say(*a):
    torch.cuda.synchronize(); print(*a, flush=True)

dev = torch.device("cuda")
rng = np.random.default_rng(0)
B, N, P = 2, 600, 3
centers = rng.random((B, 6, 3)); which = rng.integers(0, 6, (B, N))
xyz = centers[np.arange(B)[:, None], which] + 0.004 * rng.standard_normal((B, N, 3))
n, S = B * N, B * P
labels = torch.from_numpy((which % P).reshape(-1)).to(dev)
seg_key = torch.arange(B, device=dev).repeat_interleave(N) * P + labels
seg_sorted, order = torch.sort(seg_key, stable=True)
counts = torch.bincount(seg_sorted, minlength=S)
seg_offsets = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).int()
seg_cls = (torch.arange(S, device=dev) % P).int()
seg_of = seg_sorted.int()
shifted = torch.from_numpy(xyz.reshape(-1, 3).astype(np.float32)).to(dev)[order].contiguous()
fi = _pad16(torch.from_numpy(np.eye(16, dtype=np.float32)[which.reshape(-1) % 16]).to(dev)[order])
fp = _pad16(torch.randn(n, 22, device=dev) * 0.01)
point_index = (order % N).int()
lib = _lib.lib(); st = _lib.stream_of(shifted)
xx = torch.empty(n, device=dev); tiles = torch.empty(S + 1, dtype=torch.int32, device=dev)
dm = torch.empty(2, S, device=dev)
for f, d in ((fi, dm[0]), (fp, dm[1])):
    _lib.call("gcn_segment_diameter2", n, f.shape[1], _lib.ptr(f), _lib.ptr(seg_offsets), _lib.ptr(seg_cls), S, _lib.ptr(xx), _lib.ptr(tiles), _lib.ptr(d), st)
say("diameter ok", dm)
grid_ws = torch.empty(lib.gcn_ballquery_sim_ws_bytes(n), dtype=torch.uint8, device=dev)
ws = torch.empty(lib.gcn_cluster_components_ws_bytes(n), dtype=torch.uint8, device=dev)
print("ws bytes", grid_ws.numel(), ws.numel(), flush=True)
start_len = torch.empty(n, 2, dtype=torch.int32, device=dev)
status = torch.zeros(8, dtype=torch.int32, device=dev)
cap = n * 50
nbr = torch.empty(cap, dtype=torch.int32, device=dev)
_lib.call("gcn_ballquery_sim", n, 0.03, _lib.ptr(shifted), _lib.ptr(seg_of), _lib.ptr(seg_offsets), _lib.ptr(seg_cls), S, _lib.ptr(fi), fi.shape[1], _lib.ptr(dm[0]), 0.9,
          _lib.ptr(fp), fp.shape[1], _lib.ptr(dm[1]), 0.0, _lib.ptr(nbr), cap, _lib.ptr(start_len), _lib.ptr(status), _lib.ptr(grid_ws), st)
say("ballquery ok", status, int(start_len[:, 1].sum()), int(start_len[:, 1].max()))
ci = torch.empty(n, 2, dtype=torch.int32, device=dev); co = torch.empty(n + 1, dtype=torch.int32, device=dev)
_lib.call("gcn_cluster_components", n, _lib.ptr(nbr), _lib.ptr(start_len), _lib.ptr(seg_of), _lib.ptr(seg_offsets), _lib.ptr(seg_cls), S, _lib.ptr(point_index), _lib.ptr(ws),
          _lib.ptr(ci), _lib.ptr(co), _lib.ptr(status[4:]), st)
say("cluster ok", status)
