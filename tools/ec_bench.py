"""Isolated EdgeConv forward launches (for rocprofv3 --pmc runs): python tools/ec_bench.py [C] [Cout] [iters]"""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib, dgcnn  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
B, N, k = 8, 8192, 64
g = torch.Generator().manual_seed(0)
xyz = torch.rand(B, 3, N, generator=g).to(dev)
idx = dgcnn.knn(xyz, k, k)
x = torch.randn(B, N, C, generator=g).to(dev)
w = (torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5).to(dev)
Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
x_bf = torch.empty(B, N, Cp, dtype=torch.bfloat16, device=dev)
wp = torch.empty(Cout, 2 * Cp, dtype=torch.bfloat16, device=dev)
st = _lib.stream_of(x)
_lib.call("gcn_cast_pad_bf16", _lib.ptr(x), B * N, C, _lib.ptr(x_bf), st)
_lib.call("gcn_edgeconv_pack_w", _lib.ptr(w), Cout, C, _lib.ptr(wp), st)
ymax = torch.empty(B, N, Cout, device=dev)
ymin = torch.empty(B, N, Cout, device=dev)
amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
amin = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
gsum = torch.empty(B, 2, 2, dtype=torch.float64, device=dev)
ga = torch.randn(Cout, generator=g).to(dev)
for routed in (False, True):
  for with_arg in (True, False):
    if routed:
        args = (_lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), 1, B, N, N, C, k, Cout, 2, _lib.ptr(ymax), None,
                _lib.ptr(amax) if with_arg else None, None, _lib.ptr(gsum), _lib.ptr(ga), st)
        for _ in range(2):
            _lib.call("gcn_edgeconv_fwd", *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            _lib.call("gcn_edgeconv_fwd", *args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print("edgeconv_fwd ROUTED C=%d Cout=%d arg=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (
            C, Cout, with_arg, ms, 2.0 * B * N * k * 2 * C * Cout / ms / 1e9, 2.0 * B * N * k * 2 * C * Cout / ms / 1e9 / 25))
for with_arg in (True, False):
    args = (_lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), 1, B, N, N, C, k, Cout, 2, _lib.ptr(ymax), _lib.ptr(ymin),
            _lib.ptr(amax) if with_arg else None, _lib.ptr(amin) if with_arg else None, _lib.ptr(gsum), None, st)
    for _ in range(2):
        _lib.call("gcn_edgeconv_fwd", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        _lib.call("gcn_edgeconv_fwd", *args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("edgeconv_fwd C=%d Cout=%d arg=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2.5 PF)" % (
        C, Cout, with_arg, ms, 2.0 * B * N * k * 2 * C * Cout / ms / 1e9, 2.0 * B * N * k * 2 * C * Cout / ms / 1e9 / 25))
