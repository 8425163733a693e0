"""Isolated EdgeConv forward launches (for rocprofv3 / PMC runs): python tools/ec_bench.py [C] [Cout] [iters] [k]
Prints the centre-term kernel and the grouped kernel separately; TFLOP/s are ALGORITHMIC (2*B*N*k*2C*Cout)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib, dgcnn  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
k = int(sys.argv[4]) if len(sys.argv) > 4 else 64
dev = torch.device("cuda:0")
B, N = 8, 8192
g = torch.Generator().manual_seed(0)
xyz = torch.rand(B, 3, N, generator=g).to(dev)
idx = dgcnn.knn(xyz, k, k)
x = torch.randn(B, N, C, generator=g).to(dev)
w = (torch.randn(Cout, 2 * C, generator=g) / (2 * C) ** 0.5).to(dev)
Cp = _lib.lib().gcn_edgeconv_padded_channels(C)
x_bf = torch.empty(B, N, Cp, dtype=torch.bfloat16, device=dev)
wp = torch.empty(Cout, 2 * Cp, dtype=torch.bfloat16, device=dev)
q = torch.empty(B * N, Cout, device=dev)
st = _lib.stream_of(x)
_lib.call("gcn_cast_pad_bf16", _lib.ptr(x), B * N, C, _lib.ptr(x_bf), st)
_lib.call("gcn_edgeconv_pack_w", _lib.ptr(w), Cout, C, _lib.ptr(wp), st)
ymax = torch.empty(B, N, Cout, device=dev)
ymin = torch.empty(B, N, Cout, device=dev)
amax = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
amin = torch.empty(B, N, Cout, dtype=torch.uint8, device=dev)
gsum = torch.empty(B, 2, 2, dtype=torch.float64, device=dev)
ga = torch.randn(Cout, generator=g).to(dev)
flops = 2.0 * B * N * k * 2 * C * Cout


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


ms_c = timed(lambda: _lib.call("gcn_edgeconv_center", _lib.ptr(x_bf), _lib.ptr(wp), B * N, C, Cout, _lib.ptr(q), st))
print("edgeconv_center C=%d Cout=%d: %.4f ms" % (C, Cout, ms_c))
for routed in (True, False):
    for with_arg in (True, False):
        args = (_lib.ptr(x_bf), _lib.ptr(wp), _lib.ptr(idx), 1, B, N, N, C, k, Cout, 2, _lib.ptr(q), _lib.ptr(ymax),
                None if routed else _lib.ptr(ymin), _lib.ptr(amax) if with_arg else None,
                _lib.ptr(amin) if (with_arg and not routed) else None, _lib.ptr(gsum), _lib.ptr(ga) if routed else None, st)
        ms = timed(lambda: _lib.call("gcn_edgeconv_fwd", *args))
        print("edgeconv_fwd %s C=%d Cout=%d k=%d arg=%d: %.4f ms (+centre %.4f)  %.1f TFLOP/s algorithmic = %.1f%% of 2.5 PF "
              "(with centre kernel: %.1f%%)" % ("ROUTED" if routed else "full  ", C, Cout, k, with_arg, ms, ms_c,
                                                 flops / ms / 1e9, flops / ms / 1e9 / 25, flops / (ms + ms_c) / 1e9 / 25))
