"""gcn_wgrad_narrow at the step's shapes vs the library route (split-K bmm + sum + column sum): python tools/wgrad_narrow_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib  # noqa: E402
from gcanet_amd.layers import tall_skinny_tn  # noqa: E402

dev = torch.device("cuda:0")
M = 65536
for N, K, bf in ((10, 256, 1), (22, 256, 1), (3, 256, 1), (30, 30, 1), (3, 128, 0), (32, 512, 1)):
    dt = torch.bfloat16 if bf else torch.float32
    dY = torch.randn(M, N, device=dev).to(dt)
    X = torch.randn(M, K, device=dev).to(dt)
    raw = torch.empty(N * K + N, device=dev)
    ws = torch.empty(_lib.lib().gcn_wgrad_narrow_ws_bytes(M, N, K), dtype=torch.uint8, device=dev)
    st = _lib.stream_of(X)

    def own():
        _lib.call("gcn_wgrad_narrow", _lib.ptr(dY), bf, _lib.ptr(X), bf, M, N, K, _lib.ptr(raw), _lib.ptr(raw[N * K:]), _lib.ptr(ws), st)

    def lib():
        tall_skinny_tn(dY, X, out_dtype=torch.float32)
        dY.sum(0, dtype=torch.float32)

    res = []
    for fn in (own, lib):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("N=%2d K=%4d %s: own %.1f us, lib %.1f us" % (N, K, "bf16" if bf else "f32", res[0], res[1]))
