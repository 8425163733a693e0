"""Head GEMM shapes of the bench step (M = 65536 rows): csrc/gemm.hip vs torch (hipBLASLt) on the same bf16 operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
M = 65536


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N, K in [(512, 1280), (512, 256), (256, 512), (256, 256), (256, 832), (64, 256), (128, 272), (1024, 256), (32, 256)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev)
    bb = b.bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    gsum = torch.empty(8, 4, 2, dtype=torch.float64, device=dev)
    st = _lib.stream_of(A)
    G = 4 if (N // 4) % 32 == 0 else 0
    ws = torch.empty(_lib.lib().gcn_gemm_stats_ws_bytes(M, N), dtype=torch.uint8, device=dev)
    t_mine = timed(lambda: _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), 0, M, N, N, K,
                                     _lib.ptr(gsum) if G else None, _lib.ptr(ws) if G else None, 8192, G, st))
    t_plain = timed(lambda: _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), 0, M, N, N, K,
                                      None, None, 0, 0, st))
    t_lib = timed(lambda: torch.nn.functional.linear(A, W, bb))
    dY = torch.randn(M, N, device=dev).bfloat16()
    raw = torch.empty(N * K + N, device=dev)
    wws = torch.empty(_lib.lib().gcn_gemm_wgrad_ws_bytes(M, N, K), dtype=torch.uint8, device=dev)
    t_wg = timed(lambda: _lib.call("gcn_gemm_wgrad_bf16", _lib.ptr(dY), _lib.ptr(A), M, N, K, _lib.ptr(raw), _lib.ptr(raw[N * K:]),
                                   _lib.ptr(wws), st))
    t_wl = timed(lambda: (torch.bmm(dY.view(32, M // 32, N).transpose(1, 2), A.view(32, M // 32, K)).sum(0),
                          dY.sum(0, dtype=torch.float32)))
    fl = 2.0 * M * N * K
    print("N=%4d K=%4d  fwd mine %6.1f us (%5.0f TF)%s  plain %6.1f us  lib %6.1f us | wgrad+dbias mine %6.1f us  lib(split-K bmm+sum, sum) %6.1f us" % (
        N, K, t_mine, fl / t_mine / 1e6, " +GN stats" if G else "", t_plain, t_lib, t_wg, t_wl))
