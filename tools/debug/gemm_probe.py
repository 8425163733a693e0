import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import _lib
dev = torch.device("cuda:0")
M = 65536
def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N, K in [(512, 256), (256, 256), (128, 256), (512, 64), (512, 1024)]:
    A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev); outf = torch.empty(M, N, device=dev)
    st = _lib.stream_of(A)
    t1 = timed(lambda: _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), 0, M, N, N, K, None, None, 0, 0, st))
    t2 = timed(lambda: _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), 0, M, N, N, K, None, None, 0, 0, st))
    t3 = timed(lambda: _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), None, _lib.ptr(outf), 1, M, N, N, K, None, None, 0, 0, st))
    t4 = timed(lambda: torch.nn.functional.linear(A, W))
    print("N=%d K=%d: bias %.1f  nobias %.1f  nobias f32out %.1f  lib(nobias) %.1f us" % (N, K, t1, t2, t3, t4))
