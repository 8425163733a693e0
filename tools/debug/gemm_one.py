"""One head-GEMM shape in a loop, for rocprofv3 --pmc runs: python tools/debug/gemm_one.py N K [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gcanet_amd import _lib  # noqa: E402

N, K = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
M = 65536
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(iters):
    _lib.call("gcn_gemm_bf16", _lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), 0, M, N, N, K, None, None, 0, 0, _lib.stream_of(A))
torch.cuda.synchronize()
