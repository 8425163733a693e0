#!/usr/bin/env python3
"""gcn_sparse_gather_gemm / gcn_sparse_wgrad on synthetic rule tables: what bounds the kernels?
  fill 1.0, own row   -> MFMA-bound rate (4 full tiles per offset)
  fill 0.1, random    -> the tiny U-Net's regime
  fill 0.1, own row   -> same MFMA work without the random row gather"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import _lib

def run(M, K, C, fill, rand, reps=10):
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, C, device=dev, generator=g)
    W = torch.randn(K, C, C, device=dev, generator=g)
    src = torch.randint(0, M, (M, K), device=dev, generator=g) if rand else torch.arange(M, device=dev).view(M, 1).expand(M, K)
    keep = torch.rand(M, K, device=dev, generator=g) < fill
    rule = torch.where(keep, src, torch.full_like(src, -1)).int().contiguous()
    ruleT = rule.t().contiguous()
    out = torch.empty(M, C, device=dev); dW = torch.empty_like(W)
    ws = torch.empty(max(_lib.lib().gcn_sparse_gather_gemm_ws_floats(M, K, C), 1), device=dev)
    st = _lib.stream_of(x)
    pairs = int(keep.sum())
    res = []
    for name, fn in (("gather_gemm", lambda: _lib.call("gcn_sparse_gather_gemm", M, K, C, C, _lib.ptr(x), _lib.ptr(rule), _lib.ptr(W), 0, 0, _lib.ptr(out), _lib.ptr(ws), st)),
                     ("wgrad", lambda: _lib.call("gcn_sparse_wgrad", M, K, C, C, _lib.ptr(x), _lib.ptr(ruleT), _lib.ptr(out), _lib.ptr(dW), st))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append("%s %.3f ms (%.1f TF on pairs)" % (name, ms, 2.0 * pairs * C * C / ms / 1e9))
    print("M=%d K=%d C=%d fill=%.2f %s: %s" % (M, K, C, fill, "random" if rand else "own-row", " | ".join(res)), flush=True)

if __name__ == "__main__":
    for fill, rand in ((1.0, False), (1.0, True), (0.3, True), (0.1, True), (0.1, False)):
        run(200000, 27, 64, fill, rand)
    run(50000, 27, 128, 0.1, True)
