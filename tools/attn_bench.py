"""Attention kernels: accuracy of the bf16 matrix-core path vs the exact f32 kernel / torch, and timings.
   python tools/attn_bench.py [BH] [L] [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import attention  # noqa: E402

BH = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(BH, L, D, generator=g).to(dev) for _ in range(3))
scale = D ** -0.5


def timeit(fn, iters=5):
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


o_bf = attention.sdpa(q, k, v, None, scale, "bf16")
if L <= 8192:
    ref = torch.softmax(torch.bmm(q, k.transpose(1, 2)) * scale, -1) @ v
    o_f32 = attention.sdpa(q, k, v, None, scale, "f32")
    print("f32 kernel vs torch: max abs %.3e" % (o_f32 - ref).abs().max().item())
    print("bf16 kernel vs torch: max abs %.3e  (ref max %.3f)" % ((o_bf - ref).abs().max().item(), ref.abs().max().item()))
    t = timeit(lambda: attention.sdpa(q, k, v, None, scale, "f32"), 3)
    print("f32  fwd: %.3f ms  %.1f TFLOP/s" % (t, 4.0 * BH * L * L * D / t / 1e9))
t = timeit(lambda: attention.sdpa(q, k, v, None, scale, "bf16"))
print("bf16 fwd: %.3f ms  %.1f TFLOP/s (incl. operand packing)" % (t, 4.0 * BH * L * L * D / t / 1e9))

# backward: bf16 flash kernels vs torch autograd of the f32 reference
qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
do = torch.randn(BH, L, D, generator=g).to(dev)
o = attention.sdpa(qq, kk, vv, None, scale, "bf16")
o.backward(do)
if L <= 8192 and BH * L * L * 4 < 8e9:
    q2, k2, v2 = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = torch.softmax(torch.bmm(q2, k2.transpose(1, 2)) * scale, -1) @ v2
    ref.backward(do)
    for n, a, b in (("dq", qq.grad, q2.grad), ("dk", kk.grad, k2.grad), ("dv", vv.grad, v2.grad)):
        print("%s: max abs err %.3e  (ref max %.3e, rel fro %.3e)" % (n, (a - b).abs().max().item(), b.abs().max().item(),
                                                                 ((a - b).norm() / b.norm()).item()))


def fb():
    o = attention.sdpa(qq, kk, vv, None, scale, "bf16")
    o.backward(do)


t = timeit(fb, 3)
print("bf16 fwd+bwd: %.3f ms  %.1f TFLOP/s (4+14 = 18 B H L^2 D incl. packing)" % (t, 18.0 * BH * L * L * D / t / 1e9))

# IEEE-half variant (the type BASELINE config 5 names)
o_h = attention.sdpa(q, k, v, None, scale, "fp16")
print("fp16 vs bf16 kernel: max abs %.3e" % (o_h - o_bf).abs().max().item())
t = timeit(lambda: attention.sdpa(q, k, v, None, scale, "fp16"))
print("fp16 fwd: %.3f ms  %.1f TFLOP/s (incl. operand packing)" % (t, 4.0 * BH * L * L * D / t / 1e9))


def fbh():
    o = attention.sdpa(qq, kk, vv, None, scale, "fp16")
    o.backward(do)


t = timeit(fbh, 3)
print("fp16 fwd+bwd: %.3f ms  %.1f TFLOP/s" % (t, 18.0 * BH * L * L * D / t / 1e9))
