"""GroupNorm(+ReLU) forward / backward timings at the head shapes (GPU box): python tools/gn_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd.layers import GroupNormReLUFunction  # noqa: E402

dev = torch.device("cuda:0")
for C, G in ((512, 8), (256, 4), (128, 4)):
    x = torch.randn(8, 8192, C, device=dev).bfloat16().requires_grad_(True)
    ga = torch.ones(C, device=dev, requires_grad=True)
    be = torch.zeros(C, device=dev, requires_grad=True)
    dy = torch.randn(8, 8192, C, device=dev).bfloat16()
    for _ in range(3):
        y = GroupNormReLUFunction.apply(x, ga, be, G, 1e-5, True)
        y.backward(dy)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(20):
        ev[0].record()
        y = GroupNormReLUFunction.apply(x, ga, be, G, 1e-5, True)
        ev[1].record()
        y.backward(dy)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    mb = 8 * 8192 * C * 2 / 1e6
    print("C=%4d  fwd %.1f us (%.2f TB/s of 3 passes)  bwd %.1f us (%.2f TB/s of 5 passes)"
          % (C, tf / 20 * 1e3, 3 * mb / (tf / 20 * 1e3), tb / 20 * 1e3, 5 * mb / (tb / 20 * 1e3)))
