"""Isolated feature-space kNN launches (for rocprofv3 --pmc): python tools/knn_bench.py [C] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import dgcnn  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(8, C, 8192, generator=g).to(dev)
for _ in range(2):
    dgcnn.knn(x, 64, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    dgcnn.knn(x, 64, 64)
e1.record()
torch.cuda.synchronize()
print("knn C=%d: %.3f ms" % (C, e0.elapsed_time(e1) / iters))
