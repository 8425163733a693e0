"""Kernel list of ONE forward pass (autocast bf16) of the hot-path module, in launch order with durations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
pts, nrm = bench.synth_clouds(range(8), 8192, dev)
for _ in range(3):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(pts, nrm)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(pts, nrm)
    torch.cuda.synchronize()
rows = [(e.time_range.start, e.time_range.elapsed_us(), e.name) for e in prof.events()
        if e.device_type == torch.autograd.DeviceType.CUDA]
rows.sort()
print("kernels: %d, sum %.3f ms" % (len(rows), sum(r[1] for r in rows) / 1e3))
for _, d, n in rows:
    print("%8.1f us  %s" % (d, n[:110]))
