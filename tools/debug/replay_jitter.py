"""Per-replay duration of the captured step (HIP events around every replay): is the default 10-step window noisy?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import dgcnn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
pts, nrm = bench.synth_clouds(range(8), 8192, dev)
st = bench.make_step(model, pts, nrm, 1)
g, _loss = bench.capture_step(st["step"], 3)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
torch.cuda.synchronize()
evs[0].record()
for i in range(60):
    g.replay()
    evs[i + 1].record()
torch.cuda.synchronize()
ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(60)]
print(" ".join("%.2f" % t for t in ts))
print("mean first 10: %.3f  mean 10-20: %.3f  mean last 30: %.3f  min %.3f max %.3f" % (sum(ts[:10]) / 10, sum(ts[10:20]) / 10,
                                                                                       sum(ts[30:]) / 30, min(ts), max(ts)))
