"""Attribute the bench step's GPU time to aten / autograd ops with torch.profiler:
   python tools/op_profile.py [steps]  ->  table sorted by self device time (per step)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 1)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
from gcanet_amd.layers import CastCache  # noqa: E402
casts = CastCache(model)
pts, nrm = bench.synth_clouds(range(8), 8192, dev)


def step():
    dp.zero_grad()
    casts.refresh()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(pts, nrm)
    loss = bench.loss_of(out)
    loss.backward()
    dp.all_reduce_grads()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=False)
rows = []
for e in ka:
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / steps / 1e3, e.count / steps, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total self device time per step: %.3f ms" % tot)
for t, c, k in rows[:70]:
    print("%8.3f ms  n=%6.1f  %s" % (t, c, k[:100]))

print("\nautograd nodes (device time incl. nested ops, per step):")
rows = []
for e in ka:
    if "Backward" in e.key or e.key.startswith("autograd::engine") or "Function" in e.key:
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = e.cuda_time_total
        rows.append((dt / steps / 1e3, e.count / steps, e.key))
rows.sort(reverse=True)
for t, c, k in rows[:40]:
    print("%8.3f ms  n=%6.1f  %s" % (t, c, k[:90]))
