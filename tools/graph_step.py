"""Capture the bench step into a HIP graph (torch.cuda.CUDAGraph) and compare replay with eager launches.
   python tools/graph_step.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402
from gcanet_amd.layers import CastCache, ZeroArena  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 1)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True, capturable=True)
arena = ZeroArena(dev)
casts = CastCache(model, pad_k={model.conv3.weight: (model.conv3.weight.shape[1] + 15) // 16 * 16})
pts, nrm = bench.synth_clouds(range(8), 8192, dev)


def step():
    dp.zero_grad()
    arena.begin_step()
    casts.refresh()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(pts, nrm)
    loss = bench.loss_of(out)
    loss.backward()
    dp.all_reduce_grads()
    opt.step()
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager: %.3f ms/step (host enqueue %.3f)" % timed(step, steps))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
torch.cuda.synchronize()
print("captured; loss tensor", float(loss))
print("graph replay: %.3f ms/step (host %.3f)" % timed(g.replay, steps))
l0 = float(loss)
g.replay()
torch.cuda.synchronize()
print("loss after replays: %.6f -> %.6f (training continues inside the graph)" % (l0, float(loss)))
