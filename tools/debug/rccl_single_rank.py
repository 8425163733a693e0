"""RCCL smoke on the ONE GPU of a test box: a 1-rank `nccl` process group, FlatGradDP told it has two replicas (so every
collective of the multi-rank step is really issued: broadcasts, the early all-reduce started from the backward hook, the
late one, the wait), bench.py's eager step around it.  Checks API / stream semantics only (a 1-rank all-reduce moves no
data); prints ms per step and how often the early all-reduce started inside backward.
   python tools/debug/rccl_single_rank.py [steps]"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402
from gcanet_amd.layers import CastCache, ZeroArena  # noqa: E402
from gcanet_amd.optim import FlatAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 2, late=model.encoder.parameters())     # "two replicas": collectives are issued
dp.sync_params()
opt = FlatAdam(dp, lr=1e-3)
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


for _ in range(3):
    step()
torch.cuda.synchronize()
dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dist.barrier()
dt = (time.perf_counter() - t0) / steps
print("rccl single-rank: %.3f ms/step, loss %.5f (finite: %s), early all-reduce started inside backward in %d of %d steps"
      % (dt * 1e3, float(loss), bool(torch.isfinite(loss)), dp.early_started_in_backward, steps + 3))
dist.destroy_process_group()
