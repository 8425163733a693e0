"""cProfile of the eager bench step: where does the host spend its ~8 ms per step?  python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402
from gcanet_amd.layers import CastCache, ZeroArena  # noqa: E402
from gcanet_amd.optim import FlatAdam  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 1)
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


for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
