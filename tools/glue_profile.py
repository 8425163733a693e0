"""Which Python lines launch the small torch kernels of the bench step (copies, fills, adds, reductions)?
   python tools/glue_profile.py [steps]  ->  aten op x innermost gcanet_amd/bench frame, sorted by launches per step"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402
from gcanet_amd.layers import CastCache  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 1, late=model.encoder.parameters())
opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
casts = CastCache(model, pad_k={model.conv3.weight: (model.conv3.weight.shape[1] + 15) // 16 * 16})
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

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(steps):
        step()
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, "self_device_time_total", 0) or 0
    if dt <= 0 or not e.name.startswith("aten::"):
        continue
    # python stacks are not recorded on this build: attribute to the chain of enclosing ops instead (autograd node /
    # custom Function at the outside, the nearest non-trivial aten parents inside)
    chain = []
    par = e.cpu_parent
    while par is not None:
        chain.append(par.name)
        par = par.cpu_parent
    chain = [c for c in chain if not c.startswith("ProfilerStep")]
    where = " < ".join(chain[:3] + (["..", chain[-1]] if len(chain) > 3 else [])) or "(top level)"
    a = agg[(e.name, where)]
    a[0] += 1
    a[1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("launching aten ops per step: %.1f, device time %.3f ms" % (sum(v[0] for _, v in rows) / steps,
                                                                sum(v[1] for _, v in rows) / steps / 1e3))
for (name, where), (n, dt) in rows[:90]:
    print("%6.1f x %8.1f us  %-28s %s" % (n / steps, dt / steps, name, where[:110]))
