"""Forward / backward / optimizer split of the bench step, and forward time per sub-module (CUDA events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gcanet_amd import dgcnn, parallel  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
dp = parallel.FlatGradDP(model, 1)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
pts, nrm = bench.synth_clouds(range(8), 8192, dev)
ev = lambda: torch.cuda.Event(enable_timing=True)


def step(rec=None):
    e = [ev() for _ in range(5)]
    e[0].record()
    dp.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(pts, nrm)
    loss = bench.loss_of(out)
    e[1].record()
    loss.backward()
    e[2].record()
    dp.all_reduce_grads()
    e[3].record()
    opt.step()
    e[4].record()
    torch.cuda.synchronize()
    return [e[i].elapsed_time(e[i + 1]) for i in range(4)]


for _ in range(3):
    step()
ts = [step() for _ in range(5)]
avg = [sum(t[i] for t in ts) / len(ts) for i in range(4)]
print("zero_grad+forward+loss %.2f ms | backward %.2f ms | all_reduce %.2f ms | adam %.2f ms | total %.2f" %
      (avg[0], avg[1], avg[2], avg[3], sum(avg)))

# forward per child module via hooks
times = {}
starts = {}


def pre(name):
    def f(m, i):
        s = ev(); s.record(); starts[name] = s
    return f


def post(name):
    def f(m, i, o):
        e = ev(); e.record(); times.setdefault(name, []).append((starts[name], e))
    return f


for name, m in model.named_children():
    m.register_forward_pre_hook(pre(name))
    m.register_forward_hook(post(name))
with torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        model(pts, nrm)
torch.cuda.synchronize()
for name, lst in times.items():
    print("  fwd %-24s %.3f ms x %d" % (name, sum(a.elapsed_time(b) for a, b in lst) / 3, len(lst) // 3))
