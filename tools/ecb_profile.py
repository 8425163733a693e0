"""Kernel list of ONE EdgeConv backward at an encoder-layer shape: python tools/ecb_profile.py [C] [Cout]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcanet_amd import dgcnn  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B, N, k = 8, 8192, 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
# features that live near a 3-D manifold (like real layer activations): a geometric kNN graph without the extreme
# hubs of i.i.d. 64-D gaussians
xyz = torch.rand(B, N, 3, device=dev)
x = (torch.tanh(xyz @ torch.randn(3, C, device=dev)) + 0.05 * torch.randn(B, N, C, device=dev)).requires_grad_(True)
xc = x.detach().transpose(1, 2).contiguous()
idx = dgcnn.knn(xc, k, k)
w = (torch.randn(Cout, 2 * C, device=dev) * 0.1).requires_grad_(True)
gn = torch.nn.GroupNorm(2, Cout).to(dev)
dout = torch.randn(B, N, Cout, device=dev)


def run():
    out, _ = dgcnn.edge_conv_pm(x, idx, w, gn, "bf16", want_cm=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out.backward(dout)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    run()
print("backward: %.3f ms (events)" % min(run() for _ in range(5)))
out, _ = dgcnn.edge_conv_pm(x, idx, w, gn, "bf16", want_cm=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    out.backward(dout)
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        rows.append((e.time_range.start, e.time_range.elapsed_us(), e.name))
rows.sort()
print("kernels: %d, sum %.3f ms" % (len(rows), sum(r[1] for r in rows) / 1e3))
for _, d, n in rows:
    print("%8.1f us  %s" % (d, n[:120]))
