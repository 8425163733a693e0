"""Host timeline of the literal full step by phase (perf_counter marks; the GPU runs asynchronously, so a phase that
ends in a host synchronisation shows the device's backlog as well)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import gcanet, grouping  # noqa: E402

marks = defaultdict(float)
last = [0.0]


def mark(name):
    t = time.perf_counter()
    marks[name] += t - last[0]
    last[0] = t


def wrap(mod, fn, name):
    orig = getattr(mod, fn)

    def f(*a, **k):
        mark("before " + name)
        r = orig(*a, **k)
        mark(name)
        return r
    setattr(mod, fn, f)


wrap(gcanet, "forward_grouping_device", "forward_grouping (ends in the status read-back)")
wrap(gcanet, "clusters_voxelization", "clusters_voxelization")
orig_head = gcanet.InstanceHead.forward


def head(self, *a, **k):
    mark("before instance head")
    r = orig_head(self, *a, **k)
    mark("instance head forward (one size read-back)")
    return r


gcanet.InstanceHead.forward = head
orig_bw = torch.Tensor.backward


def bw(self, *a, **k):
    mark("losses")
    r = orig_bw(self, *a, **k)
    mark("backward")
    return r


torch.Tensor.backward = bw
args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.full_workload(args, dev, steps=2, warmup=3)
marks.clear()
last[0] = time.perf_counter()
t0 = time.perf_counter()
steps = 10
out = bench.full_workload(args, dev, steps=steps, warmup=0)
print(out["ms_per_step"], out["proposals"])
tot = 0.0
for k, v in marks.items():
    print("%-55s %7.3f ms/step" % (k, v * 1e3 / steps))
    tot += v
print("sum %.3f" % (tot * 1e3 / steps))
