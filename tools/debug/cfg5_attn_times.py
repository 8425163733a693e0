"""Per-entry-point device times of the cfg5 workload (HIP events around every library call): which attention shapes cost what."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.cfg5_workload(dev, steps=1, warmup=2)
_lib.enable_timing(True)
steps = 3
bench.cfg5_workload(dev, steps=steps, warmup=0)
torch.cuda.synchronize()
res = _lib.timing_results()
_lib.enable_timing(False)
for k, (n, ms) in sorted(res.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%-60s %5.1f calls/step %8.3f ms/step  %7.1f us each" % (k, n / steps, ms / steps, 1e3 * ms / n))
