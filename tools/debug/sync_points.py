"""Where the literal full workload synchronises with the host: torch.cuda.set_sync_debug_mode("warn") over one step."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.full_workload(args, dev, steps=1, warmup=2)
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as rec:
    warnings.simplefilter("always")
    bench.full_workload(args, dev, steps=1, warmup=0)
torch.cuda.set_sync_debug_mode("default")
for w in rec:
    print("%s:%d  %s" % (w.filename.split("/repo/")[-1], w.lineno, str(w.message)[:100]))
