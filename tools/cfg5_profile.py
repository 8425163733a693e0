"""bench.cfg5_workload on its own (for rocprofv3 --kernel-trace --stats): python3 tools/cfg5_profile.py [steps warmup]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

steps, warmup = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3, 2)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
print(bench.cfg5_workload(dev, steps=steps, warmup=warmup))
