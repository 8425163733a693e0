"""cProfile of the literal full workload's host side (bench.full_workload): which calls hold the host while the GPU
waits.  python tools/debug/full_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

args = bench.parse_args([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.full_workload(args, dev, steps=2, warmup=3)
pr = cProfile.Profile()
pr.enable()
out = bench.full_workload(args, dev, steps=10, warmup=1)
pr.disable()
print(out)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
