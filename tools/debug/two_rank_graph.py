"""Where does a two-ranks-on-one-GPU graph step spend its time?  (torchrun --nproc-per-node 2, gloo)"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
x = torch.randn(4096, 4096, device="cuda")
flat = torch.randn(1500000, device="cuda")


def work():
    y = x
    for _ in range(40):
        y = y @ x * 1e-3
    return y


for _ in range(3):
    work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    work()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out = work()
torch.cuda.synchronize()
dist.barrier()
for mode in ("eager", "graph"):
    ts = []
    for i in range(5):
        t0 = time.perf_counter()
        if mode == "eager":
            work()
        else:
            g.replay()
        t1 = time.perf_counter()
        dist.all_reduce(flat)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    if rank == 0:
        print(mode, ["%.1f/%.1f/%.1f ms" % tuple(1e3 * v for v in t) for t in ts])
dist.destroy_process_group()
