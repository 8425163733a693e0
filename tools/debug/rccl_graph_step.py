"""Rehearsal of a CAPTURED multi-rank step on the one GPU of a test box: a 1-rank `nccl` process group (RCCL), FlatGradDP
told it has two replicas so that every collective of the multi-rank step is really issued -- the heads' all-reduce
started asynchronously from the backward hook, the encoder's at the end, the wait, the 1/world scaling -- and the whole
step (forward, backward, both all-reduces, Adam) captured into ONE HIP graph (RCCL collectives are capturable).  One
replay is compared with one eager step from the same saved state.  Prints `OK ...` on success.
Run in a fresh process per attempt (tests/test_parallel_gpu.py does)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from gcanet_amd import dgcnn  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = dgcnn.PrimitivesEmbeddingDGCNGn(nn_nb=64, dtype="bf16").to(dev)
pts, nrm = bench.synth_clouds(range(B), 8192, dev)
st = bench.make_step(model, pts, nrm, world=2)          # "two replicas": the collectives are issued, gradients halved
opt, dp = st["opt"], st["dp"]
graph, _ = bench.capture_step(st["step"], 2)
torch.cuda.synchronize()
early_in_capture = dp.early_started_in_backward
saved = [t.clone() for t in (opt.flat_p, opt.m, opt.v, opt.state)]


def one(fn):
    for t, s_ in zip((opt.flat_p, opt.m, opt.v, opt.state), saved):
        t.copy_(s_)
    fn()
    torch.cuda.synchronize()
    return dp.flat.clone(), opt.flat_p.clone()


g_b, p_b = one(graph.replay)
g_a, p_a = one(st["step"])
g_c, p_c = one(st["step"])
gs = float(g_a.abs().max())
d, noise = float((g_a - g_b).abs().max()) / gs, float((g_a - g_c).abs().max()) / gs
rel_l2 = float((g_a - g_b).norm() / g_a.norm())
ok = d <= max(4 * noise, 5e-3) and rel_l2 <= 1e-3 and bool(torch.isfinite(p_b).all()) and float((p_b - saved[0]).abs().max()) > 1e-4   # (bar: see test_graph_replay_equals_eager_step)
print("%s captured step with RCCL all-reduce: graph vs eager gradients %.2e of max|g| (eager vs eager %.2e); the early "
      "all-reduce was started inside backward in %d steps incl. the capture" % ("OK" if ok else "FAILED", d, noise, early_in_capture))
dist.destroy_process_group()
sys.exit(0 if ok else 1)
