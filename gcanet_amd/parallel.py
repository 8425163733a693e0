"""Data parallelism for the hot path: one process per GPU, full replica each, ONE RCCL all-reduce of
the flattened fp32 gradient per step over xGMI.  The reference wraps the
model in single-process nn.DataParallel (trainer_new.py:94-96: broadcast params + scatter + gather
every step); that design is not reproduced.  Clouds are independent (GroupNorm statistics are
per-sample), so there is no collective inside forward/backward.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun-style env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_*).  Returns
    (rank, local_rank, world_size).  backend: 'nccl' (= RCCL on ROCm) on GPUs, 'gloo' on CPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("GCANET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class FlatGradDP:
    """One flat fp32 gradient bucket per replica: the gradient exchange is a single all_reduce(SUM) then a scale
    by 1/world (about 20 MB for GCANet: ~0.25 ms on a 7-link xGMI ring).
    During backward the parameters' .grad are None, so autograd hands over its gradient tensors instead of
    launching one accumulate kernel per parameter into pre-zeroed views; `all_reduce_grads` packs them into the
    bucket with one multi-tensor copy and re-points every .grad at its view (what the optimizer then reads)."""

    def __init__(self, module, world_size=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def sync_params(self, src=0):
        """Make replicas identical at start (same seed already does; this is the belt-and-braces broadcast)."""
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src)

    def pack_grads(self):
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()                       # parameter unused this step
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad.detach().to(torch.float32))
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def all_reduce_grads(self):
        self.pack_grads()
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (clouds are the unit; no data-path collective)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
