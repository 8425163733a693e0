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
    """One flat fp32 gradient buffer per replica, exchanged by all_reduce(SUM) and scaled by 1/world (about 20 MB for
    GCANet: ~0.4 ms on a 7-link xGMI ring).
    During backward the parameters' .grad are None, so autograd hands over its gradient tensors instead of launching
    one accumulate kernel per parameter into pre-zeroed views (which is what bucket-view DDP costs a model with ~150
    small parameters: more than the collective it hides); the gradients are packed into the buffer with one
    multi-tensor copy and every .grad re-pointed at its view (what the optimizer then reads).

    late: parameters at the UPSTREAM end of the network (the encoder) -- their gradients arrive last.  When given (and
    world > 1) the buffer is laid out [early | late]; the first late gradient to arrive means every early one is
    complete (all of them sit downstream), so the early segment -- the per-point heads, ~95 % of the bytes -- is packed
    and its all-reduce started asynchronously right there, overlapping the encoder's backward; the small late segment
    follows at the end of backward."""

    def __init__(self, module, world_size=None, late=None):
        allp = [p for p in module.parameters() if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        late_ids = {id(p) for p in late} if (late is not None and self.world > 1) else set()
        early = [p for p in allp if id(p) not in late_ids]
        latep = [p for p in allp if id(p) in late_ids]
        self.params = early + latep
        self.n_early = len(early) if latep else len(self.params)
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for i, p in enumerate(self.params):
            if i == self.n_early:
                self.split = off
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        if self.n_early == len(self.params):
            self.split = off
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._early_work = None
        self._early_done = False
        self._zero = set()                          # parameters whose flat-gradient segment is known to hold zeros
        self.early_started_in_backward = 0          # statistics for tests / logs
        if latep:
            for p in latep:
                p.register_post_accumulate_grad_hook(self._on_late_grad)

    def zero_grad(self):
        for p in self.params:
            p.grad = None
        self._early_work, self._early_done = None, False

    def sync_params(self, src=0):
        """Make replicas identical at start (same seed already does; this is the belt-and-braces broadcast)."""
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, src)

    def _pack(self, lo, hi):
        dst, src = [], []
        for p, v in zip(self.params[lo:hi], self.views[lo:hi]):
            if p.grad is None:                  # parameter unused this step: its segment of the flat gradient is zero
                if id(p) not in self._zero:     # ... and stays zero (all-reduce sums zeros, Adam only reads): one fill, not
                    v.zero_()                   # one per step
                    self._zero.add(id(p))
            elif p.grad.data_ptr() != v.data_ptr():
                self._zero.discard(id(p))
                dst.append(v)
                src.append(p.grad.detach().to(torch.float32))
            else:                               # accumulated in place into the view (no zero_grad in between)
                self._zero.discard(id(p))
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def _on_late_grad(self, _param):
        """First gradient of the late group: the early group is complete -> pack it and start its all-reduce."""
        if self._early_done or self.world <= 1:
            return
        if any(p.grad is None for p in self.params[:self.n_early]):
            return                                # not the expected order: everything goes at the end of backward
        self._early_done = True
        with torch.no_grad():
            self._pack(0, self.n_early)
            self._early_work = dist.all_reduce(self.flat[:self.split], op=dist.ReduceOp.SUM, async_op=True)
        self.early_started_in_backward += 1

    def pack_grads(self):
        self._pack(0 if not self._early_done else self.n_early, len(self.params))

    def all_reduce_grads(self):
        self.pack_grads()
        if self.world > 1:
            if self._early_done:
                dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM)
                self._early_work.wait()
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.mul_(1.0 / self.world)
        self._early_work = None


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (clouds are the unit; no data-path collective)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
