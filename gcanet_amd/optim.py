"""Adam over flat buffers (csrc/optim.hip): the reference trains with torch.optim.Adam (option_new.py:83-90); same update
rule, one elementwise kernel over all parameters instead of torch's multi-tensor launches (0.2 ms -> ~15 us per step for
this model's 57 tensors / 1.5 M parameters)."""
import torch

from . import _lib


class FlatAdam:
    """Adam(lr, betas, eps, weight_decay) on the parameters of a parallel.FlatGradDP: the parameters are re-pointed at
    views of one flat fp32 buffer (same order as the flat gradient), moments live in two more flat buffers, the step count
    on the device (graph-capturable).  step() consumes dp.flat, i.e. call it after dp.all_reduce_grads()."""

    def __init__(self, dp, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.dp = dp
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        n = dp.flat.numel()
        dev = dp.flat.device
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        off = 0
        with torch.no_grad():
            for p in dp.params:
                k = p.numel()
                self.flat_p[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + k].view_as(p)          # the module now reads / the kernel updates this memory
                off += k
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)    # [t, 1 - b1^t, 1 - b2^t, -]

    @torch.no_grad()
    def step(self):
        f = self.dp.flat
        if not f.is_cuda:                                               # CPU rehearsal path: plain torch, same rule
            self.state[0] += 1
            t = float(self.state[0])
            b1, b2 = self.betas
            g = f + self.weight_decay * self.flat_p if self.weight_decay else f
            self.m.lerp_(g, 1 - b1)
            self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = self.v.sqrt() / (1 - b2 ** t) ** 0.5 + self.eps
            self.flat_p.addcdiv_(self.m, denom, value=-self.lr / (1 - b1 ** t))
            return
        with _lib.on_device(f):
            _lib.call("gcn_adam_flat", _lib.ptr(self.flat_p), _lib.ptr(f), _lib.ptr(self.m), _lib.ptr(self.v), f.numel(),
                      self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, _lib.ptr(self.state),
                      _lib.stream_of(f))
