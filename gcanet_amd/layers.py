"""Point-major building blocks for the per-point heads: activations live as (B, N, C) so that one point
is one contiguous row (what both the row gathers and the per-point GEMMs want).  GroupNorm(+ReLU) runs
through csrc/gn.hip; the 1x1 convolutions are plain library GEMMs (torch.nn.functional.linear)."""
import torch

from . import _lib


def _run(name, like, *args):
    with torch.cuda.device_of(like):
        _lib.call(name, *args, _lib.stream_of(like))


def _acc_buffers(n_f64, C, device):
    """(S (n_f64,) f64, dgamma (C,) f32, dbeta (C,) f32) carved back to back from ONE allocation, so that the library
    zeroes them with a single fill (csrc/common.h:zero_spans) instead of three 4.5-us launches."""
    raw = torch.empty(n_f64 * 8 + 2 * C * 4, dtype=torch.uint8, device=device)
    S = raw[:n_f64 * 8].view(torch.float64)
    dg = raw[n_f64 * 8:n_f64 * 8 + C * 4].view(torch.float32)
    db = raw[n_f64 * 8 + C * 4:].view(torch.float32)
    return S, dg, db


class GroupNormReLUFunction(torch.autograd.Function):
    """y = [ReLU](GroupNorm(x)) for x (B,N,C) f32 or bf16 (output dtype = input dtype)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        _lib.require_cuda(x)
        assert x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        B, N, C = x.shape
        dt = 1 if x.dtype == torch.bfloat16 else 0
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x)
        mean_rstd = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        ws = torch.empty(B, groups, 2, dtype=torch.float64, device=x.device)
        _run("gcn_gn_fwd", x, _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), B, N, C, groups, float(eps), int(relu),
             _lib.ptr(y), _lib.ptr(mean_rstd), _lib.ptr(ws))
        ctx.save_for_backward(x, ga, be, mean_rstd)
        ctx.cfg = (groups, relu, dt)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ga, be, mean_rstd = ctx.saved_tensors
        groups, relu, dt = ctx.cfg
        B, N, C = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        ws, dgamma, dbeta = _acc_buffers(B * groups * 2, C, x.device)
        _run("gcn_gn_bwd", x, _lib.ptr(dy), _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd), B, N, C,
             groups, int(relu), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None


def group_norm_relu(x, gn, relu=True):
    """x (B,N,C) point-major; gn: an nn.GroupNorm holding (num_groups, weight, bias, eps)."""
    return GroupNormReLUFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


def tall_skinny_tn(a, b, chunks=32, out_dtype=None):
    """a^T @ b for a (M,P), b (M,Q) with M >> P,Q (weight-gradient shape).  The library picks a tile
    config with a few dozen workgroups for the direct call (each looping over all M rows: 150-350 us at
    M=65536 on MI355X); splitting the long reduction into `chunks` batched GEMMs + a sum fills the chip.
    out_dtype: the partial sums are added up directly into that type (no separate cast kernel)."""
    M = a.shape[0]
    if M % chunks != 0 or M // chunks < 64:
        r = a.t() @ b
        return r if out_dtype is None else r.to(out_dtype)
    part = torch.bmm(a.view(chunks, M // chunks, -1).transpose(1, 2), b.view(chunks, M // chunks, -1))
    return part.sum(0, dtype=out_dtype)


class CastCache:
    """bf16 (autocast-dtype) copies of a model's floating-point parameters, refreshed by ONE multi-tensor copy per
    step instead of one cast kernel per layer and call (a training step had ~40 of them, 5 us each).  LinearPMFunction
    looks a weight up by identity and version; anything not registered (weight slices, ...) is cast as before."""
    _live = None

    def __init__(self, module, dtype=torch.bfloat16):
        self.params = [p for p in module.parameters() if p.dtype == torch.float32 and p.dim() >= 1]
        self.copies = [torch.empty_like(p, dtype=dtype) for p in self.params]
        self.dtype = dtype
        self.version = {}
        self.by_id = {id(p): i for i, p in enumerate(self.params)}

    def refresh(self):
        """Call once per step after the optimizer update (or before the first forward)."""
        with torch.no_grad():
            torch._foreach_copy_(self.copies, self.params)
        self.version = {id(p): p._version for p in self.params}
        CastCache._live = self

    @staticmethod
    def lookup(t, dtype):
        """A current low-precision copy of parameter (or flatten(1) view of a parameter) `t`, or None."""
        c = CastCache._live
        if c is None or c.dtype != dtype:
            return None
        base = t._base if t._base is not None else t
        i = c.by_id.get(id(base))
        if (i is None or c.version.get(id(base)) != base._version or base.numel() != t.numel() or not t.is_contiguous()
                or t.storage_offset() != base.storage_offset()):      # only whole-tensor reshapes of the parameter
            return None
        return c.copies[i].view(t.shape)


class LinearPMFunction(torch.autograd.Function):
    """y = x @ W^T + b on point-major rows with a split-K weight gradient (see tall_skinny_tn).
    Runs in the autocast dtype (bf16 under torch.autocast, as the plain F.linear would)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        xc = x.to(dt)
        wc = (CastCache.lookup(weight, dt) if weight.dtype != dt else None)
        wc = weight.to(dt) if wc is None else wc
        bc = None
        if bias is not None:
            bc = CastCache.lookup(bias, dt) if bias.dtype != dt else None
            bc = bias.to(dt) if bc is None else bc
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.in_dtypes = (x.dtype, weight.dtype)
        with torch.autocast("cuda", enabled=False):
            return torch.nn.functional.linear(xc, wc, bc)

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        dy = dy.to(xc.dtype).contiguous()
        with torch.autocast("cuda", enabled=False):
            dx = (dy @ wc).to(ctx.in_dtypes[0])
            rows = dy.reshape(-1, dy.shape[-1])
            dw = tall_skinny_tn(rows, xc.reshape(-1, xc.shape[-1]), out_dtype=ctx.in_dtypes[1])
            db = rows.sum(0, dtype=torch.float32) if ctx.has_bias else None   # f32 accumulation, no f32 copy of dy
        return dx, dw, db


def linear_pm(x, weight, bias=None):
    return LinearPMFunction.apply(x, weight, bias)


def conv1x1(x, conv):
    """Conv1d(kernel 1) applied to point-major x (B,N,Cin) as a GEMM with the SAME parameter tensor."""
    return linear_pm(x, conv.weight.flatten(1), conv.bias)      # (Cout,Cin,1) -> (Cout,Cin) view


class GlobalMaxPoolFunction(torch.autograd.Function):
    """x (B,N,C) -> max over points (B,C)  (M4:513 `x.max(dim=2)` on the channel-major tensor).  Backward routes the
    gradient to the arg-max row only: one zero fill + a (B,C)-element scatter instead of torch's dense
    `grad * (x == max) / count` passes over (B,N,C)."""

    @staticmethod
    def forward(ctx, x):
        vals, arg = x.max(dim=1)
        ctx.save_for_backward(arg)
        ctx.shape = x.shape
        return vals

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        g = torch.zeros(ctx.shape, dtype=dout.dtype, device=dout.device)
        g.scatter_(1, arg.unsqueeze(1), dout.unsqueeze(1))
        return g


def global_max_pool(x):
    return GlobalMaxPoolFunction.apply(x)


class GroupNormReLUMaxFunction(torch.autograd.Function):
    """max over points of [ReLU](GroupNorm(x)) for x (B,N,C) -> (B,C), the (B,N,C) activation never written
    (csrc/gn.hip: gn_apply_max_kernel).  Backward routes the gradient to the arg-max rows and runs the ordinary
    GroupNorm backward on that (B,N,C) gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        _lib.require_cuda(x)
        assert x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
        x = x.contiguous()
        B, N, C = x.shape
        dt = 1 if x.dtype == torch.bfloat16 else 0
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        vals = torch.empty(B, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(B, C, dtype=torch.int64, device=x.device)
        mean_rstd = torch.empty(B, groups, 2, dtype=torch.float32, device=x.device)
        ws = torch.empty(B, groups, 2, dtype=torch.float64, device=x.device)
        best = torch.empty(B, C, dtype=torch.int64, device=x.device)
        _run("gcn_gn_max_fwd", x, _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), B, N, C, groups, float(eps), int(relu),
             _lib.ptr(vals), _lib.ptr(arg), _lib.ptr(mean_rstd), _lib.ptr(ws), _lib.ptr(best))
        ctx.save_for_backward(x, ga, be, mean_rstd, arg)
        ctx.cfg = (groups, relu, dt)
        return vals.to(x.dtype)

    @staticmethod
    def backward(ctx, dout):
        x, ga, be, mean_rstd, arg = ctx.saved_tensors
        groups, relu, dt = ctx.cfg
        B, N, C = x.shape
        dy = torch.zeros(B, N, C, dtype=x.dtype, device=x.device)
        dy.scatter_(1, arg.unsqueeze(1), dout.to(x.dtype).unsqueeze(1))
        dx = torch.empty_like(x)
        ws, dgamma, dbeta = _acc_buffers(B * groups * 2, C, x.device)
        _run("gcn_gn_bwd", x, _lib.ptr(dy), _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd), B, N, C,
             groups, int(relu), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None


def group_norm_relu_max(x, gn, relu=True):
    """max over dim 1 of group_norm_relu(x, gn): (B,N,C) -> (B,C)."""
    return GroupNormReLUMaxFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


class ParamNormaliseFunction(torch.autograd.Function):
    """(..., 22) parameter rows: unit-normalise the direction triples 4:7, 8:11, 15:18 (M4:664-676)."""

    @staticmethod
    def forward(ctx, p):
        _lib.require_cuda(p)
        assert p.shape[-1] == 22
        pc = p.float().contiguous()
        out = torch.empty_like(pc)
        _run("gcn_param_normalise_fwd", pc, _lib.ptr(pc), pc.numel() // 22, _lib.ptr(out))
        ctx.save_for_backward(pc)
        ctx.in_dtype = p.dtype
        return out

    @staticmethod
    def backward(ctx, go):
        (pc,) = ctx.saved_tensors
        go = go.float().contiguous()
        gi = torch.empty_like(pc)
        _run("gcn_param_normalise_bwd", pc, _lib.ptr(pc), _lib.ptr(go), pc.numel() // 22, _lib.ptr(gi))
        return gi.to(ctx.in_dtype)


def param_normalise(p):
    return ParamNormaliseFunction.apply(p)
