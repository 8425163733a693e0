"""Point-major building blocks for the per-point heads: activations live as (B, N, C) so that one point
is one contiguous row (what both the row gathers and the per-point GEMMs want).  GroupNorm(+ReLU) runs
through csrc/gn.hip; the 1x1 convolutions are plain library GEMMs (torch.nn.functional.linear)."""
import torch

from . import _lib


def _run(name, like, *args):
    with torch.cuda.device_of(like):
        _lib.call(name, *args, _lib.stream_of(like))


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
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(B, groups, 2, dtype=torch.float64, device=x.device)
        _run("gcn_gn_bwd", x, _lib.ptr(dy), _lib.ptr(x), dt, _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd), B, N, C,
             groups, int(relu), _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None


def group_norm_relu(x, gn, relu=True):
    """x (B,N,C) point-major; gn: an nn.GroupNorm holding (num_groups, weight, bias, eps)."""
    return GroupNormReLUFunction.apply(x, gn.weight, gn.bias, gn.num_groups, gn.eps, relu)


def conv1x1(x, conv):
    """Conv1d(kernel 1) applied to point-major x (B,N,Cin) as a GEMM with the SAME parameter tensor."""
    return torch.nn.functional.linear(x, conv.weight[:, :, 0], conv.bias)
