"""Drop-in for ``pointnet2_ops.pointnet2_utils`` (reference:
models/Pointnet2_PyTorch-master/pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py).

Same callables, argument order, dtypes, differentiability set and error behaviour
(RuntimeError on non-contiguous / wrong-dtype / CPU tensors, utils.h:5-25); every op is
one call into libgcanet_hip.so (csrc/pointnet2.hip).  Outputs are allocated here, as the
reference's C++ side does (e.g. group_points.cpp:22-24).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


def _check(t, name, dtype):
    _lib.require_cuda(t)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s tensor" % (name, "an int" if dtype == torch.int32 else "a float"))


def _run(name, like, *args):
    with _lib.on_device(like):
        _lib.call(name, *args, _lib.stream_of(like))


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) -> (B,npoint) int32  (pointnet2_utils.py:34-62)."""
        _check(xyz, "points", torch.float32)
        B, N, _ = xyz.shape
        out = torch.zeros(B, npoint, dtype=torch.int32, device=xyz.device)
        temp = torch.empty(B, N, dtype=torch.float32, device=xyz.device)
        _run("gcn_furthest_point_sampling", xyz, B, N, npoint, _lib.ptr(xyz), _lib.ptr(temp), _lib.ptr(out))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)  (pointnet2_utils.py:68-89)."""
        _check(features, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        ctx.save_for_backward(idx, features)
        B, Cc, N = features.shape
        m = idx.size(1)
        out = torch.empty(B, Cc, m, dtype=torch.float32, device=features.device)
        _run("gcn_gather_points", features, B, Cc, N, m, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, Cc, N = features.shape
        grad_out = grad_out.contiguous()
        g = torch.empty(B, Cc, N, dtype=torch.float32, device=grad_out.device)
        _run("gcn_gather_points_grad", grad_out, B, Cc, N, idx.size(1), _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(g))
        return g, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> dist (B,n,3) L2, idx (B,n,3) int32
        (pointnet2_utils.py:104-131)."""
        _check(unknown, "unknowns", torch.float32)
        _check(known, "knows", torch.float32)
        B, n, _ = unknown.shape
        m = known.size(1)
        dist2 = torch.empty(B, n, 3, dtype=torch.float32, device=unknown.device)
        idx = torch.empty(B, n, 3, dtype=torch.int32, device=unknown.device)
        _run("gcn_three_nn", unknown, B, n, m, _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2), _lib.ptr(idx))
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx/weight (B,n,3) -> (B,c,n)  (pointnet2_utils.py:139-162)."""
        _check(features, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        _check(weight, "weight", torch.float32)
        ctx.save_for_backward(idx, weight, features)
        B, c, m = features.shape
        n = idx.size(1)
        out = torch.empty(B, c, n, dtype=torch.float32, device=features.device)
        _run("gcn_three_interpolate", features, B, c, m, n, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(out))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, features = ctx.saved_tensors
        B, c, m = features.shape
        n = idx.size(1)
        grad_out = grad_out.contiguous()
        g = torch.empty(B, c, m, dtype=torch.float32, device=grad_out.device)
        _run("gcn_three_interpolate_grad", grad_out, B, c, n, m, _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(weight), _lib.ptr(g))
        return g, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)
        (pointnet2_utils.py:194-214)."""
        _check(features, "points", torch.float32)
        _check(idx, "idx", torch.int32)
        ctx.save_for_backward(idx, features)
        B, Cc, N = features.shape
        _, npoint, nsample = idx.shape
        out = torch.empty(B, Cc, npoint, nsample, dtype=torch.float32, device=features.device)
        _run("gcn_group_points", features, B, Cc, N, npoint, nsample, _lib.ptr(features), _lib.ptr(idx), _lib.ptr(out))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        B, Cc, N = features.shape
        _, npoint, nsample = idx.shape
        grad_out = grad_out.contiguous()
        g = torch.empty(B, Cc, N, dtype=torch.float32, device=grad_out.device)
        _run("gcn_group_points_grad", grad_out, B, Cc, N, npoint, nsample, _lib.ptr(grad_out), _lib.ptr(idx), _lib.ptr(g))
        return g, torch.zeros_like(idx)  # quirk kept: pointnet2_utils.py:237


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32
        (pointnet2_utils.py:243-270)."""
        _check(new_xyz, "new_xyz", torch.float32)
        _check(xyz, "xyz", torch.float32)
        B, N, _ = xyz.shape
        m = new_xyz.size(1)
        out = torch.empty(B, m, nsample, dtype=torch.int32, device=xyz.device)
        _run("gcn_ball_query", xyz, B, N, m, float(radius), nsample, _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(out))
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:279-338."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        xyz_trans = xyz.transpose(1, 2).contiguous()
        grouped_xyz = grouping_operation(xyz_trans, idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped_features = grouping_operation(features, idx)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
        return grouped_xyz


class GroupAll(nn.Module):
    """pointnet2_utils.py:341-379."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            if self.use_xyz:
                return torch.cat([grouped_xyz, grouped_features], dim=1)
            return grouped_features
        return grouped_xyz
