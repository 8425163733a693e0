"""Sparse 3-D convolutions and the instance "tiny U-Net" of the reference (softgroup/model/blocks.py:44-143,
models/dgcnn-hais-concat-direct-4.py:611-616,1379-1392) on csrc/sparseconv.hip.  SURVEY.md section 8(f) rank 3.

The reference takes SubMConv3d / SparseConv3d / SparseInverseConv3d / SparseConvTensor from the third-party `spconv`
package, which is neither vendored nor version-pinned: parity of this stage is "unpinned" by reference fixtures.  The
operators are therefore pinned to their DEFINITION -- a dense torch.nn.functional.conv3d / conv_transpose3d evaluated on
the densified grid and read back at the active sites (tests/test_sparseconv_gpu.py, oracle/ref_model.py) -- and the
module tree keeps the reference's names (blocks.block0.conv_branch.2 ...), so a state_dict maps one to one once the
weights are brought to this layout: (K, Cin, Cout) with K = 27 offsets (dx,dy,dz) in x-major order, resp. 8 corners.

A sparse tensor is the tuple clusters_voxelization returns (gcanet_amd/grouping.py): features (M,C) f32, indices (M,4)
int32 [sample, x, y, z], spatial_shape, batch_size.  Rule tables are built once per tensor geometry and shared by every
layer with the same `indice_key`, as in spconv.
"""
import math

import torch
from torch import nn

from . import _lib
from .layers import zeroed_like


def _call(name, like, *args):
    with _lib.on_device(like):
        _lib.call(name, *args, _lib.stream_of(like))


class SparseConvTensor:
    """features (M,C) f32 cuda, indices (M,4) int32 cuda [sample,x,y,z], spatial_shape (cubic: D), batch_size."""

    def __init__(self, features, indices, spatial_shape, batch_size, rules=None):
        self.features, self.indices = features, indices.int().contiguous()
        self.spatial_shape = [int(s) for s in spatial_shape]
        assert len(set(self.spatial_shape)) == 1, "cubic grids only (the reference uses [D, D, D], M4:1352)"
        self.batch_size = int(batch_size)
        self.rules = {} if rules is None else rules          # indice_key -> rule tables, shared along the network

    def replace_feature(self, features):
        return SparseConvTensor(features, self.indices, self.spatial_shape, self.batch_size, self.rules)

    @property
    def D(self):
        return self.spatial_shape[0]

    def dense(self):
        """(batch, C, D, D, D) -- for tests."""
        D, C = self.D, self.features.shape[1]
        out = self.features.new_zeros(self.batch_size, D, D, D, C)
        i = self.indices.long()
        out[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = self.features
        return out.permute(0, 4, 1, 2, 3).contiguous()


def subm_rules(x: SparseConvTensor):
    """(M,27) neighbour table of a submanifold 3x3x3 convolution."""
    _lib.require_cuda(x.indices)
    M = x.indices.shape[0]
    lib = _lib.lib()
    nbytes = lib.gcn_sparse_grid_bytes(x.batch_size, x.D)
    if nbytes > (16 << 30):
        raise RuntimeError("sparseconv: dense index grid of %d x %d^3 needs %.1f GB" % (x.batch_size, x.D, nbytes / 2**30))
    grid = torch.empty(nbytes // 4, dtype=torch.int32, device=x.indices.device)
    nbr = torch.empty(M, 27, dtype=torch.int32, device=x.indices.device)
    _call("gcn_sparse_subm_rules", x.indices, M, _lib.ptr(x.indices), x.batch_size, x.D, _lib.ptr(grid), _lib.ptr(nbr))
    return nbr


def coarse_rules(x: SparseConvTensor):
    """stride-2 / kernel-2: -> (coords2 (M2,4), child (M2,8), parent (M,8))."""
    _lib.require_cuda(x.indices)
    M, dev = x.indices.shape[0], x.indices.device
    ws = torch.empty(_lib.lib().gcn_sparse_coarse_ws_bytes(x.batch_size, x.D), dtype=torch.uint8, device=dev)
    coords2 = torch.empty(M, 4, dtype=torch.int32, device=dev)
    child = torch.empty(M, 8, dtype=torch.int32, device=dev)
    parent = torch.empty(M, 8, dtype=torch.int32, device=dev)
    m2 = torch.empty(1, dtype=torch.int32, device=dev)
    _call("gcn_sparse_coarse_rules", x.indices, M, _lib.ptr(x.indices), x.batch_size, x.D, _lib.ptr(ws), _lib.ptr(coords2),
          _lib.ptr(child), _lib.ptr(parent), _lib.ptr(m2))
    M2 = int(m2.item())                                       # the one size the host has to know (allocation)
    return coords2[:M2].contiguous(), child[:M2].contiguous(), parent


def _scratch(Mout, K, Cout, device):
    """Partial-sum scratch of a gather-GEMM launch whose offsets are split over workgroup groups (small Mout)."""
    n = _lib.lib().gcn_sparse_gather_gemm_ws_floats(Mout, K, Cout)
    return torch.empty(n, dtype=torch.float32, device=device) if n > 0 else None


class GatherGemmFunction(torch.autograd.Function):
    """out = sum_k in[rule[:,k]] @ W[k].  `rule_t` / `k_rev_t` describe the transposed gather (the input gradient);
    `rule_cols` = rule.t().contiguous() (K, Mout), the layout the weight-gradient kernel reads."""

    @staticmethod
    def forward(ctx, feats, weight, rule, rule_t, k_rev_t, rule_cols):
        _lib.require_cuda(feats, weight)
        feats, weight = feats.float().contiguous(), weight.float().contiguous()
        K, Cin, Cout = weight.shape
        Mout = rule.shape[0]
        out = torch.empty(Mout, Cout, dtype=torch.float32, device=feats.device)
        _call("gcn_sparse_gather_gemm", feats, Mout, K, Cin, Cout, _lib.ptr(feats), _lib.ptr(rule), _lib.ptr(weight), 0, 0,
              _lib.ptr(out), _lib.ptr(_scratch(Mout, K, Cout, feats.device)))
        ctx.save_for_backward(feats, weight, rule_cols, rule_t)
        ctx.k_rev_t = bool(k_rev_t)
        return out

    @staticmethod
    def backward(ctx, dout):
        feats, weight, rule_cols, rule_t = ctx.saved_tensors
        K, Cin, Cout = weight.shape
        dout = dout.float().contiguous()
        din = dw = None
        if ctx.needs_input_grad[0]:
            Min = rule_t.shape[0]
            din = torch.empty(Min, Cin, dtype=torch.float32, device=dout.device)
            wt = weight.transpose(1, 2).contiguous()              # (K, Cout, Cin): the weight of the transposed product
            _call("gcn_sparse_gather_gemm", dout, Min, K, Cout, Cin, _lib.ptr(dout), _lib.ptr(rule_t), _lib.ptr(wt), 0,
                  int(ctx.k_rev_t), _lib.ptr(din), _lib.ptr(_scratch(Min, K, Cin, dout.device)))
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            _call("gcn_sparse_wgrad", dout, rule_cols.shape[1], K, Cin, Cout, _lib.ptr(feats), _lib.ptr(rule_cols),
                  _lib.ptr(dout), _lib.ptr(dw))
        return din, dw, None, None, None, None


class BatchNormReLUFunction(torch.autograd.Function):
    """[ReLU](BatchNorm1d(x)) with batch statistics over the rows (csrc/sparseconv.hip: bn_* kernels)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, relu):
        _lib.require_cuda(x)
        x = x.float().contiguous()
        M, C = x.shape
        ga, be = gamma.float().contiguous(), beta.float().contiguous()
        y = torch.empty_like(x)
        mean_rstd = torch.empty(C, 2, dtype=torch.float32, device=x.device)
        ws = zeroed_like((2 * C,), torch.float64, x.device)          # call-local accumulators: from the step's pre-zeroed arena
        _call("gcn_bn_relu_fwd", x, M, C, _lib.ptr(x), _lib.ptr(ga), _lib.ptr(be), float(eps), int(relu), float(momentum),
              _lib.ptr(y), _lib.ptr(mean_rstd), _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(ws))
        ctx.save_for_backward(x, ga, be, mean_rstd)
        ctx.relu = int(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ga, be, mean_rstd = ctx.saved_tensors
        M, C = x.shape
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
        ws = zeroed_like((2 * C,), torch.float64, x.device)
        _call("gcn_bn_relu_bwd", x, M, C, _lib.ptr(dy), _lib.ptr(x), _lib.ptr(ga), _lib.ptr(be), _lib.ptr(mean_rstd), ctx.relu,
              _lib.ptr(dx), _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(ws))
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_relu(x, bn: nn.BatchNorm1d, relu=True):
    """bn(x) [+ ReLU] for features (M,C).  Training mode with >= 2 rows: the fused kernels (running statistics and
    num_batches_tracked updated as the module would); otherwise the module itself."""
    if not (bn.training and x.is_cuda and x.shape[0] >= 2 and bn.momentum is not None and bn.affine
            and x.shape[1] % 4 == 0 and 256 % (x.shape[1] // 4) == 0):
        y = bn(x)
        return torch.relu(y) if relu else y
    if bn.track_running_stats:
        bn.num_batches_tracked.add_(1)
    return BatchNormReLUFunction.apply(x, bn.weight, bn.bias, bn.running_mean if bn.track_running_stats else None,
                                       bn.running_var if bn.track_running_stats else None, bn.momentum, bn.eps, relu)


class _SparseConvBase(nn.Module):
    K = 27

    def __init__(self, in_channels, out_channels, indice_key=None):
        super().__init__()
        assert in_channels % 64 == 0 and out_channels % 64 == 0, "channels are multiples of 64 in the tiny U-Net"
        self.in_channels, self.out_channels, self.indice_key = in_channels, out_channels, indice_key
        self.weight = nn.Parameter(torch.empty(self.K, in_channels, out_channels))
        nn.init.kaiming_uniform_(self.weight.view(-1, out_channels).t(), a=math.sqrt(5))      # fan_in = K * Cin


class SubMConv3d(_SparseConvBase):
    """spconv.SubMConv3d(kernel_size=3, padding=1, bias=False): outputs live on the input's active sites."""
    K = 27

    def forward(self, x: SparseConvTensor):
        key = ("subm", self.indice_key)
        if key not in x.rules:
            nbr = subm_rules(x)
            x.rules[key] = (nbr, nbr.t().contiguous())
        nbr, nbr_cols = x.rules[key]
        return x.replace_feature(GatherGemmFunction.apply(x.features, self.weight, nbr, nbr, True, nbr_cols))


class SparseConv3d(_SparseConvBase):
    """spconv.SparseConv3d(kernel_size=2, stride=2, bias=False)."""
    K = 8

    def forward(self, x: SparseConvTensor):
        key = ("spconv", self.indice_key)
        if key not in x.rules:
            coords2, child, parent = coarse_rules(x)
            x.rules[key] = (coords2, child, parent, x.indices, x.spatial_shape, child.t().contiguous(), parent.t().contiguous())
        coords2, child, parent, _, _, child_cols, _ = x.rules[key]
        feats = GatherGemmFunction.apply(x.features, self.weight, child, parent, False, child_cols)
        return SparseConvTensor(feats, coords2, [(x.D + 1) // 2] * 3, x.batch_size, x.rules)


class SparseInverseConv3d(_SparseConvBase):
    """spconv.SparseInverseConv3d(kernel_size=2, indice_key=...): back onto the sites the paired SparseConv3d consumed."""
    K = 8

    def forward(self, x: SparseConvTensor):
        coords2, child, parent, fine_indices, fine_shape, _, parent_cols = x.rules[("spconv", self.indice_key)]
        feats = GatherGemmFunction.apply(x.features, self.weight, parent, child, False, parent_cols)
        return SparseConvTensor(feats, fine_indices, fine_shape, x.batch_size, x.rules)


class SparseSequential(nn.Sequential):
    """spconv.SparseSequential: dense modules act on .features."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (_SparseConvBase, SparseSequential, ResidualBlock, UBlock, Custom1x1Subm3d)):
                x = m(x)
            elif isinstance(m, nn.Identity):
                pass
            elif isinstance(m, nn.BatchNorm1d):                 # norm_fn is always followed by ReLU in blocks.py: fused
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = x.replace_feature(batch_norm_relu(x.features, m, relu=fuse))
                i += int(fuse)
            else:
                x = x.replace_feature(m(x.features))
            i += 1
        return x


class Custom1x1Subm3d(nn.Module):
    """blocks.py:31-41: a 1x1 "convolution" as a plain matmul on the features."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, x):
        return x.replace_feature(x.features @ self.weight.t())


class ResidualBlock(nn.Module):
    """blocks.py:44-80."""

    def __init__(self, in_channels, out_channels, norm_fn, indice_key=None):
        super().__init__()
        self.i_branch = SparseSequential(nn.Identity() if in_channels == out_channels
                                         else Custom1x1Subm3d(in_channels, out_channels))
        self.conv_branch = SparseSequential(
            norm_fn(in_channels), nn.ReLU(), SubMConv3d(in_channels, out_channels, indice_key),
            norm_fn(out_channels), nn.ReLU(), SubMConv3d(out_channels, out_channels, indice_key))

    def forward(self, x):
        out = self.conv_branch(x)
        return out.replace_feature(out.features + self.i_branch(x).features)


class UBlock(nn.Module):
    """blocks.py:83-143."""

    def __init__(self, nPlanes, norm_fn, block_reps, block=ResidualBlock, indice_key_id=1):
        super().__init__()
        self.nPlanes = nPlanes
        self.blocks = SparseSequential()
        for i in range(block_reps):
            self.blocks.add_module("block%d" % i, block(nPlanes[0], nPlanes[0], norm_fn, "subm%d" % indice_key_id))
        if len(nPlanes) > 1:
            self.conv = SparseSequential(norm_fn(nPlanes[0]), nn.ReLU(),
                                         SparseConv3d(nPlanes[0], nPlanes[1], "spconv%d" % indice_key_id))
            self.u = UBlock(nPlanes[1:], norm_fn, block_reps, block, indice_key_id + 1)
            self.deconv = SparseSequential(norm_fn(nPlanes[1]), nn.ReLU(),
                                           SparseInverseConv3d(nPlanes[1], nPlanes[0], "spconv%d" % indice_key_id))
            self.blocks_tail = SparseSequential()
            for i in range(block_reps):
                self.blocks_tail.add_module("block%d" % i, block(nPlanes[0] * (2 - i), nPlanes[0], norm_fn,
                                                                 "subm%d" % indice_key_id))

    def forward(self, x):
        out = self.blocks(x)
        if len(self.nPlanes) > 1:
            dec = self.deconv(self.u(self.conv(out)))
            out = out.replace_feature(torch.cat((out.features, dec.features), dim=1))
            out = self.blocks_tail(out)
        return out


class MLP(nn.Sequential):
    """blocks.py:10-27."""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        mods = []
        for _ in range(num_layers - 1):
            mods.append(nn.Linear(in_channels, in_channels))
            if norm_fn:
                mods.append(norm_fn(in_channels))
            mods.append(nn.ReLU())
        mods.append(nn.Linear(in_channels, out_channels))
        super().__init__(*mods)


class InstanceHead(nn.Module):
    """The sparse half of the reference model: tiny_unet + output layer + the three score heads (M4:611-616) and
    forward_instance (M4:1379-1392)."""

    def __init__(self, channels=64, semantic_classes=10):
        super().__init__()
        import functools
        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.tiny_unet = UBlock([channels, 2 * channels], norm_fn, 2, ResidualBlock, indice_key_id=11)
        self.tiny_unet_outputlayer = SparseSequential(norm_fn(channels), nn.ReLU())
        self.cls_linear = nn.Linear(channels, semantic_classes)
        self.mask_linear = MLP(channels, semantic_classes, norm_fn=None, num_layers=2)
        self.iou_score_linear = nn.Linear(channels, semantic_classes)

    def forward(self, inst_feats: SparseConvTensor, inst_map):
        from .grouping import global_pool
        feats = self.tiny_unet_outputlayer(self.tiny_unet(inst_feats))
        mask_scores = self.mask_linear(feats.features).index_select(0, inst_map.long())   # backward = one index_add
        instance_batch_idxs = feats.indices[:, 0][inst_map.long()]
        pooled = global_pool(feats.features, feats.indices[:, 0], feats.batch_size)
        return instance_batch_idxs, self.cls_linear(pooled), self.iou_score_linear(pooled), mask_scores
