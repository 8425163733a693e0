"""Host-side mirror of the offset module of the reference's earlier model variant
(models/dgcnn-hais-concat-direct-2.py, "M2": KPAM :326-348, OFFSET_PRED_MODULE :351-462) -- the one place where
the reference runs its native kNN (KNN_CUDA) and pointnet2 `grouping_operation` inside the network graph:

    group_points(k=60, key points, all points, key features)  ->  search_knn.py:23-39 -> KNN + grouping_operation
    grouping_operation(semantic / instance features, idx)      ->  M2:412-415

Same class / parameter names and the same forward signature as M2; the ops are this package's drop-ins
(gcanet_amd.search_knn.group_points, gcanet_amd.pointnet2_ops.pointnet2_utils.grouping_operation), i.e. the HIP
kNN and the HIP gather / LDS scatter-add gradient.  Quirks kept as they are in the reference: the semantic and
instance features are gathered from the FULL cloud with key-point ranks (idx < sampling_ratio) (M2:412-415); the
semantic distances are computed and not used (the `attention_seg` call is commented out, M2:444); KPAM here is a
sigmoid gate, not M4's softmax."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .pointnet2_ops.pointnet2_utils import grouping_operation
from .search_knn import group_points


def inst_and_seg_dist(semantic_feature, semantic_feature_knn, instance_feature, instance_feature_knn):
    """M2:305-323: L2 distance of every point's feature to its k grouped features -> (B,N,1,k) each."""
    distances_semantic = torch.cdist(semantic_feature.unsqueeze(2), semantic_feature_knn, p=2)
    distances_instance = torch.cdist(instance_feature.unsqueeze(2), instance_feature_knn, p=2)
    return distances_semantic, distances_instance


class KPAM(nn.Module):
    """M2:326-348: Conv1d(k->k) - ReLU - Conv1d(k->k) over the neighbour axis of the (B,N,1,k) distances, sigmoid,
    broadcast over the feature axis and multiplied into x (B,N,k,F)."""

    def __init__(self, C):
        super().__init__()
        self.dim = C
        self.conv1 = nn.Sequential(nn.Conv1d(C, C, kernel_size=1, bias=False), nn.ReLU(),
                                   nn.Conv1d(C, C, kernel_size=1, bias=False))
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, attention_feature):
        a = attention_feature.squeeze(2)                                       # (B,N,k)
        # the two 1x1 convolutions act along the last axis of the point-major tensor: plain GEMMs, same parameters
        a = torch.relu(a @ self.conv1[0].weight.flatten(1).t()) @ self.conv1[2].weight.flatten(1).t()
        return self.sigmoid(a).unsqueeze(-1) * x


class OFFSET_PRED_MODULE(nn.Module):
    """M2:351-462.  forward(points (B,N,3), feature (B,N,128), semantic_feature (B,N,Cs), instance_feature (B,N,Ci),
    index) -> offsets (B,3,N).  `index` is accepted and unused, as in the reference."""

    def __init__(self, nn_nb=60, sampling_ratio=120):
        super().__init__()
        self.k = nn_nb
        self.dilation_factor = 1
        self.drop = 0.0
        self.sampling_ratio = sampling_ratio
        self.bn1 = nn.GroupNorm(2, 128)
        self.conv1 = nn.Sequential(nn.Conv2d(131, 128, kernel_size=1, bias=False), self.bn1,
                                   nn.LeakyReLU(negative_slope=0.2))
        self.attention_seg = KPAM(nn_nb)
        self.attention_inst = KPAM(nn_nb)
        self.mlp_offset = nn.Conv1d(256, 3, 1)

    def forward(self, points, feature, semantic_feature, instance_feature, index=None):
        from .dgcnn import grouped_block, key_point_indices
        B, N, _ = points.shape
        sub = key_point_indices(N, self.sampling_ratio, points.device)         # M2:379-384 (legacy NumPy RNG, seed 1234)
        key_points = points[:, sub].permute(0, 2, 1).contiguous()              # (B,3,n_sub)
        feature_sampling = feature[:, sub].permute(0, 2, 1).contiguous()       # (B,128,n_sub)
        pts_cm = points.permute(0, 2, 1).contiguous()                          # (B,3,N)
        # native kNN of every point among the key points + gathers (M2:401-404 -> search_knn.py:23-39)
        points_knn, feature_knn, idx = group_points(self.k, point_cloud=key_points, query_cloud=pts_cm,
                                                    point_features=feature_sampling)
        sem_cm = semantic_feature.permute(0, 2, 1).contiguous()
        ins_cm = instance_feature.permute(0, 2, 1).contiguous()
        semantic_feature_knn = grouping_operation(sem_cm, idx).permute(0, 2, 3, 1)     # (B,N,k,Cs)   M2:412
        instance_feature_knn = grouping_operation(ins_cm, idx).permute(0, 2, 3, 1)     # (B,N,k,Ci)   M2:415
        direction = points_knn.permute(0, 2, 3, 1) - points.unsqueeze(2)               # (B,N,k,3)    M2:429
        f = torch.cat([feature_knn.permute(0, 2, 3, 1), direction], 3)                 # (B,N,k,131)
        _, distances_instance = inst_and_seg_dist(semantic_feature, semantic_feature_knn, instance_feature,
                                                  instance_feature_knn)
        f = self.attention_inst(f, distances_instance)                                 # M2:445
        _lib.require_cuda(f)   # no CPU path in the product (the ops above raise on CPU tensors as well)
        # Conv2d(131->128) + GroupNorm + LeakyReLU + max over k as ONE fused grouped block
        y = grouped_block(f, self.conv1[0].weight, self.bn1.weight, self.bn1.bias, self.bn1.num_groups,
                          self.bn1.eps, 0.2, dtype="f32")                              # (B,128,N)
        y = torch.cat([y, feature.permute(0, 2, 1)], dim=1)                            # (B,256,N)
        return self.mlp_offset(y)
