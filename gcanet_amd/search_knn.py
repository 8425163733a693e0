"""Drop-in for ``models/search_knn.py``: ``knn_point``, ``group_points`` and ``SoftProjection``
(same signatures), on top of the fused HIP kNN and the pointnet2 grouping op."""
import torch
import torch.nn as nn

from .knn_cuda import KNN
from .pointnet2_ops.pointnet2_utils import grouping_operation as group_point


def knn_point(group_size, point_cloud, query_cloud):
    """search_knn.py:11-14 -> (dist (B,k,Q), idx (B,k,Q) int64)."""
    return KNN(k=group_size, transpose_mode=False)(point_cloud, query_cloud)


def group_points(group_size, point_cloud, query_cloud, point_features=None):
    """search_knn.py:23-39 -> (grouped_points (B,3,Q,k), grouped_features (B,F,Q,k) | None, idx (B,Q,k) int32)."""
    _, idx = knn_point(group_size, point_cloud, query_cloud)
    idx = idx.permute(0, 2, 1).type(torch.int32).contiguous()
    grouped_points = group_point(point_cloud.contiguous(), idx)
    grouped_features = None if point_features is None else group_point(point_features.contiguous(), idx)
    return grouped_points, grouped_features, idx


class SoftProjection(nn.Module):
    """search_knn.py:44-174 (soft nearest-neighbour projection / feature propagation)."""

    def __init__(self, group_size, initial_temperature=1.0, is_temperature_trainable=True, min_sigma=1e-4):
        super().__init__()
        self._group_size = group_size
        self._temperature = nn.Parameter(torch.tensor(initial_temperature, dtype=torch.float32),
                                         requires_grad=is_temperature_trainable)
        self._min_sigma = torch.tensor(min_sigma, dtype=torch.float32)

    def forward(self, point_cloud, query_cloud, point_features=None, action="project"):
        point_cloud, query_cloud = point_cloud.contiguous(), query_cloud.contiguous()
        if action == "project":
            return self.project(point_cloud, query_cloud)
        if action == "propagate":
            return self.propagate(point_cloud, point_features, query_cloud)
        if action == "project_and_propagate":
            return self.project_and_propagate(point_cloud, point_features, query_cloud)
        raise ValueError("action should be one of the following: 'project', 'propagate', 'project_and_propagate'")

    def _group_points(self, point_cloud, query_cloud, point_features=None):
        gp, gf, _ = group_points(self._group_size, point_cloud, query_cloud, point_features)
        return gp, gf

    def sigma(self):
        return torch.max(self._temperature ** 2, self._min_sigma.to(self._temperature.device))

    def _weights(self, grouped_points, query_cloud, sigma=None):
        deltas = grouped_points - query_cloud.unsqueeze(-1).expand_as(grouped_points)
        dist = torch.sum(deltas ** 2, dim=1, keepdim=True) / (self.sigma() if sigma is None else sigma)
        return torch.softmax(-dist, dim=3)

    def project_and_propagate(self, point_cloud, point_features, query_cloud):
        gp, gf = self._group_points(point_cloud, query_cloud, point_features)
        w = self._weights(gp, query_cloud)
        return torch.sum(gp * w, dim=3), torch.sum(gf * w, dim=3)

    def propagate(self, point_cloud, point_features, query_cloud):
        gp, gf = self._group_points(point_cloud, query_cloud, point_features)
        return torch.sum(gf * self._weights(gp, query_cloud), dim=3)

    def project(self, point_cloud, query_cloud, hard=False):
        if hard:
            raise NotImplementedError
        gp, _ = self._group_points(point_cloud, query_cloud)
        return torch.sum(gp * self._weights(gp, query_cloud), dim=3)
