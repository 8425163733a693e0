"""Host-side mirror of the reference model's hot path
(models/dgcnn-hais-concat-direct-4.py, "M4"): same function names, argument meaning and
tensor layouts, with the heavy lifting done by libgcanet_hip.so.
"""
import numpy as np
import torch

from . import _lib


def _knn_model(x, k1, k2, metric):
    if x.dim() != 3:
        raise RuntimeError("knn: x must be (B, C, N)")
    _lib.require_cuda(x)
    x = x.float().contiguous()
    B, C, N = x.shape
    step = k2 // k1
    kout = len(range(0, k2, step))
    idx = torch.empty(B, N, kout, dtype=torch.int64, device=x.device)
    xx = torch.empty(B, N, dtype=torch.float32, device=x.device)
    with torch.cuda.device_of(x):
        _lib.call("gcn_knn_model", _lib.ptr(x), B, C, N, k1, k2, metric, _lib.ptr(idx), None, _lib.ptr(xx),
                  _lib.stream_of(x))
    return idx


def knn(x, k1, k2):
    """M4:30-47: x (B,C,N) -> idx (B,N,k1) int64: the k2 nearest in feature space (self included),
    every (k2//k1)-th kept.  One fused kernel per call for the whole batch; the reference loops over the
    batch in Python and materialises N x N per cloud.  Ties -> lowest index."""
    with torch.no_grad():
        return _knn_model(x, k1, k2, 0)


def knn_points_normals(x, k1, k2):
    """M4:50-90: metric |p_i-p_j|^2 * (1 + (2 - 2 n_i.n_j)) on x = [xyz; normal] (B,6,N)."""
    with torch.no_grad():
        return _knn_model(x, k1, k2, 1)
