"""Drop-in for ``knn_cuda`` (reference: models/KNN_CUDA/knn_cuda/__init__.py).

``KNN(k, transpose_mode).forward(ref, query) -> (D, I)`` with D the L2 distance (sqrt
applied, knn.cu:178-183) and I int64 0-based indices, ties resolved to the lowest
reference index (stable insertion sort, knn.cu:125-131).  One fused HIP kernel per call
(csrc/knn.hip) instead of the reference's per-batch Python loop over three kernels and a
materialised (nr, nq) distance matrix (knn.cpp:36).
"""
import torch
import torch.nn as nn

from . import _lib

__version__ = "0.2"


def _knn_batched(ref, query, k, point_major):
    if ref.dim() != 3 or query.dim() != 3:
        raise RuntimeError("knn: ref and query must be 3-D (batch first)")
    _lib.require_cuda(ref, query)
    ref = ref.float().contiguous()
    query = query.float().contiguous()
    B = ref.size(0)
    if point_major:
        nr, dim = ref.size(1), ref.size(2)
        nq = query.size(1)
        if query.size(2) != dim:
            raise RuntimeError("knn: ref and query dimensionality differ")
        shape = (B, nq, k)
    else:
        dim, nr = ref.size(1), ref.size(2)
        nq = query.size(2)
        if query.size(1) != dim:
            raise RuntimeError("knn: ref and query dimensionality differ")
        shape = (B, k, nq)
    D = torch.empty(shape, dtype=torch.float32, device=ref.device)
    I = torch.empty(shape, dtype=torch.int64, device=ref.device)
    tile_ws = None
    if dim == 3 and k <= 64 and nr >= 512 and nr == nq and ref.data_ptr() == query.data_ptr():
        # a 3-D cloud against itself: Morton-tiled kernel with box pruning (identical results)
        tile_ws = torch.empty(_lib.lib().gcn_knn_tiles_ws_bytes(B, 3, nr), dtype=torch.uint8, device=ref.device)
    with _lib.on_device(ref):
        _lib.call("gcn_knn_cuda", _lib.ptr(ref), _lib.ptr(query), B, dim, nr, nq, k, int(point_major),
                  _lib.ptr(D), _lib.ptr(I), _lib.ptr(tile_ws), _lib.stream_of(ref))
    return D, I


def knn(ref, query, k):
    """ref (dim, nr), query (dim, nq) -> d (k, nq), i (k, nq)  (KNN/__init__.py:41-44)."""
    d, i = _knn_batched(ref.unsqueeze(0), query.unsqueeze(0), k, False)
    return d[0], i[0]


class KNN(nn.Module):
    """KNN/__init__.py:54-74."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        with torch.no_grad():
            return _knn_batched(ref, query, self.k, self._t)
