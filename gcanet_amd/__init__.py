"""gcanet_amd -- MI355X-native (gfx950) implementation of GCANet's per-point
feature-aggregation hot path behind the reference's own op signatures.

Sub-modules mirror the reference's boundary layer:
  gcanet_amd.knn_cuda                      <-> models/KNN_CUDA/knn_cuda
  gcanet_amd.pointnet2_ops.pointnet2_utils <-> .../pointnet2_ops/pointnet2_utils.py
  gcanet_amd.softgroup.ops                 <-> softgroup/ops
  gcanet_amd.search_knn                    <-> models/search_knn.py
  gcanet_amd.dgcnn                         <-> models/dgcnn-hais-concat-direct-4.py (hot path)
All compute goes through libgcanet_hip.so (include/gcanet_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
