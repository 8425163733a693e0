"""Callers on either side of the SoftGroup ops in the reference model (models/dgcnn-hais-concat-direct-4.py,
"M4"): forward_grouping (M4:1123-1295), compute_batch_adjacency_matrix (M4:210-233), clusters_voxelization
(M4:1300-1355), global_pool (M4:1358-1370), get_batch_offsets (M4:1372-1377) -- same argument meaning and
results, built on the drop-in ops of gcanet_amd.softgroup.ops.

Differences that do not change results:
  * get_batch_offsets is one bincount + cumsum (the reference loops over the batch with a device sync per item);
  * no torch.cuda.empty_cache() calls (M4:1189,1250);
  * clusters_voxelization returns the sparse-tensor FIELDS (features, int32 (M,4) coords [cluster,x,y,z],
    spatial_shape, batch_size) instead of an spconv.SparseConvTensor -- spconv is an un-vendored third-party
    package; the tuple is the stable on-wire format for any sparse-conv backend.
Reference quirk kept: proposals index points WITHIN their cloud (object_idxs are per-cloud positions,
M4:1270), although the features they are applied to are flattened over the batch.
"""
import ctypes as C

import torch

from . import _lib
from .softgroup.ops import (ball_query, global_avg_pool, hierarchical_aggregation, sec_max, sec_min, voxelization,
                            voxelization_idx)


def compute_batch_adjacency_matrix(batch_point_clouds, radius=0, dist_state=True, sigma=1.0):
    """M4:210-233: cdist -> zero diagonal -> GLOBAL min/max normalisation -> Gaussian -> zero diagonal."""
    distances = torch.cdist(batch_point_clouds, batch_point_clouds)
    adjacency = distances if dist_state else (distances <= radius).float()
    adjacency = adjacency - torch.diag_embed(torch.diagonal(adjacency, dim1=-2, dim2=-1))
    mn, mx = adjacency.min(), adjacency.max()
    adjacency = (adjacency - mn) / (mx - mn)
    adjacency = torch.exp(-adjacency ** 2 / (2 * sigma ** 2))
    return adjacency - torch.diag_embed(torch.diagonal(adjacency, dim1=-2, dim2=-1))


def _counts(idx, n):
    """Occurrences of 0..n-1 in idx (other values must not occur).  torch.bincount reads the maximum back to size its
    result -- a host synchronisation per call, three per literal training step -- the callers here know n."""
    idx = idx.reshape(-1).long()
    return torch.zeros(n, dtype=torch.int64, device=idx.device).scatter_add_(0, idx, torch.ones_like(idx))


def get_batch_offsets(batch_idxs, bs):
    """M4:1372-1377 -> int32 (bs+1) on the device of batch_idxs."""
    counts = _counts(batch_idxs, bs)
    return torch.cat([counts.new_zeros(1), torch.cumsum(counts, 0)]).int()


def forward_grouping(semantic_scores, pt_offsets, batch_idxs, coords_float, type_per_point, param_per_point,
                     feature_per_point, semantic_classes=10, training_mode='train', using_set_aggr=False,
                     radius=0.03, similarity_threshold_inst=0.989, similarity_threshold_para=0.0, mean_active=300,
                     min_npoint=50):
    """M4:1123-1295.  semantic_scores (B*N,P), pt_offsets (B*N,3), batch_idxs (B*N), coords_float (B*N,3),
    type_per_point (B,N,P) [shape donor], param_per_point (B,N,22), feature_per_point (B,N,emb).
    Returns (proposals_idx (S,2) int32 CPU [cluster id, point index within its cloud], proposals_offset int32 CPU)."""
    B, N = type_per_point.shape[0], type_per_point.shape[1]
    batch_size = int(batch_idxs.max()) + 1
    semantic_scores = semantic_scores.softmax(dim=-1).view(B, N, -1)
    coords_float = coords_float.view(B, N, -1)
    pt_offsets = pt_offsets.view(B, N, -1)
    batch_idxs = batch_idxs.view(B, N, -1)
    proposals_idx_list, proposals_offset_list = [], []
    for b in range(batch_size):
        labels = semantic_scores[b].argmax(dim=1)
        for class_id in range(semantic_classes):
            object_idxs = (labels == class_id).nonzero().view(-1)
            if object_idxs.size(0) < min_npoint:
                continue
            batch_idxs_ = batch_idxs[b][object_idxs].reshape(-1).int().contiguous()
            shifted = (coords_float[b][object_idxs] + pt_offsets[b][object_idxs]).float().contiguous()
            batch_offsets_ = get_batch_offsets(batch_idxs_, batch_size)
            adj_inst = compute_batch_adjacency_matrix(feature_per_point[b][object_idxs].float().unsqueeze(0)).squeeze(0)
            adj_para = compute_batch_adjacency_matrix(param_per_point[b][object_idxs].float().unsqueeze(0)).squeeze(0)
            neighbor_inds, start_len = ball_query(shifted, batch_idxs_, batch_offsets_, adj_inst.contiguous(),
                                                  similarity_threshold_inst, adj_para.contiguous(),
                                                  similarity_threshold_para, radius, mean_active)
            semantic_preds_cpu = torch.full((object_idxs.numel(),), class_id, dtype=torch.int32)
            proposals_idx, proposals_offset = hierarchical_aggregation(
                semantic_preds_cpu, shifted.cpu(), neighbor_inds.cpu(), start_len.cpu(), batch_idxs_.cpu(),
                training_mode, using_set_aggr if training_mode != 'train' else False)
            proposals_idx[:, 1] = object_idxs.cpu()[proposals_idx[:, 1].long()].int()
            if len(proposals_offset_list) > 0:          # merge proposals (M4:1273-1276)
                proposals_idx[:, 0] += sum(x.size(0) for x in proposals_offset_list) - 1
                proposals_offset = (proposals_offset + proposals_offset_list[-1][-1])[1:]
            if proposals_idx.size(0) > 0:
                proposals_idx_list.append(proposals_idx)
                proposals_offset_list.append(proposals_offset)
    if proposals_idx_list:
        return torch.cat(proposals_idx_list, dim=0), torch.cat(proposals_offset_list)
    return torch.zeros((0, 2), dtype=torch.int32), torch.zeros((0,), dtype=torch.int32)


def _pad16(x):
    """Rows zero-padded to a multiple of 16 columns (zeros leave every distance unchanged)."""
    c = x.shape[1]
    return x.contiguous() if c % 16 == 0 else torch.nn.functional.pad(x, (0, 16 - c % 16)).contiguous()


def forward_grouping_device(semantic_scores, pt_offsets, batch_idxs, coords_float, type_per_point, param_per_point,
                            feature_per_point, semantic_classes=10, training_mode='train', using_set_aggr=False,
                            radius=0.03, similarity_threshold_inst=0.989, similarity_threshold_para=0.0, mean_active=300,
                            min_npoint=50, to_cpu=True):
    """forward_grouping (M4:1123-1295) with every (cloud, class) subset handled at once on the device
    (csrc/softgroup.hip: seg_diameter / ballquery_sim kernels, csrc/cluster_dev.hip):

      * points are sorted by cloud*P + class (stable, so a subset keeps the ascending order of `nonzero()`),
        subsets below `min_npoint` are switched off instead of skipped by a Python `continue`;
      * the two (n,n) similarity matrices of compute_batch_adjacency_matrix are never formed: only their global
        maximum is needed (the per-subset diameter, one MFMA Gram pass) and the exponentials are evaluated at the
        pairs inside the search radius -- the same predicate `adj_inst > thr_inst and adj_para > thr_para`;
      * every point reserves its neighbour list in an n*mean_active buffer with one atomic (a single pass, retried
        with the exact size if it did not fit -- the reference's meanActive retry loop, functions.py:460-474);
      * connected components + the reference's BFS member order + the kept/primary merge run on the device;
        ONE host synchronisation in total (the result sizes) instead of several per subset.

    Same return value as forward_grouping (CPU int32 tensors; device tensors with to_cpu=False).  Pairs whose
    similarity lies within float rounding of a threshold can fall on the other side than with the dense path, which
    takes its distances from torch.cdist's matmul form; the reference's own result moves with the torch version there.
    Set aggregation (evaluation: `training_mode != 'train' and using_set_aggr`, hierarchical_aggregation.cu:22-196) runs
    on the device as well (csrc/cluster_dev.hip: gcn_set_aggregation).  If a neighbour list hits the 3000 cap
    (bfs_cluster.cu:54: lists stop being symmetric) the literal path is taken."""
    set_aggr = training_mode != 'train' and bool(using_set_aggr)
    _lib.require_cuda(semantic_scores, pt_offsets, coords_float, param_per_point, feature_per_point)
    B, N = type_per_point.shape[0], type_per_point.shape[1]
    P = int(semantic_classes)
    assert 1 <= P <= 10, "hierarchical_aggregation.cpp:7-8 knows 10 classes"
    dev = semantic_scores.device
    n, S = B * N, B * P
    labels = semantic_scores.softmax(dim=-1).view(n, -1).argmax(dim=1)
    seg_key = torch.arange(B, device=dev).repeat_interleave(N) * P + labels
    seg_sorted, order = torch.sort(seg_key, stable=True)
    counts = _counts(seg_sorted, S)
    seg_offsets = torch.cat([counts.new_zeros(1), counts.cumsum(0)]).int()
    seg_cls = torch.where(counts >= min_npoint, torch.arange(S, device=dev) % P, torch.full_like(counts, -1)).int()
    seg_of = seg_sorted.int()
    shifted = (coords_float.view(n, -1).float() + pt_offsets.view(n, -1).float())[order].contiguous()
    fi = _pad16(feature_per_point.reshape(n, -1).float()[order])
    fp = _pad16(param_per_point.reshape(n, -1).float()[order])
    point_index = (order % N).int()

    lib = _lib.lib()
    st = _lib.stream_of(shifted)
    dm = torch.empty(2, S, dtype=torch.float32, device=dev)
    with _lib.on_device(shifted):
        for f, d in ((fi, dm[0]), (fp, dm[1])):
            nb = lib.gcn_segment_diameter2_ws_bytes(n, f.shape[1], S)
            if 0 <= nb <= (4 << 30):     # bf16 tile prefilter + exact recheck: the same bits as the exhaustive kernel
                dws = torch.empty(nb, dtype=torch.uint8, device=dev)
                _lib.call("gcn_segment_diameter2_filtered", n, f.shape[1], _lib.ptr(f), _lib.ptr(seg_offsets),
                          _lib.ptr(seg_cls), S, _lib.ptr(dws), _lib.ptr(d), st)
            else:                        # > 1 M rows: the tile-pair table would not pay; exhaustive kernel
                xx = torch.empty(n, dtype=torch.float32, device=dev)
                tiles = torch.empty(S + 1, dtype=torch.int32, device=dev)
                _lib.call("gcn_segment_diameter2", n, f.shape[1], _lib.ptr(f), _lib.ptr(seg_offsets), _lib.ptr(seg_cls),
                          S, _lib.ptr(xx), _lib.ptr(tiles), _lib.ptr(d), st)
        grid_ws = torch.empty(lib.gcn_ballquery_sim_ws_bytes(n), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.gcn_cluster_components_ws_bytes(n), dtype=torch.uint8, device=dev)
        start_len = torch.empty(n, 2, dtype=torch.int32, device=dev)
        cluster_idxs = torch.empty(n, 2, dtype=torch.int32, device=dev)
        cluster_offsets = torch.empty(n + 1, dtype=torch.int32, device=dev)
        status = torch.empty(8, dtype=torch.int32, device=dev)     # [0:4] ball query, [4:6] (rows, clusters)
        capacity = n * int(mean_active)
        while True:
            nbr = torch.empty(max(capacity, 1), dtype=torch.int32, device=dev)
            _lib.call("gcn_ballquery_sim", n, float(radius), _lib.ptr(shifted), _lib.ptr(seg_of), _lib.ptr(seg_offsets),
                      _lib.ptr(seg_cls), S, _lib.ptr(fi), fi.shape[1], _lib.ptr(dm[0]), float(similarity_threshold_inst),
                      _lib.ptr(fp), fp.shape[1], _lib.ptr(dm[1]), float(similarity_threshold_para), _lib.ptr(nbr),
                      capacity, _lib.ptr(start_len), _lib.ptr(status), _lib.ptr(grid_ws), st)
            if not set_aggr:
                _lib.call("gcn_cluster_components_clobber", n, _lib.ptr(nbr), _lib.ptr(start_len), _lib.ptr(seg_of),
                          _lib.ptr(seg_offsets), _lib.ptr(seg_cls), S, _lib.ptr(point_index), -1.0, _lib.ptr(ws),
                          _lib.ptr(cluster_idxs), _lib.ptr(cluster_offsets), _lib.ptr(status[4:]), st)
            else:
                # every component (dropped fragments too) in sorted positions, then the absorption pass
                ident = torch.arange(n, dtype=torch.int32, device=dev)
                all_idxs = torch.empty(n, 2, dtype=torch.int32, device=dev)
                all_offs = torch.empty(n + 1, dtype=torch.int32, device=dev)
                all_counts = torch.empty(2, dtype=torch.int32, device=dev)
                _lib.call("gcn_cluster_components_clobber", n, _lib.ptr(nbr), _lib.ptr(start_len), _lib.ptr(seg_of),
                          _lib.ptr(seg_offsets), _lib.ptr(seg_cls), S, _lib.ptr(ident), -2.0, _lib.ptr(ws),
                          _lib.ptr(all_idxs), _lib.ptr(all_offs), _lib.ptr(all_counts), st)
                cluster_idxs = torch.empty(2 * n, 2, dtype=torch.int32, device=dev)      # absorbed points appear twice
                sa_ws = torch.empty(lib.gcn_set_aggregation_ws_bytes(n, S), dtype=torch.uint8, device=dev)
                _lib.call("gcn_set_aggregation", n, S, _lib.ptr(all_counts), _lib.ptr(all_idxs), _lib.ptr(all_offs),
                          _lib.ptr(seg_of), _lib.ptr(seg_cls), _lib.ptr(shifted), _lib.ptr(point_index), _lib.ptr(sa_ws),
                          _lib.ptr(cluster_idxs), _lib.ptr(cluster_offsets), _lib.ptr(status[4:]), st)
            total, capped, overflow, _, nsum, ncl = status.cpu().tolist()[:6]       # the one host synchronisation
            if not overflow:
                break
            capacity = total                # lists did not fit (the reference's meanActive retry, functions.py:460-474)
    if capped:
        return forward_grouping(semantic_scores, pt_offsets, batch_idxs, coords_float, type_per_point,
                                param_per_point, feature_per_point, semantic_classes, training_mode, using_set_aggr,
                                radius, similarity_threshold_inst, similarity_threshold_para, mean_active, min_npoint)
    if ncl == 0:
        z = torch.zeros((0, 2), dtype=torch.int32), torch.zeros((0,), dtype=torch.int32)
        return z if to_cpu else (z[0].to(dev), z[1].to(dev))
    pi, po = cluster_idxs[:nsum], cluster_offsets[:ncl + 1]
    return (pi.cpu(), po.cpu()) if to_cpu else (pi, po)


def clusters_voxelization(clusters_idx, clusters_offset, feats, coords, scale, spatial_shape, rand_quantize=False,
                          rand=None, inp_map_on_device=False):
    """M4:1300-1355.  clusters_idx (S,2) int32 CPU, clusters_offset (P+1) int32 CPU, feats (M,C) cuda, coords (M,3)
    cuda.  Returns (voxel_feats (V,C) cuda, voxel_coords (V,4) int32 cuda, [spatial_shape]*3, batch_size, inp_map).
    `rand` (two (3,) tensors) replaces the reference's torch.rand(3) draws to make tests reproducible.
    inp_map_on_device: keep inp_map where it was computed (the reference returns it on the CPU, M4:1352, and its caller
    moves it straight back, M4:771: a blocking round trip per step)."""
    dev = feats.device
    if clusters_idx.size(0) == 0:
        c = torch.tensor([[0, 0, 0, 0], [0, spatial_shape - 1, spatial_shape - 1, spatial_shape - 1]], dtype=torch.int,
                         device=dev)
        return feats[0:2], c, [spatial_shape] * 3, 1, feats.new_zeros((1,), dtype=torch.long)
    batch_idx = clusters_idx[:, 0].to(dev).long()
    c_idxs = clusters_idx[:, 1].to(dev).long()
    feats = feats.index_select(0, c_idxs).float().contiguous()      # backward = one index_add (advanced indexing: ~250 us)
    coords = coords.index_select(0, c_idxs).float().contiguous()
    offs = clusters_offset.to(dev).int().contiguous()
    coords_min = sec_min(coords, offs)
    coords_max = sec_max(coords, offs)
    clusters_scale = 1 / ((coords_max - coords_min) / spatial_shape).max(1)[0] - 0.01
    clusters_scale = torch.clamp(clusters_scale, min=None, max=scale)
    coords_min = coords_min * clusters_scale[:, None]
    coords_max = coords_max * clusters_scale[:, None]
    coords = coords * clusters_scale[batch_idx][:, None]
    if rand_quantize:
        r1, r2 = rand if rand is not None else (torch.rand(3, device=dev), torch.rand(3, device=dev))
        rng = coords_max - coords_min
        coords_min = coords_min - torch.clamp(spatial_shape - rng - 0.001, min=0) * r1.to(dev)
        coords_min = coords_min - torch.clamp(spatial_shape - rng + 0.001, max=0) * r2.to(dev)
    coords = coords - coords_min[batch_idx]
    if coords.is_cuda:                       # M4:1343's assert, checked on the device: no host synchronisation
        torch._assert_async(((coords >= 0) & (coords < spatial_shape)).all())
    else:
        assert coords.shape.numel() == int(((coords >= 0) * (coords < spatial_shape)).sum())
    nb = int(clusters_offset.shape[0]) - 1   # proposals are numbered 0..P-1 in order: == clusters_idx[-1, 0] + 1, no read-back
    if nb <= 65536 and spatial_shape <= 65536:
        # device voxelize_idx (csrc/voxelize_dev.hip): no .cpu() round trip of the coordinates; inp_map is returned
        # on the CPU as the reference does (M4:1352)
        coords = torch.cat([batch_idx.view(-1, 1), coords.long()], 1).contiguous()
        out_coords, inp_map, out_map = voxelization_idx(coords, nb)
        out_feats = voxelization(feats, out_map)
        return out_feats, out_coords.int(), [spatial_shape] * 3, nb, (inp_map if inp_map_on_device else inp_map.cpu())
    coords = torch.cat([clusters_idx[:, 0].view(-1, 1).long(), coords.long().cpu()], 1).contiguous()
    out_coords, inp_map, out_map = voxelization_idx(coords, nb)
    out_feats = voxelization(feats, out_map.to(dev))
    return out_feats, out_coords.int().to(dev), [spatial_shape] * 3, nb, inp_map


def global_pool(features, indices, batch_size=None):
    """M4:1358-1370 (expand=False): per-sample mean of sparse-tensor features; indices = first coord column.
    batch_size: number of samples when the caller knows it (no read-back of max(indices))."""
    batch_counts = torch.bincount(indices.long()) if batch_size is None else _counts(indices, batch_size)
    batch_offset = torch.cat([batch_counts.new_zeros(1), torch.cumsum(batch_counts, 0)]).int()
    return global_avg_pool(features.float().contiguous(), batch_offset)
