"""Drop-in for ``softgroup.ops`` (reference: softgroup/ops/functions.py).

Same callables, argument order, dtypes and device conventions: the GPU ops take CUDA
tensors; ``voxelization_idx``, ``bfs_cluster`` and ``hierarchical_aggregation`` take and
return CPU tensors exactly like the reference (voxelize.cpp / bfs_cluster.cpp /
hierarchical_aggregation.cpp are host C++ there too).  Variable-size outputs use the
C ABI's two-call protocol instead of ``resize_()`` on empty tensors.
"""
import ctypes as C

import torch
from torch.autograd import Function

from ... import _lib

__all__ = ["hierarchical_aggregation", "ball_query", "ball_query_easy", "octree_ball_query",
           "get_mask_iou_on_cluster", "get_mask_iou_on_pred", "get_mask_label", "voxelization_idx",
           "voxelization", "ballquery_batch_p", "ballquery_batch_p_easy", "bfs_cluster", "global_avg_pool",
           "sec_mean", "sec_min", "sec_max"]


def _run(name, like, *args):
    with _lib.on_device(like):
        _lib.call(name, *args, _lib.stream_of(like))


def _cpu_i32(t):
    assert not t.is_cuda, "expected a CPU tensor (the reference passes .cpu() tensors here)"
    return t.to(torch.int32).contiguous()


# ------------------------------------------------------------------ clustering (host)
class HierarchicalAggregation(Function):
    @staticmethod
    def forward(ctx, semantic_label, coord_shift, ball_query_idxs, start_len, batch_idxs, training_mode,
                using_set_aggr):
        """functions.py:7-72.  CPU int32/float32 tensors -> (cluster_idxs (S,2) i32, cluster_offsets i32), CPU."""
        assert semantic_label.is_contiguous()
        assert coord_shift.is_contiguous()
        assert ball_query_idxs.is_contiguous()
        assert start_len.is_contiguous()
        sem, bq, sl, bi = map(_cpu_i32, (semantic_label, ball_query_idxs, start_len, batch_idxs))
        cs = coord_shift.to(torch.float32).contiguous()
        N = sl.size(0)
        idxs = torch.zeros(max(2 * N, 1), 2, dtype=torch.int32)
        offs = torch.zeros(N + 1, dtype=torch.int32)
        s, c = C.c_int(0), C.c_int(0)
        _lib.call("gcn_hierarchical_aggregation_host", _lib.ptr(sem), _lib.ptr(cs), _lib.ptr(bi), _lib.ptr(bq),
                  _lib.ptr(sl), N, int(bool(using_set_aggr)), _lib.ptr(idxs), _lib.ptr(offs),
                  C.addressof(s), C.addressof(c))
        return idxs[:s.value].clone(), offs[:c.value + 1].clone()

    @staticmethod
    def backward(ctx, a=None):
        return None


hierarchical_aggregation = HierarchicalAggregation.apply


class BFSCluster(Function):
    @staticmethod
    def forward(ctx, cluster_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
        """functions.py:599-626."""
        assert cluster_numpoint_mean.is_contiguous()
        assert ball_query_idxs.is_contiguous()
        assert start_len.is_contiguous()
        cm = cluster_numpoint_mean.to(torch.float32).contiguous()
        if ball_query_idxs.is_cuda and start_len.is_cuda and 0 < start_len.size(0) < (1 << 20):
            # device path (csrc/cluster_dev.hip: union-find components + the reference's BFS member order) for complete,
            # i.e. symmetric, neighbour lists; a list that hit a cap (1000 easy / 3000) may be truncated -> host BFS below
            sl32, bq32 = start_len.int().contiguous(), ball_query_idxs.int().contiguous()
            n = sl32.size(0)
            mean = float(cm.reshape(-1)[int(class_id)])
            thr = float(threshold) if mean == -1 else float(threshold) * mean          # bfs_cluster.cpp:86-91
            if int(sl32[:, 1].max()) < 1000 and thr >= 0:
                dev = sl32.device
                z = torch.zeros(n, dtype=torch.int32, device=dev)
                seg_offsets = torch.tensor([0, n], dtype=torch.int32, device=dev)
                seg_cls = torch.zeros(1, dtype=torch.int32, device=dev)
                pidx = torch.arange(n, dtype=torch.int32, device=dev)
                ws = torch.empty(_lib.lib().gcn_cluster_components_ws_bytes(n), dtype=torch.uint8, device=dev)
                idxs = torch.empty(n, 2, dtype=torch.int32, device=dev)
                offs = torch.empty(n + 1, dtype=torch.int32, device=dev)
                counts = torch.empty(2, dtype=torch.int32, device=dev)
                _run("gcn_cluster_components", sl32, n, _lib.ptr(bq32), _lib.ptr(sl32), _lib.ptr(z), _lib.ptr(seg_offsets),
                     _lib.ptr(seg_cls), 1, _lib.ptr(pidx), thr, _lib.ptr(ws), _lib.ptr(idxs), _lib.ptr(offs), _lib.ptr(counts))
                nsum, ncl = counts.tolist()
                return idxs[:nsum].cpu(), offs[:ncl + 1].cpu()
        bq, sl = _cpu_i32(ball_query_idxs), _cpu_i32(start_len)
        N = sl.size(0)
        s, c = C.c_int(0), C.c_int(0)
        args = (_lib.ptr(cm), _lib.ptr(bq), _lib.ptr(sl), N, float(threshold), int(class_id),
                C.addressof(s), C.addressof(c))
        _lib.call("gcn_bfs_cluster_host", *args, None, None)
        idxs = torch.zeros(s.value, 2, dtype=torch.int32)
        offs = torch.zeros(c.value + 1, dtype=torch.int32)
        _lib.call("gcn_bfs_cluster_host", *args, _lib.ptr(idxs), _lib.ptr(offs))
        return idxs, offs

    @staticmethod
    def backward(ctx, a=None):
        return None


bfs_cluster = BFSCluster.apply


# ------------------------------------------------------------------ ball query (GPU)
def _ballquery(coords, batch_idxs, batch_offsets, radius, meanActive, adj_inst=None, thr_inst=0.0,
               adj_para=None, thr_para=0.0):
    n = coords.size(0)
    for t in (coords, batch_idxs, batch_offsets) + ((adj_inst, adj_para) if adj_inst is not None else ()):
        assert t.is_contiguous() and t.is_cuda
    dev = coords.device
    count_ws = torch.empty(n + 1, dtype=torch.int32, device=dev)
    grid_ws = None
    if adj_inst is None and n >= 2048:       # uniform-grid candidate search (easy form)
        grid_ws = torch.empty(_lib.lib().gcn_ballquery_grid_ws_bytes(n), dtype=torch.uint8, device=dev)
    nbatch = int(batch_offsets.numel()) - 1
    total = C.c_int(0)
    while True:  # functions.py:460-474 retry loop, kept verbatim in behaviour
        idx = torch.zeros(n * meanActive, dtype=torch.int32, device=dev)
        start_len = torch.zeros(n, 2, dtype=torch.int32, device=dev)
        _run("gcn_ballquery_batch_p", coords, n, int(meanActive), float(radius), _lib.ptr(coords),
             _lib.ptr(batch_idxs), _lib.ptr(batch_offsets), _lib.ptr(adj_inst), float(thr_inst),
             _lib.ptr(adj_para), float(thr_para), _lib.ptr(idx), _lib.ptr(start_len), _lib.ptr(count_ws),
             nbatch, _lib.ptr(grid_ws), C.addressof(total))
        nActive = total.value
        if nActive <= n * meanActive:
            break
        meanActive = int(nActive // n + 1)
    return idx[:nActive], start_len


class BallQueryBatchP(Function):
    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, adj_mat_inst, similarity_threshold_inst, adj_mat_para,
                similarity_threshold_para, radius, meanActive):
        """functions.py:434-478: coords (n,3) f32, batch_idxs (n) i32, batch_offsets (B+1) i32, adj (n,n) f32."""
        return _ballquery(coords, batch_idxs, batch_offsets, radius, meanActive, adj_mat_inst,
                          similarity_threshold_inst, adj_mat_para, similarity_threshold_para)

    @staticmethod
    def backward(ctx, a=None, b=None):
        return (None,) * 9


ballquery_batch_p = BallQueryBatchP.apply


class BallQueryBatchP_Easy(Function):
    @staticmethod
    def forward(ctx, coords, batch_idxs, batch_offsets, radius, meanActive):
        """functions.py:499-538."""
        return _ballquery(coords, batch_idxs, batch_offsets, radius, meanActive)

    @staticmethod
    def backward(ctx, a=None, b=None):
        return (None,) * 5


ballquery_batch_p_easy = BallQueryBatchP_Easy.apply


def _octree_leaf(coords):
    """Breadth-first leaf number (0..511) of every point in the reference's fixed 3-level octree over the cloud's
    bounding box (octree_ball_query.cpp:19-108: octant bit = coordinate >= box centre, child centre = centre +- w/4),
    in the same f32 operations."""
    mx, mn = coords.max(0)[0], coords.min(0)[0]
    c = ((mx + mn) / 2).unsqueeze(0).expand_as(coords).clone()
    w = (mx - mn).unsqueeze(0).expand_as(coords).clone()
    node = torch.zeros(coords.shape[0], dtype=torch.int64, device=coords.device)
    for _ in range(3):
        bit = ~(coords < c)                                   # (n,3) x,y,z
        node = node * 8 + (bit[:, 2].long() << 2) + (bit[:, 1].long() << 1) + bit[:, 0].long() + 1
        w = w / 2
        c = torch.where(bit, c + w / 2, c - w / 2)
    return node - 73


def octree_ball_query(coords, mean_active, radius):
    """functions.py:127-157.  The reference builds a fixed 3-level octree on the host to prune the pair tests
    (octree_ball_query.cpp:19-165) and lists, per point, the in-radius points of every active leaf in breadth-first
    leaf order, ascending index inside a leaf, capped at 1000 (octree_ball_query.cu:56-126).  Here the candidate
    search is the uniform-grid radius query of ball_query_easy (same neighbour sets) and the lists are put into the
    reference's order by one segmented sort on (point, leaf, index).  A point with more than 1000 neighbours keeps
    its 1000 lowest-index ones instead of the first 1000 in leaf order."""
    coords = coords.cuda().contiguous().float()
    n = coords.size(0)
    batch_idxs = torch.zeros(n, dtype=torch.int32, device=coords.device)
    batch_offsets = torch.tensor([0, n], dtype=torch.int32, device=coords.device)
    idx, start_len = ballquery_batch_p_easy(coords, batch_idxs, batch_offsets, radius, mean_active)
    if idx.numel() == 0:
        return idx, start_len
    leaf = _octree_leaf(coords)
    lens = start_len[:, 1].long()
    seg = torch.repeat_interleave(torch.arange(n, device=coords.device), lens)      # CSR segments are in point order
    j = idx.long()
    order = torch.argsort((seg * 512 + leaf[j]) * n + j)
    return idx[order].contiguous(), start_len


def ball_query(coords, batch_idxs, batch_offsets, adj_mat_inst, similarity_threshold_inst, adj_mat_para,
               similarity_threshold_para, radius, mean_active, with_octree=False):
    """functions.py:93-102."""
    if with_octree:
        return octree_ball_query(coords, mean_active, radius)
    return ballquery_batch_p(coords, batch_idxs, batch_offsets, adj_mat_inst, similarity_threshold_inst,
                             adj_mat_para, similarity_threshold_para, radius, mean_active)


def ball_query_easy(coords, batch_idxs, batch_offsets, radius, mean_active, with_octree=False):
    """functions.py:106-108."""
    return ballquery_batch_p_easy(coords, batch_idxs, batch_offsets, radius, mean_active)


# ------------------------------------------------------------------ IoU / mask labels (GPU)
class GetMaskIoUOnCluster(Function):
    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum):
        """functions.py:160-189."""
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        for t in (proposals_idx, proposals_offset, instance_labels, instance_pointnum):
            assert t.is_contiguous() and t.is_cuda
        iou = torch.zeros(nProposal, nInstance, dtype=torch.float32, device=proposals_idx.device)
        _run("gcn_get_mask_iou", proposals_idx, nInstance, nProposal, _lib.ptr(proposals_idx),
             _lib.ptr(proposals_offset), _lib.ptr(instance_labels), _lib.ptr(instance_pointnum), None, _lib.ptr(iou))
        return iou

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


get_mask_iou_on_cluster = GetMaskIoUOnCluster.apply


class GetMaskIoUOnPred(Function):
    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid):
        """functions.py:195-229."""
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        for t in (proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid):
            assert t.is_contiguous() and t.is_cuda
        iou = torch.zeros(nProposal, nInstance, dtype=torch.float32, device=proposals_idx.device)
        _run("gcn_get_mask_iou", proposals_idx, nInstance, nProposal, _lib.ptr(proposals_idx),
             _lib.ptr(proposals_offset), _lib.ptr(instance_labels), _lib.ptr(instance_pointnum),
             _lib.ptr(mask_scores_sigmoid), _lib.ptr(iou))
        return iou

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None, None


get_mask_iou_on_pred = GetMaskIoUOnPred.apply


class GetMaskLabel(Function):
    @staticmethod
    def forward(ctx, proposals_idx, proposals_offset, instance_labels, instance_cls, instance_pointnum,
                proposals_iou, iou_thr):
        """functions.py:235-271."""
        nInstance = instance_pointnum.size(0)
        nProposal = proposals_offset.size(0) - 1
        for t in (proposals_iou, proposals_idx, proposals_offset, instance_labels, instance_cls):
            assert t.is_contiguous() and t.is_cuda
        mask_label = torch.full(proposals_idx.shape, -1.0, dtype=torch.float32, device=proposals_idx.device)
        _run("gcn_get_mask_label", proposals_idx, nInstance, nProposal, float(iou_thr), _lib.ptr(proposals_idx),
             _lib.ptr(proposals_offset), _lib.ptr(instance_labels), _lib.ptr(instance_cls),
             _lib.ptr(proposals_iou), _lib.ptr(mask_label))
        return mask_label

    @staticmethod
    def backward(ctx, a=None):
        return (None,) * 7


get_mask_label = GetMaskLabel.apply


# ------------------------------------------------------------------ voxelization
class Voxelization_Idx(Function):
    @staticmethod
    def forward(ctx, coords, batchsize, mode=4):
        """functions.py:281-307: coords CPU int64 (N,3|4) -> (output_coords i64 (M,.), input_map i32 (N),
        output_map i32 (M, maxActive+1)), all CPU."""
        assert coords.is_contiguous()
        assert coords.dtype == torch.int64
        N, ncol = coords.shape
        if coords.is_cuda:       # device path (extension: the reference only takes CPU tensors here); CUDA tensors out
            dev = coords.device
            input_map = torch.empty(N, dtype=torch.int32, device=dev)
            ws = torch.empty(max(_lib.lib().gcn_voxelize_idx_ws_bytes(N), 64), dtype=torch.uint8, device=dev)
            M, maxA = C.c_int(0), C.c_int(0)
            args = (_lib.ptr(coords), N, ncol, int(mode), _lib.ptr(input_map), C.addressof(M), C.addressof(maxA))
            _run("gcn_voxelize_idx", coords, *args, None, None, _lib.ptr(ws))
            output_coords = torch.empty(M.value, ncol, dtype=torch.int64, device=dev)
            output_map = torch.empty(M.value, maxA.value + 1, dtype=torch.int32, device=dev)
            if N > 0:
                _run("gcn_voxelize_idx", coords, *args, _lib.ptr(output_coords), _lib.ptr(output_map), _lib.ptr(ws))
            return output_coords, input_map, output_map
        input_map = torch.zeros(N, dtype=torch.int32)
        M, maxA = C.c_int(0), C.c_int(0)
        args = (_lib.ptr(coords), N, ncol, int(mode), _lib.ptr(input_map), C.addressof(M), C.addressof(maxA))
        _lib.call("gcn_voxelize_idx_host", *args, None, None)
        output_coords = torch.zeros(M.value, ncol, dtype=torch.int64)
        output_map = torch.zeros(M.value, maxA.value + 1, dtype=torch.int32)
        _lib.call("gcn_voxelize_idx_host", *args, _lib.ptr(output_coords), _lib.ptr(output_map))
        return output_coords, input_map, output_map

    @staticmethod
    def backward(ctx, a=None, b=None, c=None):
        return None


voxelization_idx = Voxelization_Idx.apply


class Voxelization(Function):
    @staticmethod
    def forward(ctx, feats, map_rule, mode=4):
        """functions.py:313-334: feats (N,C) cuda f32, map_rule (M,maxActive+1) cuda i32 -> (M,C)."""
        assert map_rule.is_contiguous()
        assert feats.is_contiguous()
        _lib.require_cuda(feats, map_rule)
        N, Cc = feats.size()
        M = map_rule.size(0)
        maxActive = map_rule.size(1) - 1
        out = torch.empty(M, Cc, dtype=torch.float32, device=feats.device)
        ctx.for_backwards = (map_rule, mode, maxActive, N)
        _run("gcn_voxelize_fp", feats, M, maxActive, Cc, _lib.ptr(feats), _lib.ptr(out), _lib.ptr(map_rule), int(mode == 4))
        return out

    @staticmethod
    def backward(ctx, d_output_feats):
        map_rule, mode, maxActive, N = ctx.for_backwards
        M, Cc = d_output_feats.size()
        d_output_feats = d_output_feats.contiguous()
        d_feats = torch.zeros(N, Cc, dtype=torch.float32, device=d_output_feats.device)
        _run("gcn_voxelize_bp", d_output_feats, M, maxActive, Cc, _lib.ptr(d_output_feats), _lib.ptr(d_feats),
             _lib.ptr(map_rule), int(mode == 4))
        return d_feats, None, None


voxelization = Voxelization.apply


# ------------------------------------------------------------------ segment reductions
class GlobalAvgPool(Function):
    @staticmethod
    def forward(ctx, feats, proposals_offset):
        """functions.py:632-652."""
        nProposal = proposals_offset.size(0) - 1
        sumNPoint, Cc = feats.size()
        assert feats.is_contiguous()
        assert proposals_offset.is_contiguous()
        _lib.require_cuda(feats, proposals_offset)
        out = torch.zeros(nProposal, Cc, dtype=torch.float32, device=feats.device)
        _run("gcn_global_avg_pool_fp", feats, nProposal, Cc, _lib.ptr(feats), _lib.ptr(proposals_offset), _lib.ptr(out))
        ctx.for_backwards = (proposals_offset, sumNPoint)
        return out

    @staticmethod
    def backward(ctx, d_output_feats):
        nProposal, Cc = d_output_feats.size()
        proposals_offset, sumNPoint = ctx.for_backwards
        d_output_feats = d_output_feats.contiguous()
        d_feats = torch.zeros(sumNPoint, Cc, dtype=torch.float32, device=d_output_feats.device)
        _run("gcn_global_avg_pool_bp", d_output_feats, nProposal, Cc, _lib.ptr(d_feats), _lib.ptr(proposals_offset),
             _lib.ptr(d_output_feats))
        return d_feats, None


global_avg_pool = GlobalAvgPool.apply


def _make_sec(op_id, doc):
    class _Sec(Function):
        @staticmethod
        def forward(ctx, inp, offsets):
            nProposal = offsets.size(0) - 1
            Cc = inp.size(1)
            assert inp.is_contiguous()
            assert offsets.is_contiguous()
            _lib.require_cuda(inp, offsets)
            out = torch.zeros(nProposal, Cc, dtype=torch.float32, device=inp.device)
            _run("gcn_sec_op", inp, op_id, nProposal, Cc, _lib.ptr(inp), _lib.ptr(offsets), _lib.ptr(out))
            return out

        @staticmethod
        def backward(ctx, a=None):
            return None, None

    _Sec.__doc__ = doc
    return _Sec


SecMean = _make_sec(0, "functions.py:670-694")
SecMin = _make_sec(1, "functions.py:700-724")
SecMax = _make_sec(2, "functions.py:730-754")
sec_mean, sec_min, sec_max = SecMean.apply, SecMin.apply, SecMax.apply
