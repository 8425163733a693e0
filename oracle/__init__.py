"""CPU ORACLE -- test infrastructure, not product code.

numpy front-end to ``oracle/_build/libgcanet_oracle.so`` (plain-C restatement of the
reference's native ops, see ``gcanet_oracle.c``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; nothing under ``gcanet_amd/`` does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgcanet_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (idempotent)."""
    src = os.path.join(_HERE, "gcanet_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_ballquery_batch_p.restype = C.c_int
        _lib.orc_opt_n_threads.restype = C.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# ---------------------------------------------------------------- KNN_CUDA
def knn_cuda(ref, query, k):
    """ref (dim,nr), query (dim,nq) -> D (k,nq) f32 L2, I (k,nq) int64 0-based
    (KNN/__init__.py:41-44 ``knn``)."""
    ref, query = _f32(ref), _f32(query)
    dim, nr = ref.shape
    nq = query.shape[1]
    assert query.shape[0] == dim and 1 <= k <= nr
    d = np.empty((k, nq), np.float32)
    i = np.empty((k, nq), np.int64)
    lib().orc_knn_cuda(_p(ref), nr, _p(query), nq, dim, k, _p(d), _p(i))
    return d, i


def KNN_forward(ref, query, k, transpose_mode=False):
    """Batched ``KNN(k, transpose_mode).forward`` (KNN/__init__.py:61-74)."""
    D, I = [], []
    for b in range(ref.shape[0]):
        r, q = ref[b], query[b]
        if transpose_mode:
            r, q = r.T, q.T
        d, i = knn_cuda(r, q, k)
        if transpose_mode:
            d, i = d.T, i.T
        D.append(np.ascontiguousarray(d))
        I.append(np.ascontiguousarray(i))
    return np.stack(D), np.stack(I)


# ---------------------------------------------------------------- in-model kNN
def knn_model(x, k1, k2, metric=0, return_values=False):
    """x (B,C,N) -> idx (B,N,len(range(0,k2,k2//k1))) int64 (M4:30-90).
    metric 0 = ``knn`` (expanded form), 1 = ``knn_points_normals``."""
    x = _f32(x)
    B, Cc, N = x.shape
    idx = np.empty((B, N, k2), np.int64)
    val = np.empty((B, N, k2), np.float32)
    for b in range(B):
        xb = np.ascontiguousarray(x[b])
        lib().orc_knn_model(_p(xb), Cc, N, k2, metric, _p(idx[b]), _p(val[b]))
    pick = np.arange(0, k2, k2 // k1)
    if return_values:
        return idx[:, :, pick], val[:, :, pick]
    return idx[:, :, pick]


# ---------------------------------------------------------------- pointnet2_ops
def ball_query(radius, nsample, xyz, new_xyz):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(b, n, m, C.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
    return idx


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.empty((b, c, npoints, nsample), np.float32)
    lib().orc_group_points(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, npoints, nsample = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad(b, c, n, npoints, nsample, _p(grad_out), _p(idx), _p(g))
    return g


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.empty((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad(b, c, n, m, _p(grad_out), _p(idx), _p(g))
    return g


def furthest_point_sampling(xyz, npoint):
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)
    idx = np.zeros((b, npoint), np.int32)
    lib().orc_furthest_point_sampling(b, n, npoint, _p(xyz), _p(temp), _p(idx))
    return idx


def three_nn(unknown, known):
    """Returns (dist2, idx) -- the Python wrapper takes sqrt (pointnet2_utils.py:124-125)."""
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    with np.errstate(over="ignore"):
        lib().orc_three_nn(b, n, m, _p(unknown), _p(known), _p(d2), _p(idx))
    return d2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.empty((b, c, n), np.float32)
    lib().orc_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), np.float32)
    lib().orc_three_interpolate_grad(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


# ---------------------------------------------------------------- softgroup.ops
def voxelization_idx(coords, batchsize, mode=4):
    """coords (N,3|4) int64 -> (output_coords int64 (M,ncol), input_map i32 (N),
    output_map i32 (M,maxActive+1))  (SG/functions.py:281-310)."""
    coords = _i64(coords)
    N, ncol = coords.shape
    input_map = np.zeros(N, np.int32)
    M, maxA = C.c_int(0), C.c_int(0)
    lib().orc_voxelize_idx(_p(coords), N, ncol, mode, _p(input_map), C.byref(M), C.byref(maxA), None, None)
    oc = np.zeros((M.value, ncol), np.int64)
    om = np.zeros((M.value, maxA.value + 1), np.int32)
    lib().orc_voxelize_idx(_p(coords), N, ncol, mode, _p(input_map), C.byref(M), C.byref(maxA), _p(oc), _p(om))
    return oc, input_map, om


def voxelization(feats, map_rule, mode=4):
    feats, map_rule = _f32(feats), _i32(map_rule)
    M, W = map_rule.shape
    Cc = feats.shape[1]
    out = np.zeros((M, Cc), np.float32)
    lib().orc_voxelize_fp(M, W - 1, Cc, _p(feats), _p(out), _p(map_rule), int(mode == 4))
    return out


def voxelization_bp(d_out, map_rule, N, mode=4):
    d_out, map_rule = _f32(d_out), _i32(map_rule)
    M, W = map_rule.shape
    Cc = d_out.shape[1]
    d_feats = np.zeros((N, Cc), np.float32)
    lib().orc_voxelize_bp(M, W - 1, Cc, _p(d_out), _p(d_feats), _p(map_rule), int(mode == 4))
    return d_feats


def ballquery_batch_p(coords, batch_idxs, batch_offsets, radius, mean_active,
                      adj_inst=None, thr_inst=0.0, adj_para=None, thr_para=0.0):
    """BallQueryBatchP / _Easy incl. the meanActive retry loop (SG/functions.py:434-541)."""
    coords, batch_idxs, batch_offsets = _f32(coords), _i32(batch_idxs), _i32(batch_offsets)
    n = coords.shape[0]
    if adj_inst is not None:
        adj_inst, adj_para = _f32(adj_inst), _f32(adj_para)
    while True:
        idx = np.zeros(n * mean_active, np.int32)
        start_len = np.zeros((n, 2), np.int32)
        nActive = lib().orc_ballquery_batch_p(n, mean_active, C.c_float(radius), _p(coords), _p(batch_idxs),
                                              _p(batch_offsets), _p(adj_inst), C.c_float(thr_inst),
                                              _p(adj_para), C.c_float(thr_para), _p(idx), _p(start_len))
        if nActive <= n * mean_active:
            break
        mean_active = int(nActive // n + 1)
    return idx[:nActive], start_len


def octree_ball_query(coords, mean_active, radius):
    """softgroup/ops/functions.py:127-157 + octree_ball_query.cpp/.cu: (idx (nActive,) i32, start_len (n,2) i32) with the
    reference's per-point list order (active leaves in breadth-first order, ascending index inside a leaf); CSR
    segments in point order (the reference's segment order is an atomicAdd race)."""
    coords = _f32(coords)
    n = coords.shape[0]
    mx, mn = coords.max(0), coords.min(0)
    xyzwhl = np.concatenate([(mx + mn) / 2, mx - mn]).astype(np.float32)
    boxes = np.zeros((585, 6), np.float32)
    leaf = np.zeros(n, np.int32)
    lib().orc_octree_build(_p(coords), n, _p(xyzwhl), _p(boxes), _p(leaf))
    order = np.lexsort((np.arange(n), leaf)).astype(np.int32)                 # by (leaf, index): export_data's pt_inds
    counts = np.bincount(leaf, minlength=512).astype(np.int32)
    psl = np.stack([np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32), counts], 1).copy()
    sl = np.zeros((n, 2), np.int32)
    total = lib().orc_octree_ball_query(_p(coords), n, _p(boxes), _p(order), _p(psl), int(mean_active), C.c_float(radius),
                                        None, _p(sl))
    while total > n * mean_active:                                            # functions.py:147-154
        mean_active = int(total // n + 1)
        total = n * mean_active
    idx = np.zeros(max(total, 1), np.int32)
    total = lib().orc_octree_ball_query(_p(coords), n, _p(boxes), _p(order), _p(psl), int(mean_active), C.c_float(radius),
                                        _p(idx), _p(sl))
    return idx[:total], sl, leaf


def bfs_cluster(class_numpoint_mean, ball_query_idxs, start_len, threshold, class_id):
    cm, bq, sl = _f32(class_numpoint_mean), _i32(ball_query_idxs), _i32(start_len)
    N = sl.shape[0]
    s, c = C.c_int(0), C.c_int(0)
    lib().orc_bfs_cluster(_p(cm), _p(bq), _p(sl), N, C.c_float(threshold), class_id, C.byref(s), C.byref(c), None, None)
    ci = np.zeros((s.value, 2), np.int32)
    co = np.zeros(c.value + 1, np.int32)
    lib().orc_bfs_cluster(_p(cm), _p(bq), _p(sl), N, C.c_float(threshold), class_id, C.byref(s), C.byref(c), _p(ci), _p(co))
    return ci, co


def hierarchical_aggregation(semantic_label, coord_shift, ball_query_idxs, start_len, batch_idxs,
                             training_mode="train", using_set_aggr=False):
    """C split + the Python merge of HierarchicalAggregation.forward (SG/functions.py:7-72)."""
    sem, cs, bq = _i32(semantic_label), _f32(coord_shift), _i32(ball_query_idxs)
    sl, bi = _i32(start_len), _i32(batch_idxs)
    N = sl.shape[0]
    n1 = max(N, 1)
    idxs = [np.zeros((n1, 2), np.int32) for _ in range(3)]
    offs = [np.zeros(n1 + 1, np.int32) for _ in range(3)]
    cents = [np.zeros((n1, 5), np.float32) for _ in range(3)]
    counts = np.zeros(6, np.int32)
    lib().orc_hier_split(_p(sem), _p(cs), _p(bi), _p(bq), _p(sl), N,
                         _p(idxs[0]), _p(offs[0]), _p(cents[0]),
                         _p(idxs[1]), _p(offs[1]), _p(cents[1]),
                         _p(idxs[2]), _p(offs[2]), _p(cents[2]), _p(counts))
    cut = lambda t: (idxs[t][:counts[2 * t]].copy(), offs[t][:counts[2 * t + 1] + 1].copy(),
                     cents[t][:counts[2 * t + 1]].copy())
    frag, kept, prim = cut(0), cut(1), cut(2)
    primary_idxs, primary_offsets = prim[0], prim[1]
    if using_set_aggr:
        post = np.zeros((counts[0] + counts[4], 2), np.int32)
        post_off = np.zeros(counts[5] + 1, np.int32)
        lib().orc_hier_set_aggr(int(counts[1]), _p(frag[0]), _p(frag[1]), _p(frag[2]),
                                int(counts[5]), _p(prim[0]), _p(prim[1]), _p(prim[2]), _p(post), _p(post_off))
        primary_idxs = post[:post_off[-1]]
        primary_offsets = post_off
    cluster_idxs, cluster_offsets = kept[0], kept[1]
    if primary_idxs.shape[0] != 0:
        primary_idxs = primary_idxs.copy()
        primary_idxs[:, 0] += cluster_offsets.shape[0] - 1
        primary_offsets = primary_offsets + cluster_offsets[-1]
        cluster_idxs = np.concatenate([cluster_idxs, primary_idxs], 0)
        cluster_offsets = np.concatenate([cluster_offsets, primary_offsets[1:]])
    return cluster_idxs, cluster_offsets


def sec_op(op, inp, offsets):
    inp, offsets = _f32(inp), _i32(offsets)
    P, Cc = offsets.shape[0] - 1, inp.shape[1]
    out = np.zeros((P, Cc), np.float32)
    with np.errstate(all="ignore"):
        lib().orc_sec_op({"mean": 0, "min": 1, "max": 2}[op], P, Cc, _p(inp), _p(offsets), _p(out))
    return out


def global_avg_pool(feats, offsets):
    feats, offsets = _f32(feats), _i32(offsets)
    P, Cc = offsets.shape[0] - 1, feats.shape[1]
    out = np.zeros((P, Cc), np.float32)
    lib().orc_global_avg_pool_fp(P, Cc, _p(feats), _p(offsets), _p(out))
    return out


def global_avg_pool_bp(d_out, offsets, S):
    d_out, offsets = _f32(d_out), _i32(offsets)
    P, Cc = d_out.shape
    d_feats = np.zeros((S, Cc), np.float32)
    lib().orc_global_avg_pool_bp(P, Cc, _p(d_feats), _p(offsets), _p(d_out))
    return d_feats


def get_mask_iou(proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid=None):
    pi, po = _i32(proposals_idx), _i32(proposals_offset)
    il, ip = _i64(instance_labels), _i32(instance_pointnum)
    ms = _f32(mask_scores_sigmoid) if mask_scores_sigmoid is not None else None
    nI, nP = ip.shape[0], po.shape[0] - 1
    iou = np.zeros((nP, nI), np.float32)
    lib().orc_get_mask_iou(nI, nP, _p(pi), _p(po), _p(il), _p(ip), _p(ms), _p(iou))
    return iou


def get_mask_label(proposals_idx, proposals_offset, instance_labels, instance_cls, instance_pointnum,
                   proposals_iou, iou_thr):
    pi, po = _i32(proposals_idx), _i32(proposals_offset)
    il, ic, iou = _i64(instance_labels), _i64(instance_cls), _f32(proposals_iou)
    nI, nP = np.asarray(instance_pointnum).shape[0], po.shape[0] - 1
    ml = np.full(pi.shape, -1.0, np.float32)
    lib().orc_get_mask_label(nI, nP, C.c_float(iou_thr), _p(pi), _p(po), _p(il), _p(ic), _p(iou), _p(ml))
    return ml
