/*
 * gcanet_hip.h -- C ABI of libgcanet_hip.so: hand-written HIP (gfx950) kernels for
 * GCANet's per-point feature-aggregation hot path.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host.
 *   - the caller owns all buffers (the reference's callee-allocates convention,
 *     e.g. group_points.cpp:22-24 / knn.cpp:36-37, is reproduced by the Python shims).
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all
 *     work is enqueued on it and nothing synchronises unless stated.
 *   - return value: GCN_OK (0) or an error code; gcn_last_error() gives a
 *     thread-local message.  Never exit()s (the reference does: cuda_utils.h:30-39).
 *   - re-entrant, no global mutable state.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the
 * reference repo root; P2 = models/Pointnet2_PyTorch-master/pointnet2_ops_lib/pointnet2_ops,
 * KNN = models/KNN_CUDA/knn_cuda, SG = softgroup/ops, M4 = models/dgcnn-hais-concat-direct-4.py).
 */
#ifndef GCANET_HIP_H
#define GCANET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCN_OK 0
#define GCN_EINVAL 1   /* bad argument (shape / null pointer / unsupported size) */
#define GCN_ELAUNCH 2  /* HIP launch or runtime error */

const char *gcn_last_error(void);
int gcn_version(void);

/* ------------------------------------------------------------------ kNN ------- */

/* Replaces KNN/csrc/cuda/knn.cpp:23-56 `knn(ref, query, k)` + kernels knn.cu:29-183
 * (distance matrix, insertion sort, sqrt) + the `i -= 1` of KNN/__init__.py:41-44,
 * batched (KNN.forward loops over the batch in Python, KNN/__init__.py:61-74).
 * One fused kernel; the (nr,nq) distance matrix is never materialised.
 *   ref   (B, dim, nr) if point_major == 0 else (B, nr, dim)   f32
 *   query (B, dim, nq) if point_major == 0 else (B, nq, dim)   f32
 *   dist  (B, k, nq) if point_major == 0 else (B, nq, k)       f32, L2 distance (sqrt applied)
 *   ind   same shape as dist, int64, 0-based; ties -> lowest ref index (knn.cu:125-131)
 * 1 <= k <= min(nr, 512). */
int gcn_knn_cuda(const float *ref, const float *query, int B, int dim, int nr, int nq, int k,
                 int point_major, float *dist, int64_t *ind, void *tile_ws, void *stream);

/* Replaces the pure-torch `knn(x,k1,k2)` (M4:30-47, metric 0) and
 * `knn_points_normals(x,k1,k2)` (M4:50-90, metric 1): per-cloud N x N negative squared
 * distance + topk(k2) + dilated pick `[:, :, arange(0,k2,k2//k1)]`.
 *   x    (B, C, N) f32 channel-major (metric 1 needs C >= 6: xyz + normal)
 *   idx  (B, N, kout) int64, kout = ceil(k2 / (k2/k1)); order: value desc, index asc
 *   val  optional (B, N, kout) f32: the selected pairwise_distance values (may be NULL)
 *   xx_ws workspace (B, N) f32 (squared norms; written by the call)
 * 1 <= k1 <= k2 <= min(N, 512). */
int gcn_knn_model(const float *x, int B, int C, int N, int k1, int k2, int metric,
                  int64_t *idx, float *val, float *xx_ws, void *tile_ws, void *stream);

/* Scratch for the spatially pruned path of gcn_knn_model (3-D clouds: C == 3, or C == 6 with metric 1; k2 <= 64,
 * N >= 512): Morton sort + 64-point tiles with bounding boxes, tiles skipped by a conservative lower bound of the
 * metric -- identical indices and values to the brute-force kernel.  tile_ws == NULL keeps the brute-force scan.
 * gcn_knn_cuda takes the same workspace (gcn_knn_tiles_ws_bytes(B, 3, nr)) and uses it when a 3-D cloud is searched
 * against itself (ref == query, k <= 64, nr >= 512). */
long gcn_knn_tiles_ws_bytes(int B, int C, int N);

/* knn_points_normals (metric 1, C == 6) is served by threshold + filter + re-rank in the reference's exact arithmetic
 * (csrc/knn_normal.hip: sampled order statistic -> one bit per pair -> exact ranking of ~3k survivors; identical
 * indices/values) when this returns 1 (1024 <= N <= 16384 -- any N: the candidate rows are padded to a multiple of 1024
 * with far-away points --, k2 <= 128) AND tile_ws is given: gcn_knn_tiles_ws_bytes(B, 6, N) then includes the
 * B*N*Np/8-byte bitmap.  The same scheme serves gcn_knn_cuda for a
 * 3-D cloud searched against itself (KNN(k)(x, x), squared distances by differences) under the same size conditions,
 * with gcn_knn_tiles_ws_bytes(B, 3, nr) bytes of workspace. */
int gcn_knn_normal_supported(int B, int N, int k2);

/* Feature-space kNN of the in-model `knn` (models/dgcnn-hais-concat-direct-4.py:30-47) for C in {32,64,128} as a
 * bf16 matrix-core PREFILTER + exact f32 re-rank (csrc/knn_filter.hip): same indices as gcn_knn_model(metric 0), bit
 * for bit (ties -> lowest index), several times faster at N >= 1024.
 *   x_pm (B,N,C) f32 POINT-major rows (the layout the fused EdgeConv already keeps), idx (B,N,kout) int64 as
 *   gcn_knn_model; ws: gcn_knn_feature_ws_bytes(B,N,C) bytes of device scratch, 256-B aligned (B*N*N/8 bytes of
 *   candidate bits dominate: 67 MB at B=8, N=8192).
 * gcn_knn_feature_supported: 1 when the shape is served (C in {32,64,128}, 1024 <= N <= 16384 -- any N, e.g. the
 * reference's default 7000: candidate rows are padded to a multiple of 128 --, k2 <= 128, e.g. its default 80); other
 * shapes use gcn_knn_model.
 * gcn_knn_feature_stats (diagnostics, synchronises): queries of the last call that fell back to the exact brute-force
 * search, and the total number of prefilter candidates. */
int gcn_knn_feature_supported(int B, int N, int C, int k2);
long gcn_knn_feature_ws_bytes(int B, int N, int C);
int gcn_knn_feature(const float *x_pm, int B, int N, int C, int k1, int k2, int64_t *idx, void *ws,
                    void *stream);
int gcn_knn_feature_stats(const void *ws, int B, int N, int C, long *flagged, long *candidates, void *stream);

/* ------------------------------------------------------------ pointnet2_ops ---- */

/* P2/_ext-src/src/ball_query.cpp:10-33 `ball_query(new_xyz, xyz, radius, nsample)`.
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample) int32 (fully written: rows with no
 * hit are zero, ball_query.cpp:19-21). */
int gcn_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                   const float *xyz, int32_t *idx, void *stream);

/* P2/_ext-src/src/group_points.cpp:14-37 `group_points(points, idx)`:
 * points (b,c,n), idx (b,npoints,nsample) i32 -> out (b,c,npoints,nsample). */
int gcn_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                     const int32_t *idx, float *out, void *stream);

/* P2/_ext-src/src/group_points.cpp:39-63 `group_points_grad(grad_out, idx, n)`:
 * grad_points (b,c,n) is fully written by the call (zero-init included). */
int gcn_group_points_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                          const int32_t *idx, float *grad_points, void *stream);

/* P2/_ext-src/src/sampling.cpp:18-40 `gather_points(points, idx)`: (b,c,n),(b,m) -> (b,c,m). */
int gcn_gather_points(int b, int c, int n, int m, const float *points, const int32_t *idx,
                      float *out, void *stream);

/* P2/_ext-src/src/sampling.cpp:42-66 `gather_points_grad(grad_out, idx, n)`. */
int gcn_gather_points_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                           float *grad_points, void *stream);

/* P2/_ext-src/src/sampling.cpp:68-88 `furthest_point_sampling(points, nsamples)`:
 * dataset (b,n,3) -> idxs (b,m) i32; temp (b,n) f32 workspace (initialised by the call). */
int gcn_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                int32_t *idxs, void *stream);

/* P2/_ext-src/src/interpolate.cpp:22-44 `three_nn(unknowns, knows)`:
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) f32 (SQUARED; the Python wrapper takes
 * sqrt, pointnet2_utils.py:124-125), idx (b,n,3) i32. */
int gcn_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                 int32_t *idx, void *stream);

/* P2/_ext-src/src/interpolate.cpp:46-72 `three_interpolate(points, idx, weight)`:
 * points (b,c,m), idx/weight (b,n,3) -> out (b,c,n). */
int gcn_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                          const float *weight, float *out, void *stream);

/* P2/_ext-src/src/interpolate.cpp:74-101 `three_interpolate_grad(grad_out, idx, weight, m)`:
 * grad_points (b,c,m) fully written. */
int gcn_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                               const int32_t *idx, const float *weight, float *grad_points,
                               void *stream);

/* ------------------------------------------------------------- softgroup.ops --- */

/* SG/src/voxelize/voxelize.cpp:168-183 `voxelize_fp` (kernel voxelize.cu:9-25):
 * feats (N,C), rules (M, maxActive+1) i32 [count, idx...] -> output_feats (M,C) fully written. */
int gcn_voxelize_fp(int M, int maxActive, int C, const float *feats, float *output_feats,
                    const int32_t *rules, int average, void *stream);

/* SG/src/voxelize/voxelize.cpp:185-198 `voxelize_bp` (kernel voxelize.cu:38-54):
 * d_feats (N,C) must be zero-initialised by the caller (functions.py:340). */
int gcn_voxelize_bp(int M, int maxActive, int C, const float *d_output_feats, float *d_feats,
                    const int32_t *rules, int average, void *stream);

/* SG/src/bfs_cluster/bfs_cluster.cpp:20-46 `ballquery_batch_p` (kernel bfs_cluster.cu:18-77)
 * and SG/src/bfs_cluster_easy (adj_* == NULL; per-point cap 1000 instead of 3000).
 * Two-pass count -> scan -> fill: CSR segments are in POINT ORDER (the reference's
 * atomicAdd slot allocation makes its segment order non-deterministic).
 *   xyz (n,3), batch_idxs (n) i32, batch_offsets (B+1) i32, adj_* (n,n) f32 or NULL
 *   idx (n*meanActive) i32 (entries past the truncation point untouched),
 *   start_len (n,2) i32, count_ws (n+1) i32 workspace, total_host: pinned or pageable
 *   HOST int receiving the untruncated total (the call synchronises the stream for it,
 *   as the reference's blocking cudaMemcpy does, bfs_cluster.cu:118).
 * With grid_ws (easy form only) candidates come from a uniform grid of cells >= radius built on the device
 * (counting sort by cell, 27-cell neighbourhood): same lists, ~75 distance tests per point instead of n. */
int gcn_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz,
                          const int32_t *batch_idxs, const int32_t *batch_offsets,
                          const float *adj_inst, float thr_inst, const float *adj_para,
                          float thr_para, int32_t *idx, int32_t *start_len, int32_t *count_ws,
                          int nbatch, void *grid_ws, int *total_host, void *stream);

/* ---- forward_grouping on the device (models/dgcnn-hais-concat-direct-4.py:1123-1295) ----
 * The reference loops over (cloud, class) subsets; per subset it builds two dense (n,n) similarity matrices
 * (compute_batch_adjacency_matrix, :210-233), calls ballquery_batch_p (bfs_cluster.cu:18-77) and then the HOST BFS of
 * hierarchical_aggregation.cpp:20-131.  These entry points do all subsets ("segments": points sorted by
 * cloud*P + class) at once and never form an (n,n) tensor.
 *
 * gcn_segment_diameter2: dmax2[s] = max_{i != j in segment s} ||f_i - f_j||^2 (the `max` of :223-225 squared; its `min`
 * is 0, the zeroed diagonal).  feats (n,C) rows in segment order, C a multiple of 16 (zero-pad), seg_offsets (S+1),
 * seg_cls (S) (< 0: segment skipped, dmax2 = 0), xx_ws (n) floats, tile_ws (S+1) ints. */
int gcn_segment_diameter2(int n, int C, const float *feats, const int32_t *seg_offsets, const int32_t *seg_cls, int S,
                          float *xx_ws, int32_t *tile_ws, float *dmax2, void *stream);
/* The same result, bit for bit, from a bf16 tile-maximum prefilter on the matrix cores plus the exhaustive kernel's f32
 * arithmetic on the few 64x64 tiles that can hold the maximum (csrc/segdiam.hip); 16 <= C <= 256.  ws: at least
 * gcn_segment_diameter2_ws_bytes(n, C, S) bytes, 256-B aligned (-1: shape not supported). */
long gcn_segment_diameter2_ws_bytes(int n, int C, int S);
int gcn_segment_diameter2_filtered(int n, int C, const float *feats, const int32_t *seg_offsets, const int32_t *seg_cls,
                                   int S, void *ws, float *dmax2, void *stream);
long gcn_ballquery_sim_ws_bytes(int n);
/* ballquery_batch_p with `adj_inst[p][k] > thr_inst && adj_para[p][k] > thr_para` evaluated from the feature rows:
 * adj = exp(-(||f_p - f_k|| / dmax)^2 / 2), 0 for p == k, NaN (never accepted) when dmax == 0.  One asynchronous pass:
 * every point reserves its list in idx (capacity ints, cut into 64 regions with a counter each) with one atomic, so the
 * lists lie in completion order and
 * start_len (n,2) = (start, count capped at 3000 as bfs_cluster.cu:54) addresses them; each list is ascending, as the
 * reference's scan order.  Feature rows zero-padded to a multiple of 16 columns.  status (4 ints, device):
 * [0] a capacity that holds every list, [1] != 0 if a list hit the cap, [2] != 0 if idx was too small (lists missing:
 * call again with capacity = status[0]; the reference's n*meanActive retry, functions.py:460-474).  No host sync. */
int gcn_ballquery_sim(int n, float radius, const float *xyz, const int32_t *seg_of, const int32_t *seg_offsets,
                      const int32_t *seg_cls, int S, const float *feat_inst, int Ci, const float *dmax2_inst,
                      float thr_inst, const float *feat_para, int Cp, const float *dmax2_para, float thr_para,
                      int32_t *idx, int capacity, int32_t *start_len, int32_t *status, void *grid_ws, void *stream);
/* hierarchical_aggregation (hierarchical_aggregation.cpp:20-131 + the kept/primary merge of functions.py:52-72,
 * using_set_aggr = False) for all segments on the device: components of the (symmetric) neighbour lists, members in the
 * reference's BFS dequeue order, segment by segment, kept fragments before primaries.  cluster_idxs (n,2) gets
 * (cluster id, point_index[member]) rows, cluster_offsets (n+1); counts (2 ints, device) = (rows, clusters) that are
 * valid.  No host synchronisation.  seg_cls (S): semantic class 0..9 of the segment or < 0 to skip it.  n < 2^20.
 * size_threshold = -2: as -1 but the fragments below 0.05*mean are listed as well (input of gcn_set_aggregation).
 * size_threshold < 0: the kept/primary rule above; >= 0: `bfs_cluster` (bfs_cluster.cpp:48-143) -- every component of at
 * least size_threshold points, in discovery order (one segment: S = 1, seg_cls[0] = 0). */
long gcn_cluster_components_ws_bytes(int n);
int gcn_cluster_components(int n, const int32_t *nbr, const int32_t *start_len, const int32_t *seg_of,
                           const int32_t *seg_offsets, const int32_t *seg_cls, int S, const int32_t *point_index,
                           float size_threshold, void *ws, int32_t *cluster_idxs, int32_t *cluster_offsets,
                           int32_t *counts, void *stream);
/* The same result for a caller that does not need the neighbour lists afterwards: `nbr` is used as scratch (each node's
 * discovered neighbours are compacted to the front of its list), which saves one of the three list scans per BFS level. */
int gcn_cluster_components_clobber(int n, int32_t *nbr, const int32_t *start_len, const int32_t *seg_of,
                                   const int32_t *seg_offsets, const int32_t *seg_cls, int S, const int32_t *point_index,
                                   float size_threshold, void *ws, int32_t *cluster_idxs, int32_t *cluster_offsets,
                                   int32_t *counts, void *stream);
/* Set aggregation of hierarchical_aggregation (hierarchical_aggregation.cu:22-196, using_set_aggr = True: evaluation) on
 * the device.  Input: gcn_cluster_components(size_threshold = -2) -- ALL components per segment (fragments in discovery
 * order, then primaries; pass point_index = 0..n-1 there so that rows carry sorted positions), counts_in (2) device.
 * Every fragment joins the nearest primary of its segment if the centroids are closer than 0.01*sqrt(|primary|) (at
 * most 1000 fragments / 3000 points per primary, fragment index order -- the oracle's order; the reference's comes from
 * atomicAdd).  Output as gcn_cluster_components: kept fragments then primaries (with their absorbed points) per
 * segment, out_counts (2) device = (rows, clusters).  xyz (n,3) = the shifted coordinates in sorted order. */
long gcn_set_aggregation_ws_bytes(int n, int S);
int gcn_set_aggregation(int n, int S, const int32_t *counts_in, const int32_t *rows, const int32_t *offs,
                        const int32_t *seg_of, const int32_t *seg_cls, const float *xyz, const int32_t *point_index,
                        void *ws, int32_t *out_idxs, int32_t *out_offs, int32_t *out_counts, void *stream);

/* ---- sparse 3-D convolutions of the instance "tiny U-Net" (softgroup/model/blocks.py:44-143; M4:611-616,1379-1392;
 * the reference calls the un-vendored third-party spconv package: SubMConv3d / SparseConv3d(k=2,s=2) /
 * SparseInverseConv3d).  Sparse tensor = features (M,C) f32 + coords (M,4) int32 [sample,x,y,z] with 0 <= x,y,z < D,
 * the on-wire format of clusters_voxelization (M4:1300-1355).  SURVEY.md section 8f rank 3. ---- */
long gcn_sparse_grid_bytes(int batch, int D);                 /* dense voxel-index grid, int32 [batch][D][D][D] */
/* nbr (M,27): index of the active voxel at offset (dx,dy,dz) in {-1,0,1}^3, column k = (dx+1)*9 + (dy+1)*3 + (dz+1);
 * -1 where there is none.  grid: gcn_sparse_grid_bytes scratch (left filled: grid[sample][x][y][z] = voxel or -1). */
int gcn_sparse_subm_rules(int M, const int32_t *coords, int batch, int D, int32_t *grid, int32_t *nbr, void *stream);
/* stride-2 kernel-2 downsampling: coarse voxels = occupied 2x2x2 cells, numbered in (sample,x,y,z) order.  coords2 (<=M,4),
 * child (<=M,8): fine voxel of coarse voxel o at corner k = (x&1)*4 + (y&1)*2 + (z&1) or -1; parent (M,8): row i holds its
 * coarse voxel in column k(i), -1 elsewhere (the rule table of the inverse convolution); *m2_dev (device) = number of
 * coarse voxels.  ws: gcn_sparse_coarse_ws_bytes. */
long gcn_sparse_coarse_ws_bytes(int batch, int D);
int gcn_sparse_coarse_rules(int M, const int32_t *coords, int batch, int D, void *ws, int32_t *coords2, int32_t *child,
                            int32_t *parent, int32_t *m2_dev, void *stream);
/* out (Mout,Cout) = sum_k in[rule[o,k], :] . W[k]  (rule < 0 contributes nothing); rule (Mout,K); W (K,Cin,Cout).
 * k_reversed pairs rule column K-1-k with W[k] (input gradient of a submanifold convolution, with the per-offset
 * transposed weight).  w_transposed must be 0 (reserved).  Cin, Cout multiples of 64.  f32 on v_mfma_f32_16x16x4_f32. */
int gcn_sparse_gather_gemm(int Mout, int K, int Cin, int Cout, const float *in, const int32_t *rule, const float *W,
                           int w_transposed, int k_reversed, float *out, float *ws, void *stream);
/* Small launches split the offsets over several workgroup groups and add the partial sums afterwards (fixed order):
 * floats of scratch `ws` the call above needs for this shape (0: none, ws may be NULL). */
long gcn_sparse_gather_gemm_ws_floats(int Mout, int K, int Cout);
/* dW (K,Cin,Cout) = sum_o in[rule[o,k], :]^T (x) dout[o, :]  (zeroed here first).  ruleT (K,Mout): the rule table
 * transposed, so that one offset's column is contiguous. */
int gcn_sparse_wgrad(int Mout, int K, int Cin, int Cout, const float *in, const int32_t *ruleT, const float *dout, float *dW,
                     void *stream);
/* BatchNorm1d(+ReLU) in training mode over the M rows of a sparse tensor's features (blocks.py norm_fn, eps 1e-4,
 * momentum 0.1, every instance followed by nn.ReLU).  y, mean_rstd (C,2) out; running_mean/var (C) updated as
 * nn.BatchNorm1d does (unbiased variance) or both NULL.  sums_ws / acc_ws: 2C doubles.  C/4 must divide 256. */
int gcn_bn_relu_fwd(int M, int C, const float *x, const float *gamma, const float *beta, float eps, int relu, float momentum,
                    float *y, float *mean_rstd, float *running_mean, float *running_var, double *sums_ws, void *stream);
int gcn_bn_relu_bwd(int M, int C, const float *dy, const float *x, const float *gamma, const float *beta,
                    const float *mean_rstd, int relu, float *dx, float *dgamma, float *dbeta, double *acc_ws, void *stream);
/* Device scratch for the uniform-grid path of gcn_ballquery_batch_p (easy form, n >= 2048): pass it as grid_ws
 * (NULL selects the brute-force scan).  nbatch = number of batch segments (len(batch_offsets) - 1). */
long gcn_ballquery_grid_ws_bytes(int n);

/* SG/src/sec_mean/sec_mean.cpp `sec_mean/sec_min/sec_max` (kernels sec_mean.cu:13-85).
 * op 0 mean, 1 min, 2 max.  inp (N,C), offsets (P+1) i32 -> out (P,C). */
int gcn_sec_op(int op, int P, int C, const float *inp, const int32_t *offsets, float *out,
               void *stream);

/* SG/src/roipool/roipool.cpp `global_avg_pool_fp` (roipool.cu:12-32). */
int gcn_global_avg_pool_fp(int P, int C, const float *feats, const int32_t *offsets, float *out,
                           void *stream);

/* SG/src/roipool/roipool.cpp `global_avg_pool_bp` (roipool.cu:46-60); d_feats (S,C) rows
 * inside [offsets[0], offsets[P]) are fully written (no zero-init needed there). */
int gcn_global_avg_pool_bp(int P, int C, float *d_feats, const int32_t *offsets,
                           const float *d_out, void *stream);

/* SG/src/cal_iou_and_masklabel `get_mask_iou_on_cluster` (mask == NULL, .cu:9-34) and
 * `get_mask_iou_on_pred` (.cu:36-68).  proposals_iou (P, I) f32. */
int gcn_get_mask_iou(int nInstance, int nProposal, const int32_t *proposals_idx,
                     const int32_t *proposals_offset, const int64_t *instance_labels,
                     const int32_t *instance_pointnum, const float *mask_scores_sigmoid,
                     float *proposals_iou, void *stream);

/* SG/src/cal_iou_and_masklabel `get_mask_label` (.cu:70-104); mask_label (S) pre-filled
 * with -1 by the caller (functions.py:252). */
int gcn_get_mask_label(int nInstance, int nProposal, float iou_thr, const int32_t *proposals_idx,
                       const int32_t *proposals_offset, const int64_t *instance_labels,
                       const int64_t *instance_cls, const float *proposals_iou, float *mask_label,
                       void *stream);

/* `voxelize_idx` (SG/src/voxelize/voxelize.cpp:11-165, a host hash table in the reference) on DEVICE buffers:
 * sort-based, same numbering (voxels by first appearance, rule rows in input order), every coordinate must lie
 * in [0, 65535].  Two-call protocol with the same ws (gcn_voxelize_idx_ws_bytes(N) bytes): call 1 with
 * output_coords == NULL fills input_map (N) and returns M / maxActive through host ints (synchronises the
 * stream); call 2 fills output_coords (M,ncol) i64 and output_map (M,maxActive+1) i32.  mode as the reference
 * (0 unique, 1 front, 2 back, 3/4 full rule rows). */
long gcn_voxelize_idx_ws_bytes(int N);
int gcn_voxelize_idx(const int64_t *coords, int N, int ncol, int mode, int32_t *input_map, int *M_host,
                     int *maxActive_host, int64_t *output_coords, int32_t *output_map, void *ws, void *stream);

/* ---- host-side SoftGroup routines (the reference runs these on CPU tensors) ---- */

/* SG/src/voxelize/voxelize.cpp:11-39 `voxelize_idx` (host C++ hash dedup).  HOST pointers.
 * Two-call protocol: call with output_coords_host == NULL to obtain *M and *maxActive
 * (input_map_host is filled), then again with buffers of (M,ncol) / (M,maxActive+1). */
int gcn_voxelize_idx_host(const int64_t *coords_host, int N, int ncol, int mode,
                          int32_t *input_map_host, int *M, int *maxActive,
                          int64_t *output_coords_host, int32_t *output_map_host);

/* SG/src/bfs_cluster/bfs_cluster.cpp:122-143 `bfs_cluster` (host BFS).  HOST pointers.
 * Two-call: cluster_idxs_host == NULL -> sizes only. */
int gcn_bfs_cluster_host(const float *class_numpoint_mean_host, const int32_t *ball_query_idxs_host,
                         const int32_t *start_len_host, int N, float threshold, int class_id,
                         int *sumNPoint, int *nCluster, int32_t *cluster_idxs_host,
                         int32_t *cluster_offsets_host);

/* SG/src/hierarchical_aggregation/hierarchical_aggregation.cpp:102-183 (host BFS + split;
 * using_set_aggr additionally runs the fragment->primary absorption of
 * hierarchical_aggregation.cu:22-196 on the host).  HOST pointers; buffers sized for the
 * worst case: idxs (2N,2) i32, offsets (N+1) i32.  Outputs are the MERGED result of
 * HierarchicalAggregation.forward (functions.py:52-72): *sumNPoint rows, *nCluster clusters. */
int gcn_hierarchical_aggregation_host(const int32_t *semantic_label_host, const float *coord_shift_host,
                                      const int32_t *batch_idxs_host, const int32_t *ball_query_idxs_host,
                                      const int32_t *start_len_host, int N, int using_set_aggr,
                                      int32_t *cluster_idxs_host, int32_t *cluster_offsets_host,
                                      int *sumNPoint, int *nCluster);

/* ------------------------------------------------- fused EdgeConv (DGCNN) ------ */

/* Fused replacement of get_graph_feature / get_graph_feature_with_normals (M4:93-161) +
 * Conv2d 1x1 (no bias) + the statistics/extreme half of GroupNorm + LeakyReLU + max over k
 * (M4:463-505).  The grouped (N*k, 2C) x (2C, Cout) contraction runs on MFMA (dtype 1, bf16
 * operands, f32 accumulate) or as an exact k-ordered f32 fmaf chain (dtype 0, parity path).
 * Operands are point-major so that one neighbour is one contiguous row:
 *   dtype 1: x_pm (B,N,Cp) bf16 and w = W' (Cout,2Cp) bf16 from gcn_edgeconv_pack_x / _pack_w,
 *            Cp = gcn_edgeconv_padded_channels(C) (power of two >= 16); Cout in {64,128}, C <= 128, (Cout/G) % 32 == 0
 *   dtype 2: the same kernels on IEEE-half operand images (gcn_edgeconv_pack_x16 / _pack_w16 / gcn_cast_pad16 with
 *            half = 1, v_mfma_f32_32x32x16_f16): BASELINE configs[4] "fp16+MFMA"; 33 <= C <= 256, k <= 128
 *   dtype 0: x_pm (B,N,C) f32, w (Cout,2C) f32 (the reference's own Conv2d weight)
 *   idx    (B,N,k) int64 neighbour ids within the cloud (what knn()/topk returns), 1 <= k <= 255;
 *          ids index the NX rows per cloud of x_pm (NX == N for EdgeConv; a generic grouped block
 *          passes NX = N*k materialised edge rows with identity ids)
 *   ymax, ymin (B,N,Cout) f32: max / min over the k neighbours of the RAW conv output
 *   amax, amin (B,N,Cout) u8 or both NULL: neighbour slot attaining it (lowest slot on ties)
 *   gsum   (B,G,2) f64: per-(cloud, group) sum and sum of squares of the raw conv output over
 *          all N*k*(Cout/G) elements (zeroed by the call).
 * y -> LeakyReLU(gamma*(y-mu)*rstd+beta) is monotone, so max_k f(y_k) = f(max_k y_k) for
 * gamma >= 0 and f(min_k y_k) for gamma < 0, bitwise; gcn_edgeconv_finish applies it.
 * ROUTED mode (gamma_route = the GroupNorm gain (Cout), non-NULL): only the extreme that will be
 * routed is kept -- ymax/amax receive max for gamma_c >= 0 and min otherwise; ymin/amin may be NULL
 * (halves the epilogue work and the output bytes; gcn_edgeconv_finish / gcn_route_bwd accept NULL ymin). */
int gcn_edgeconv_padded_channels(int C);
int gcn_edgeconv_pack_x(const float *x_cm, int B, int C, int N, void *x_pm_bf16, float *x_pm_f32,
                        void *stream);
int gcn_edgeconv_pack_w(const float *w, int Cout, int C, void *wp_bf16, void *stream);
/* the same with the 16-bit type chosen by the caller: half = 0 bf16, half = 1 IEEE half (round to nearest even) */
int gcn_edgeconv_pack_x16(const float *x_cm, int B, int C, int N, void *x_pm_16, float *x_pm_f32, int half, void *stream);
int gcn_edgeconv_pack_w16(const float *w, int Cout, int C, void *wp_16, int half, void *stream);
int gcn_edgeconv_fwd(const void *x_pm, const void *w, const int64_t *idx, int dtype, int B, int N,
                     int NX, int C, int k, int Cout, int G, const float *q, float *ymax, float *ymin,
                     uint8_t *amax, uint8_t *amin, double *gsum, const float *gamma_route, void *stream);

/* Centre term of the grouped contraction (dtype 1): y[n,j] = W1.x_j + (W2 - W1).x_n, and the second summand is the
 * same for the k rows of point n.  q (rows, Cout) f32 = x_pm_bf16 (rows, Cp) . (W2 - W1)^T from the [W1 | W2 - W1]
 * image of gcn_edgeconv_pack_w -- a per-point (rows x C x Cout) MFMA GEMM, k-fold cheaper than the grouped part.
 * gcn_edgeconv_fwd takes it as `q` (NULL = no centre term, e.g. a generic grouped block with W' = [W | 0]) and adds it
 * after the reduction over k (adding a per-point constant commutes with max/min bitwise; the GroupNorm sums are
 * corrected in closed form), so the matrix cores contract K = Cp instead of 2 Cp.  Needed for k <= 128 only. */
int gcn_edgeconv_center(const void *x_pm_bf16, const void *wp_bf16, long rows, int C, int Cout, float *q,
                        void *stream);
int gcn_edgeconv_center_f16(const void *x_pm_f16, const void *wp_f16, long rows, int C, int Cout, float *q, void *stream);   /* dtype 2 */

/* Point-major operand preparation when activations are already (rows, C) f32: cast to bf16 and zero-pad
 * the channel axis to gcn_edgeconv_padded_channels(C) (no transpose, unlike gcn_edgeconv_pack_x). */
int gcn_cast_pad_bf16(const float *x_pm, long rows, int C, void *x_pm_bf16, void *stream);
int gcn_cast_pad16(const float *x_pm, long rows, int C, void *x_pm_16, int half, void *stream);

/* GroupNorm(G, Cout, eps) + LeakyReLU(slope) on the routed extreme:
 *   out_cm (B,Cout,N) f32 (the reference's layout) and/or out_pm (B,N,Cout); either may be NULL
 *   mean_rstd (B,G,2) f32 (may be NULL): statistics for backward.
 *   out_pm_bf16 (may be NULL): the point-major result once more, rounded to bf16, rows bf16_pitch elements apart
 *   (>= Cout) -- a column slice of the consumer's concatenated input (the encoder's cat(x1,x2,x3), M4:507), so that
 *   neither the cat nor the autocast conversion runs as a separate pass. */
int gcn_edgeconv_finish(const float *ymax, const float *ymin, const double *gsum,
                        const float *gamma, const float *beta, int B, int N, int k, int Cout,
                        int G, float eps, float slope, float *out_cm, float *out_pm,
                        float *mean_rstd, void *out_pm_bf16, int bf16_pitch, void *stream);

/* Graph aggregations used by the EdgeConv backward (no counterpart kernel in the reference: its
 * autograd walks the materialised (B,2C,N,k) tensor).  x_pm (B,N,C) f32, idx (B,N,k) int64.
 *   gcn_neighbor_sum: s[b,n,:] = sum_j x[b, idx[b,n,j], :]                      (gather)
 *   gcn_reverse_sum : r[b,m,:] = sum_{(n,j): idx[b,n,j]==m} x[b,n,:], indeg[b,m] = #(n,j) (destination-
 *                     partitioned 64-bit fixed-point LDS accumulation: no global atomics, bitwise
 *                     reproducible; both outputs fully written; indeg (B,N) f32 may be NULL, or r may
 *                     be NULL (in-degrees only: no rows are read);
 *                     ws: gcn_reverse_sum_ws_bytes(B,N,C,k) bytes of device scratch, 16-B aligned).
 *                     C in {64,128}, N <= 16384, N % (16384/C) == 0: edges are filed per destination
 *                     partition, sorted by destination row in LDS and the source rows GATHERED with 64-bit
 *                     integer adds in registers (csrc/rsum.hip) -- same guarantees, no shared accumulators. */
int gcn_neighbor_sum(const float *x_pm, const int64_t *idx, int B, int N, int C, int k, float *s,
                     void *stream);
long gcn_reverse_sum_ws_bytes(int B, int N, int C, int k);
int gcn_reverse_sum(const float *x_pm, const int64_t *idx, int B, int N, int C, int k, float *r,
                    float *indeg, void *ws, void *stream);

/* Fused conv + extreme/statistics half of OFFSET_PRED_MODULE's grouped block (M4:425-446): every
 * point has k edges to a fixed set of NK key points; the KPAM-scaled Conv2d(131->128) output is
 *   y[b,n,j,:] = att[b,n,j] * (U[b, kidx[b,n,j], :] - V[b,n,:])
 * (U = Wf.f_key + Wp.p_key, V = Wp.p_n; the conv is linear).  att (B,N,k) f32, kidx (B,N,k) int64 in
 * [0,NK), U (B,NK,Cout), V (B,N,Cout).  Outputs as gcn_edgeconv_fwd; feed them to gcn_edgeconv_finish.
 * gamma_route (Cout) f32 or NULL: as in gcn_edgeconv_fwd, the GroupNorm scale whose sign decides which extreme
 * the block will use -- only that one is kept (in ymax / amax; ymin and amin may then be NULL and are not written). */
int gcn_keyedge_fwd(const float *att, const int64_t *kidx, const float *U, const float *V, int B, int N,
                    int k, int NK, int Cout, int G, float *ymax, float *ymin, uint8_t *amax,
                    uint8_t *amin, double *gsum, const float *gamma_route, void *stream);

/* Backward of gcn_keyedge_fwd given the routed/affine decomposition of the conv-output gradient
 *   dy[b,n,j,c] = coef[b,n,c]*[j == jsel[b,n,c]] + Ac[b,c] + Bc[b,c]*y[b,n,j,c]
 * (coef (B,N,Cout) f32, jsel (B,N,Cout) int64 neighbour slot, Ac/Bc (B,Cout) f32) and the caller's
 * X (B,N,NK) = (V o Bc) . U^T (one GEMM).  Writes datt (B,N,k) and dV (B,N,Cout) complete, and the pieces of
 *   dU = dUsp + Ac (x) T1 + Bc o (U o T2 - A2^T V):
 * A2 (B,N,NK) dense incidence sum_j att^2 [kidx = m], dUsp (B,NK,Cout), T12 (B,2,NK) = column sums of the
 * att / att^2 incidences (dUsp, T12 zeroed by the call).  k <= 64. */
int gcn_keyedge_bwd(const float *att, const int64_t *kidx, const float *U, const float *V,
                    const float *coef, const int64_t *jsel, const float *Ac, const float *Bc, const float *X,
                    int B, int N, int k, int NK, int Cout, float *datt, float *dV, float *A2, float *dUsp,
                    float *T12, void *stream);

/* Normal-feature EdgeConv of M4:164-205,575-577,691-693 without its (B,7,N,k) edge tensor: the feature
 * [clamp(n_i.n_j, +-0.99), n_j - n_i, n_i] is rebuilt in registers from pts (B,N,6) point-major [xyz, normal]
 * and idx (B,N,k) int64; W (Cout,7) f32 = the Conv2d(7->Cout,1x1) weight.  Outputs as gcn_edgeconv_fwd
 * (ymax/ymin (B,N,Cout), amax/amin u8, gsum (B,G,2) f64; feed them to gcn_edgeconv_finish).  k <= 256.
 * gamma_route (Cout) f32 or NULL: routed mode as in gcn_edgeconv_fwd / gcn_keyedge_fwd (ymin, amin may be NULL). */
int gcn_normal_edge_fwd(const float *pts, const int64_t *idx, const float *W, int B, int N, int k, int Cout,
                        int G, float *ymax, float *ymin, uint8_t *amax, uint8_t *amin, double *gsum,
                        const float *gamma_route, void *stream);

/* Weight-gradient pieces of that block for dy = coef*[j == jsel] + Ac + Bc*y (the block's inputs are the
 * cloud itself and carry no gradient): dWsp (B,Cout,7) = per-cloud sum coef * ef[jsel], esum (B,7) = sum_{n,j} ef,
 * gram (B,7,7) = sum_{n,j} ef ef^T (all zeroed by the call); the caller finishes
 *   dW = sum_b dWsp_b + Ac^T esum + sum_b Bc_b o (W gram_b).   Cout <= 128. */
int gcn_normal_edge_bwd(const float *pts, const int64_t *idx, const float *coef, const int64_t *jsel, int B,
                        int N, int k, int Cout, float *dWsp, float *esum, float *gram, void *stream);

/* Row-wise top-k, largest first, of R short rows: x (R,NK) f32 (dtype 0) or bf16 (dtype 1), NK <= 128,
 * k <= min(64, NK) -> vals (R,k) f32, idx (R,k) int64; ties go to the lower column.  Replaces the
 * `torch.topk(dist, k)` over the key-point similarities of OFFSET_PRED_MODULE (M4:421-422). */
int gcn_topk_rows(const void *x, int dtype, long R, int NK, int k, float *vals, int64_t *idx, void *stream);

/* ------------------------- fused pieces of the closed-form grouped-block backward ------ */

/* One pass over (B,N,Cout): selected extreme (max for gamma >= 0, else min) -> yhat, z, routed
 * gradient gz = dout * LeakyReLU'(z).  Outputs: coef = rstd*gamma*gz (B,N,Cout); jsel (slot, may be
 * NULL), msel = idx[b,n,jsel] (may be NULL), dsp[b, msel, c] += coef (may be NULL; zeroed by the
 * call), dgamma/dbeta (Cout) and S (B,G,2) f64 = [sum gamma*gz, sum gamma*gz*yhat] (zeroed by the call).
 * idx (B,N,k) int64 may be NULL when neither msel nor dsp is wanted.  Ac/Bc (B,Cout) f32 (both or
 * neither): the affine GroupNorm terms of dy = coef*[j==jsel] + Ac + Bc*y for count_per_group =
 * (Cout/G)*N*k conv outputs per group, evaluated in double from S and mean_rstd.
 * dsp_ws: NULL, or gcn_route_bwd_ws_bytes(B,N,Cout) bytes of scratch (16-byte aligned; its first 4 bytes are zeroed by
 * the call -- place it right behind dbeta and the fill merges with the other accumulators').  With it, and Cout in
 * {64,128}, N <= 65536, N*Cout % 65536 == 0, dsp is built by a destination-partitioned LDS scatter in 64-bit fixed
 * point (bitwise reproducible, every element written, no zero fill) instead of B*N*Cout global f32 atomics.
 * part_ws: NULL (dgamma/dbeta/S are zeroed and accumulated with atomics), or gcn_route_bwd_part_bytes(B,N,Cout,G) bytes
 * of 8-byte aligned scratch: workgroups write partial sums there and a second small kernel folds them in a fixed order
 * (no contention on the few shared addresses: ~25 us per launch; dgamma/dbeta/S are then written, not accumulated). */
long gcn_route_bwd_part_bytes(int B, int N, int Cout, int G);
long gcn_route_bwd_ws_bytes(int B, int N, int Cout);
int gcn_route_bwd(const float *dout_pm, const float *ymax, const float *ymin, const uint8_t *amax,
                  const uint8_t *amin, const float *gamma, const float *beta, const float *mean_rstd,
                  const int64_t *idx, int B, int N, int k, int Cout, int G, float slope, float *coef,
                  int64_t *jsel, int64_t *msel, float *dsp, float *dgamma, float *dbeta, double *S,
                  double count_per_group, float *Ac, float *Bc, void *dsp_ws, void *part_ws, void *stream);

/* Weight gradient of the fused EdgeConv block from the pieces above, all row reductions in one pass on the
 * f32 matrix cores:  dW (Cout,2C) = [dW1 - dWd | dWd] with dWd = D2^T x and
 *   dW1 = Dsp^T x + sum_b Ac_b (x) sum_n s_b + sum_b Bc_b o (W1 x_b^T diag(indeg_b) x_b + Wd x_b^T s_b),
 * W = [W1 | W2] (Cout,2C) the conv weight (Wd = W2 - W1), x/s (B,N,C), dsp/d2 (B,N,Cout), indeg (B,N).
 * ws: gcn_edge_wgrad_ws_floats(B,C,Cout) floats of scratch.  C <= 16 or C == 64; Cout in {64,128}. */
long gcn_edge_wgrad_ws_floats(int B, int C, int Cout);
int gcn_edge_wgrad(const float *x_pm, const float *s_pm, const float *dsp, const float *d2, const float *indeg,
                   const float *W, const float *Ac, const float *Bc, int B, int N, int C, int Cout, float *dW,
                   float *ws, void *stream);

/* D2 = coef + k*A + B*(SW + k*XW);  D1 = dsp + indeg*(A + B*P1) + B*RW   (all (B,N,Cout) f32;
 * A, B (B,Cout); indeg (B,N)): per-point sums of dy over outgoing / incoming edges.  D1 may be NULL (a first layer
 * whose input carries no gradient needs D2 only; dsp, indeg, P1, RW are then not read). */
int gcn_edge_combine(const float *coef, const float *dsp, const float *indeg, const float *Ac,
                     const float *Bc, const float *P1, const float *SW, const float *XW, const float *RW,
                     int B, int N, int k, int Cout, float *D1, float *D2, void *stream);

/* ------------------------------------------- GroupNorm(+ReLU), point-major (B,N,C) ------ */

/* Replaces the `F.relu(self.bnX(self.convX(x)))` normalisation of the per-point heads (M4:644-726;
 * torch GroupNorm on (B,C,N)) for POINT-MAJOR activations x (B,N,C), dtype 0 = f32 / 1 = bf16
 * (statistics always f32/f64).  y = [ReLU]((x - mean_g) * rstd_g * gamma_c + beta_c), statistics per
 * (sample, group) over N*(C/G) elements.  Constraints: (C/G) % 4 == 0; C/4 divides 256 or is a
 * multiple of 256.  mean_rstd (B,G,2) f32 is written for backward; gsum_ws (B,G,2) f64 workspace. */
int gcn_gn_fwd(const void *x, int dtype, const float *gamma, const float *beta, int B, int N, int C,
               int G, float eps, int relu, void *y, float *mean_rstd, double *gsum_ws, void *stream);

/* Backward of the above: dx (same dtype as x), dgamma/dbeta (C) f32 (written),
 * s_ws: gcn_gn_bwd_ws_bytes(B,N,C,G) bytes of scratch, 8-byte aligned (the (B,G,2) f64 sums, then per-workgroup
 * partials that a second small kernel folds in a fixed order: no same-address atomics, deterministic, no zero fill). */
long gcn_gn_bwd_ws_bytes(int B, int N, int C, int G);
int gcn_gn_bwd(const void *dy, const void *x, int dtype, const float *gamma, const float *beta,
               const float *mean_rstd, int B, int N, int C, int G, int relu, void *dx, float *dgamma,
               float *dbeta, double *s_ws, void *stream);

/* GroupNorm(+ReLU) followed by the max over the points of each sample (M4:510-513: the 1024-channel global
 * feature), without writing the (B,N,C) activation: out_max (B,C) f32, out_arg (B,C) int64 = the row of the
 * maximum (lowest row on ties); mean_rstd as gcn_gn_fwd; gsum_ws (B,G,2) f64 and best_ws (B,C) u64 scratch.
 * With dtype 1 the maxima are rounded to bf16 exactly as the materialised tensor would have been. */
int gcn_gn_max_fwd(const void *x, int dtype, const float *gamma, const float *beta, int B, int N, int C, int G,
                   float eps, int relu, float *out_max, int64_t *out_arg, float *mean_rstd, double *gsum_ws,
                   void *best_ws, void *stream);

/* Backward of gcn_gn_max_fwd for dout (B,C) f32 routed to the rows out_arg (B,C): the upstream gradient is zero
 * except at one row per (sample, channel), so dx = rstd*gamma*g*[n == arg] + A[b,g] + Bx[b,g]*x with per-(sample,
 * group) constants from sums over B*C values: one dense affine pass (read x, write dx) + B*C corrections instead of
 * a zero fill, a scatter and the generic two-pass gcn_gn_bwd.  dx same dtype as x; dgamma/dbeta (C) f32 (written);
 * ws: gcn_gn_max_bwd_ws_floats(B,C,G) floats, 8-byte aligned. */
long gcn_gn_max_bwd_ws_floats(int B, int C, int G);
int gcn_gn_max_bwd(const void *x, int dtype, const float *gamma, const float *beta, const float *mean_rstd,
                   const float *dout, const int64_t *arg, int B, int N, int C, int G, int relu, void *dx,
                   float *dgamma, float *dbeta, float *ws, void *stream);

/* Parameter head epilogue (M4:664-676): p (R,22) f32 rows; the triples 4:7, 8:11, 15:18 are divided by
 * (their L2 norm + 1e-12), the other columns pass through -- one kernel instead of the slice / norm / div / cat
 * chain; _bwd is its vector-Jacobian product (grad_in (R,22) fully written). */
int gcn_param_normalise_fwd(const float *p, long R, float *out, void *stream);
int gcn_param_normalise_bwd(const float *p, const float *grad_out, long R, float *grad_in, void *stream);

/* y = x / |x| over the last dimension of x (R,C) f32 -- the feature normalisation of cos_dist
 * (models/dgcnn-hais-concat-direct-4.py:326-342: f / f.norm(dim=-1, keepdim=True), no epsilon) -- and its gradient
 * dx = (g - y (y.g)) / |x|. */
int gcn_row_normalise_fwd(const float *x, long R, int C, float *y, void *stream);
int gcn_row_normalise_bwd(const float *x, const float *grad_out, long R, int C, float *grad_in, void *stream);

/* L = sum over tensors t of mean(v_t^2), and grad_t = v_t * (2 / numel_t) * grad_loss[0], one launch each way for up to
 * 8 f32 / bf16 tensors (torch: one reduction per tensor forward, two scaling passes backward).  No reference
 * counterpart: it is the synthetic objective bench.py puts on the hot path's five outputs (M4:634-747 returns them to
 * loss code that is out of scope), kept in the library so that the timed step has no torch reductions in it.
 *   v, grad: HOST arrays of nt device pointers; numel, is_bf16: HOST arrays (element counts >= 1; 0 = f32, 1 = bf16).
 *   part: gcn_multi_mean_square_ws_chunks(numel, nt) doubles of device scratch; done: one device uint32, zero before
 *   the first call (every call leaves it zero); loss: one device float, written; grad_loss: one device float.
 *   Partial sums are folded in a fixed order (bit-reproducible). */
int gcn_multi_mean_square_ws_chunks(const long *numel, int nt);
int gcn_multi_mean_square_fwd(const void *const *v, const long *numel, const int *is_bf16, int nt, double *part,
                              unsigned int *done, float *loss, void *stream);
int gcn_multi_mean_square_bwd(const void *const *v, void *const *grad, const long *numel, const int *is_bf16, int nt,
                              const float *grad_loss, void *stream);

/* gcn_gn_fwd's second half alone: the (B,G,2) f64 sums and sums of squares are already in `gsum` -- written by the
 * epilogue of the GEMM that produced x (gcn_gemm_bf16), so the statistics pass over x is skipped. */
int gcn_gn_apply(const void *x, int dtype, const double *gsum, const float *gamma, const float *beta, int B,
                 int N, int C, int G, float eps, int relu, void *y, float *mean_rstd, void *stream);

/* ---------------------------------- per-point 1x1 convolutions of the heads (bf16 MFMA GEMMs) ------ */

/* The Conv1d(kernel 1) layers of the heads (M4:556-603,644-699,713) on POINT-major activations (csrc/gemm.hip):
 *   out (M,N) = A (M,K) . W (N,K)^T + bias      A, W bf16 (both contiguous along K), out bf16 (out_f32 = 0) or f32,
 * K % 16 == 0 (pad with zero columns), W holds Np >= N rows (rows N..Np-1 zero: pad N up to a multiple of 32 so the
 * tile loads stay in bounds without reading other weights).  The input gradient is the same call with A = dY and
 * W = the transposed weight.  gsum (M/rows_per_cloud, G, 2) f64 or NULL: per (cloud, group) sum / sum of squares of
 * the f32 results for the GroupNorm that follows (needs (N/G) % 32 == 0, rows_per_cloud % 128 == 0, and stats_ws =
 * gcn_gemm_stats_ws_bytes(M, N) bytes of scratch for the per-wave partial sums); feed it to gcn_gn_apply.
 * gcn_gemm_wgrad_bf16: dW (N,K) f32 = dY (M,N)^T . X (M,K), the contraction running down the rows of both row-major
 * operands (fragments by ds_read_b64_tr_b16; M is cut into row slices whose partial tiles go to ws,
 * gcn_gemm_wgrad_ws_bytes(M, N, K) bytes, and are added in slice order: no atomics, the same bits every run);
 * N % 8 == 0, K % 8 == 0, dW / db 16-byte aligned and fully overwritten.  db (N) f32 or NULL: the bias gradient
 * (column sums of dY) from the same pass. */
long gcn_gemm_stats_ws_bytes(long M, int N);
int gcn_gemm_bf16(const void *A, const void *W, const float *bias, void *out, int out_f32, long M, int N, int Np,
                  int K, double *gsum, void *stats_ws, int rows_per_cloud, int G, void *stream);
long gcn_gemm_wgrad_ws_bytes(long M, int N, int K);
int gcn_gemm_wgrad_bf16(const void *dY, const void *X, long M, int N, int K, float *dW, float *db, void *ws, void *stream);

/* Weight (and bias) gradient of a NARROW 1x1 layer: dW (N,K) f32 = dY (M,N)^T . X (M,K), db (N) f32 = column sums of dY
 * (db may be NULL), for 1 <= N <= 32 and 1 <= K <= 1024 of any value -- the 10-, 22- and 3-wide outputs of the heads
 * (M4:661,678,448) and KPAM's 30x30 layers (M4:351-373), which gcn_gemm_wgrad_bf16 does not take.  dY and X are
 * contiguous row-major, each bf16 (flag 1) or f32 (flag 0).  One streaming pass; per-workgroup partial tiles in ws
 * (gcn_wgrad_narrow_ws_bytes bytes, 16-byte aligned) are added in a fixed order.  dW / db are fully overwritten.
 * A layer with a narrow INPUT instead (K <= 32 < N) is the same call with the operands swapped: it returns dW^T. */
int gcn_wgrad_narrow_supported(long M, int N, int K);
long gcn_wgrad_narrow_ws_bytes(long M, int N, int K);
int gcn_wgrad_narrow(const void *dY, int dy_bf16, const void *X, int x_bf16, long M, int N, int K, float *dW, float *db,
                     void *ws, void *stream);

/* ------------------------------------------------------------- attention stacks ------ */

/* Fused scaled-dot-product attention forward (online softmax; the (Lq x Lk) score matrix never
 * reaches HBM).  Replaces the unfused einsum -> softmax -> einsum of models/transformer.py:52-69
 * (which materialises (b,h,n,n)) and the attention core of nn.MultiheadAttention used by
 * models/query_decoder.py:12,54.  q (BH,Lq,D), k/v (BH,Lk,D) f32 contiguous (BH = batch*heads);
 * mask: optional bytes, 1 = key masked out, (Lq,Lk) shared by all BH or (BH,Lq,Lk) if mask_per_bh;
 * out (BH,Lq,D) = softmax(scale * q.k^T) v;  lse (BH,Lq) optional log-sum-exp.  D in {8,16,32,64}. */
int gcn_attention_fwd(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                      int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *stream);

/* The same contraction on the bf16 matrix cores (flash-style, f32 accumulation, f32 in/out) for the
 * long sequences of BASELINE config 5 (transformer.py:52-69 over n = 16384 points; "fp16+MFMA").
 * D in {32,64}.  ws: device workspace of gcn_attention_ws_bytes(BH,Lq,Lk,D) bytes, 16-B aligned (bf16
 * operand images; contents are scratch).  Results agree with the f32 entry point to bf16 operand
 * rounding (~1e-2 relative), not to 1e-4: the exact kernel above remains the parity path. */
long gcn_attention_ws_bytes(int BH, int Lq, int Lk, int D);
int gcn_attention_fwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                           int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *ws,
                           void *stream);

/* Backward of gcn_attention_fwd_bf16 (what autograd derives from transformer.py:52-69 /
 * nn.MultiheadAttention): recomputes the probabilities from lse, two kernels (dQ by query block, dK/dV by
 * key block -- no atomics, deterministic), seven bf16 MFMA products per tile pair.  out/lse are the
 * forward's results, dout (BH,Lq,D) the incoming gradient; dq (BH,Lq,D), dk/dv (BH,Lk,D) f32 are fully
 * written.  Same ws contract. */
int gcn_attention_bwd_bf16(const float *q, const float *k, const float *v, const float *out, const float *dout,
                           const float *lse, const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, int D,
                           float scale, float *dq, float *dk, float *dv, void *ws, void *stream);

/* The same two entry points with IEEE half operands (v_mfma_f32_32x32x16_f16) -- the "fp16+MFMA" type BASELINE config 5
 * names: three more significand bits than bf16 (results ~8x closer to the f32 kernel), values below 6e-8 flush.
 * Same arguments, workspace and layout. */
int gcn_attention_fwd_f16(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                          int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *ws,
                          void *stream);
int gcn_attention_bwd_f16(const float *q, const float *k, const float *v, const float *out, const float *dout,
                          const float *lse, const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, int D,
                          float scale, float *dq, float *dk, float *dv, void *ws, void *stream);

/* One launch for many f32 -> bf16 (round to nearest even) copies: the per-step refresh of the low-precision parameter
 * copies the autocast path of the reference (torch.autocast casts each weight at each use) needs.
 *   segs_dev: device array of nseg records of six int64 {src f32*, dst bf16*, cols, pitch, first, count}: one
 *   workgroup converts elements [first, first+count) of a row-major tensor with `cols` columns into an image whose
 *   rows are `pitch` elements apart (pitch == cols for plain copies). */
int gcn_multi_cast_bf16(const void *segs_dev, int nseg, void *stream);

/* Adam (torch.optim.Adam's rule: exp_avg lerp, exp_avg_sq, bias corrections, p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps),
 * optional L2 weight decay folded into the gradient) over flat f32 buffers of n elements, 16-byte aligned.  state: 4
 * floats on the device, zero before the first step: [0] the step count (incremented by the call: graph-capturable),
 * [1], [2] the bias corrections 1 - beta^t, computed in double as torch.optim.Adam does, [3] the step count as a 32-bit
 * integer (bit pattern).  The reference's trainer: option_new.py:83-90. */
int gcn_adam_flat(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, float *state, void *stream);

/* Optional pre-zeroed scratch arena.  Many entry points zero small accumulators they are handed ("zeroed by the call"):
 * ~45 fills of a few hundred bytes to a few MB per training step, ~4.5 us of GPU time each.  A caller that carves those
 * buffers out of ONE device allocation, zeroes the used part of it once per step and never hands the same bytes out
 * twice within a step may register the allocation here: a span that lies inside [base, base + bytes) is then trusted to
 * be zero and its fill is skipped (gcanet_amd/layers.py:ZeroArena).  bytes == 0 unregisters.  Process-wide setting, not
 * thread safe; every other pointer is zeroed by the call exactly as before.
 * THE CONTRACT IS THE CALLER'S: the library cannot tell a clean arena span from a dirty one.  An accumulator inside the
 * registered range that was not zeroed since it was last handed out yields wrong sums WITHOUT an error.  Register an
 * arena only if (1) one owner hands out its bytes, (2) the used prefix is zeroed once per step before the first call,
 * (3) no span is used across steps and (4) results that outlive the step are never arena-backed (the Python layer
 * clones the one tensor it used to return from the arena, the EdgeConv group sums).  Without a registered arena every
 * "zeroed by the call" buffer is zeroed by the call. */
int gcn_zero_arena_register(void *base, long bytes);

#ifdef __cplusplus
}
#endif
#endif /* GCANET_HIP_H */
