// softgroup.hip -- gfx950 kernels behind the softgroup.ops boundary
// (reference: softgroup/ops/src/{voxelize,bfs_cluster,bfs_cluster_easy,sec_mean,roipool,
// cal_iou_and_masklabel}/*.cu).  All of these are HBM-bound integer/byte work: the design
// rules that matter are coalescing, enough waves in flight and no host round trips.
//
//   voxelize fp/bp     one wave per voxel row, lanes across planes (256-B coalesced rows),
//                      rule-book entries are wave-uniform scalar loads
//   ballquery_batch_p  count -> exclusive scan -> fill (deterministic CSR in point order;
//                      no 3000-int per-thread scratch array as in bfs_cluster.cu:30)
//   sec_* / avg pool   one wave per (segment, 64-plane chunk)
//   mask IoU           one workgroup per proposal: LDS histogram of instance labels instead
//                      of the reference's O(P*I*len) serial intersection loops
#include "common.h"

namespace gcn {

__device__ __forceinline__ float sqdist3s(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  float t = dx * dx;
  t = fmaf(dy, dy, t);
  t = fmaf(dz, dz, t);
  return t;
}

// ---------------------------------------------------------------- voxelize
// voxelize.cu:9-25: sequential sum over the rule list (single thread per (row,plane) in the
// reference, so its atomicAdd order is fixed) -> bit-exact restatement.
__global__ __launch_bounds__(256) void voxelize_fp_kernel(int M, int maxActive, int C,
                                                          const float *__restrict__ feats,
                                                          float *__restrict__ out,
                                                          const int32_t *__restrict__ rules, int average) {
  const int lane = lane_id();
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int32_t *r = rules + (long)row * (maxActive + 1);
  const int nActive = r[0];
  const float mult = (average && nActive > 0) ? 1.f / (float)nActive : 1.f;
  for (int p0 = 0; p0 < C; p0 += 64) {
    const int p = p0 + lane;
    float acc = 0.f;
    if (p < C) {
      for (int i = 1; i <= nActive; ++i) acc += mult * feats[(long)r[i] * C + p];
      out[(long)row * C + p] = acc;
    }
  }
}

// voxelize.cu:38-54 (accumulates into d_feats like the reference; 256-B contiguous atomics)
__global__ __launch_bounds__(256) void voxelize_bp_kernel(int M, int maxActive, int C,
                                                          const float *__restrict__ d_out,
                                                          float *__restrict__ d_feats,
                                                          const int32_t *__restrict__ rules, int average) {
  const int lane = lane_id();
  const int row = blockIdx.x * 4 + wave_id();
  if (row >= M) return;
  const int32_t *r = rules + (long)row * (maxActive + 1);
  const int nActive = r[0];
  const float mult = (average && nActive > 0) ? 1.f / (float)nActive : 1.f;
  for (int p0 = 0; p0 < C; p0 += 64) {
    const int p = p0 + lane;
    if (p < C) {
      const float g = mult * d_out[(long)row * C + p];
      for (int i = 1; i <= nActive; ++i) atomicAdd(&d_feats[(long)r[i] * C + p], g);
    }
  }
}

// ---------------------------------------------------------------- ball query (batched, CSR)
// bfs_cluster.cu:18-77 / bfs_cluster_easy.cu:15-66.  FILL == false: count pass;
// FILL == true: write neighbour ids at start_len[p][0] (truncated at thre).
template <bool FILL>
__global__ __launch_bounds__(256) void ballquery_kernel(int n, long thre, float radius2, int cap,
                                                        const float *__restrict__ xyz,
                                                        const int32_t *__restrict__ batch_idxs,
                                                        const int32_t *__restrict__ batch_offsets,
                                                        const float *__restrict__ adj_inst, float thr_inst,
                                                        const float *__restrict__ adj_para, float thr_para,
                                                        int32_t *__restrict__ idx, int32_t *__restrict__ start_len,
                                                        int32_t *__restrict__ counts) {
  const int lane = lane_id();
  const int p = blockIdx.x * 4 + wave_id();
  if (p >= n) return;
  const float ox = xyz[p * 3], oy = xyz[p * 3 + 1], oz = xyz[p * 3 + 2];
  const int bi = batch_idxs[p];
  const int start = batch_offsets[bi], end = batch_offsets[bi + 1];
  long s0 = 0;
  int limit = cap;  // entries this point may write
  if (FILL) {
    s0 = start_len[p * 2];
    if (s0 >= thre) return;
    const int cnt = start_len[p * 2 + 1];
    limit = (s0 + cnt >= thre) ? (int)(thre - s0) : cnt;
  }
  int cnt = 0;
  for (int base = start; base < end && cnt < limit; base += 64) {
    const int k = base + lane;
    bool hit = false;
    if (k < end) {
      const float d2 = sqdist3s(ox, oy, oz, xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2]);
      hit = d2 < radius2;
      if (hit && adj_inst) hit = (adj_inst[(long)p * n + k] > thr_inst) && (adj_para[(long)p * n + k] > thr_para);
    }
    const unsigned long long mask = __ballot(hit);
    if (FILL) {
      const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < limit) idx[s0 + slot] = k;
    }
    cnt += __popcll(mask);
  }
  if (!FILL) counts[p] = cnt < cap ? cnt : cap;
}

// single-workgroup exclusive scan of counts[0..n) -> start_len[:,0], start_len[:,1]=count,
// counts[n] = total
__global__ __launch_bounds__(1024) void scan_counts_kernel(int n, int32_t *__restrict__ counts,
                                                           int32_t *__restrict__ start_len) {
  __shared__ int part[1024];
  const int tid = threadIdx.x;
  const int chunk = (n + 1023) / 1024;
  const int lo = tid * chunk, hi = min(lo + chunk, n);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += counts[i];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int i = lo; i < hi; ++i) {
    const int c = counts[i];
    start_len[i * 2] = run;
    start_len[i * 2 + 1] = c;
    run += c;
  }
  if (tid == 1023) counts[n] = part[1023];
}

// ---------------------------------------------------------------- ball query on a uniform grid
// The easy form (no adjacency matrices) at BASELINE config 4 (n = 100 000) is 10^10 pair tests by brute force
// (11 ms); with cells of edge >= radius only the 27 surrounding cells hold candidates (~75 instead of 100 000).
//   grid_setup   bounding box -> cell edge (radius * 1.001, coarsened x1.26 until B*dx*dy*dz <= max_cells) and dims,
//                all on the device: no host round trip
//   cell_count / scan / cell_fill   counting sort of the points by (batch, z, y, x) cell (x fastest: the three
//                x-neighbours of a (y,z) row are ONE contiguous range -> 9 ranges per point)
//   ballquery_grid_kernel<FILL>     wave per point: candidates of the 9 ranges 64 at a time, same distance
//                expression and comparison as the brute-force kernel; hits are compacted into an LDS buffer and
//                sorted ascending (the reference's lists are in ascending index order and truncated at `cap` in
//                that order).  A point whose 27 cells hold more than 1024 candidates takes the brute-force scan.
// The 1.001 margin keeps |floor| differences of neighbours <= 1 under float rounding (dims <= 1024).
struct BqGrid {            // lives at the head of the device workspace
  float minx, miny, minz, inv;
  int dx, dy, dz, ncell;   // per batch dims; ncell = B*dx*dy*dz
  unsigned int bmin[3], bmax[3];   // ordered-uint bounding box accumulators
};

__device__ __forceinline__ unsigned int f2ord(float f) {
  const unsigned int u = __float_as_uint(f);
  return u ^ ((unsigned int)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

__global__ __launch_bounds__(1024) void bq_bbox_kernel(int n, const float *__restrict__ xyz, BqGrid *g) {
  __shared__ float red[16][6];
  float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = xyz[i * 3 + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
    if (lane_id() == 0) { red[threadIdx.x >> 6][a] = mn[a]; red[threadIdx.x >> 6][3 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {          // same-address atomics serialise (~0.45 us each): one per block and component
    const int a = threadIdx.x;
    float v = red[0][a];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
    if (a < 3) atomicMin(&g->bmin[a], f2ord(v)); else atomicMax(&g->bmax[a - 3], f2ord(v));
  }
}

__global__ void bq_setup_kernel(BqGrid *g, float radius, int B, int max_cells) {
  const float mnx = ord2f(g->bmin[0]), mny = ord2f(g->bmin[1]), mnz = ord2f(g->bmin[2]);
  const float ex = ord2f(g->bmax[0]) - mnx, ey = ord2f(g->bmax[1]) - mny, ez = ord2f(g->bmax[2]) - mnz;
  float cell = radius * 1.001f;
  int dx, dy, dz;
  for (;;) {
    dx = (int)fminf(ex / cell, 1.0e6f) + 1; dy = (int)fminf(ey / cell, 1.0e6f) + 1; dz = (int)fminf(ez / cell, 1.0e6f) + 1;
    if (dx <= 1024 && dy <= 1024 && dz <= 1024 && (double)B * dx * dy * dz <= (double)max_cells) break;
    cell *= 1.26f;
  }
  g->minx = mnx; g->miny = mny; g->minz = mnz; g->inv = 1.f / cell;
  g->dx = dx; g->dy = dy; g->dz = dz; g->ncell = B * dx * dy * dz;
}

__device__ __forceinline__ void bq_cell_of(const BqGrid *g, float x, float y, float z, int &cx, int &cy, int &cz) {
  cx = min(max((int)((x - g->minx) * g->inv), 0), g->dx - 1);
  cy = min(max((int)((y - g->miny) * g->inv), 0), g->dy - 1);
  cz = min(max((int)((z - g->minz) * g->inv), 0), g->dz - 1);
}

__global__ void bq_cell_count_kernel(int n, const float *__restrict__ xyz, const int32_t *__restrict__ batch_idxs,
                                     const BqGrid *__restrict__ g, int32_t *__restrict__ cell_of_pt, int32_t *__restrict__ cell_cnt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  bq_cell_of(g, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], cx, cy, cz);
  const int c = ((batch_idxs[i] * g->dz + cz) * g->dy + cy) * g->dx + cx;
  cell_of_pt[i] = c;
  atomicAdd(cell_cnt + c, 1);
}

// packed (optional): (x, y, z, id) rows in cell order, so that a candidate range is one coalesced 16-byte-per-lane load
__global__ void bq_cell_fill_kernel(int n, const int32_t *__restrict__ cell_of_pt, int32_t *__restrict__ cursor,
                                    int32_t *__restrict__ sorted, const float *__restrict__ xyz, float4 *__restrict__ packed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int pos = atomicAdd(cursor + cell_of_pt[i], 1);
  sorted[pos] = i;
  if (packed) packed[pos] = float4{xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2], __int_as_float(i)};
}

// ascending bitonic sort of `m` (<= 1024) ints held in a wave-private LDS buffer padded to a power of two
__device__ __forceinline__ void bq_sort_lds(int *buf, int m, int lane) {
  int p2 = 64;
  while (p2 < m) p2 <<= 1;
  for (int i = m + lane; i < p2; i += 64) buf[i] = 0x7fffffff;
  __builtin_amdgcn_wave_barrier();
  for (int sz = 2; sz <= p2; sz <<= 1)
    for (int j = sz >> 1; j >= 1; j >>= 1) {
      for (int t = lane; t < (p2 >> 1); t += 64) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool asc = (lo & sz) == 0;
        const int a = buf[lo], c = buf[hi];
        if ((a > c) == asc) { buf[lo] = c; buf[hi] = a; }
      }
      __builtin_amdgcn_wave_barrier();
    }
}

template <bool FILL>
__global__ __launch_bounds__(256) void ballquery_grid_kernel(int n, long thre, float radius2, int cap,
                                                             const float *__restrict__ xyz,
                                                             const int32_t *__restrict__ batch_idxs,
                                                             const int32_t *__restrict__ batch_offsets,
                                                             const BqGrid *__restrict__ g, const int32_t *__restrict__ cell_start,
                                                             const int32_t *__restrict__ sorted, int32_t *__restrict__ idx,
                                                             int32_t *__restrict__ start_len, int32_t *__restrict__ counts) {
  __shared__ int hits[4][1024];
  const int lane = lane_id(), wave = wave_id();
  const int p = blockIdx.x * 4 + wave;
  if (p >= n) return;
  const float ox = xyz[p * 3], oy = xyz[p * 3 + 1], oz = xyz[p * 3 + 2];
  const int bi = batch_idxs[p];
  long s0 = 0;
  int limit = cap;
  if (FILL) {
    s0 = start_len[p * 2];
    if (s0 >= thre) return;
    const int c0 = start_len[p * 2 + 1];
    limit = (s0 + c0 >= thre) ? (int)(thre - s0) : c0;
    if (limit <= 0) return;
  }
  int cx, cy, cz;
  bq_cell_of(g, ox, oy, oz, cx, cy, cz);
  // lanes 0..8: the (dy,dz) rows; each row is the contiguous range of cells x-1..x+1
  int rlo = 0, rhi = 0;
  if (lane < 9) {
    const int yy = cy + lane % 3 - 1, zz = cz + lane / 3 - 1;
    if (yy >= 0 && yy < g->dy && zz >= 0 && zz < g->dz) {
      const int rowc = ((bi * g->dz + zz) * g->dy + yy) * g->dx;
      rlo = cell_start[rowc + max(cx - 1, 0)];
      rhi = cell_start[rowc + min(cx + 1, g->dx - 1) + 1];
    }
  }
  int tot = rhi - rlo;
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);        // lanes 0..15 hold the 9-row total
  tot = readlane_i(tot, 0);
  int cnt = 0;
  int *hb = hits[wave];
  if (tot <= 1024) {
    for (int rr = 0; rr < 9; ++rr) {
      const int lo = readlane_i(rlo, rr), hi = readlane_i(rhi, rr);
      for (int base = lo; base < hi; base += 64) {
        const int t = base + lane;
        bool hit = false;
        int kk = 0;
        if (t < hi) {
          kk = sorted[t];
          hit = sqdist3s(ox, oy, oz, xyz[kk * 3], xyz[kk * 3 + 1], xyz[kk * 3 + 2]) < radius2;
        }
        const unsigned long long mask = __ballot(hit);
        if (FILL && hit) hb[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = kk;
        cnt += __popcll(mask);
      }
    }
    if (!FILL) { if (lane == 0) counts[p] = cnt < cap ? cnt : cap; return; }
    __builtin_amdgcn_wave_barrier();
    bq_sort_lds(hb, cnt, lane);
    for (int i = lane; i < min(cnt, limit); i += 64) idx[s0 + i] = hb[i];
    return;
  }
  // crowded neighbourhood: the brute-force scan of the batch segment (ascending by construction)
  const int start = batch_offsets[bi], end = batch_offsets[bi + 1];
  for (int base = start; base < end && cnt < limit; base += 64) {
    const int k = base + lane;
    const bool hit = k < end && sqdist3s(ox, oy, oz, xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2]) < radius2;
    const unsigned long long mask = __ballot(hit);
    if (FILL) {
      const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < limit) idx[s0 + slot] = k;
    }
    cnt += __popcll(mask);
  }
  if (!FILL && lane == 0) counts[p] = cnt < cap ? cnt : cap;
}

// ---------------------------------------------------------------- ball query with on-the-fly similarity
// forward_grouping (M4:1218-1226) hands bfs_cluster.cu:18-77 two dense (n,n) matrices
//     adj = exp(-(cdist(f,f) / max cdist)^2 / 2)   with a zero diagonal            (M4:210-233)
// that only ever get read at the pairs inside the search radius.  Here the two conditions `adj_inst > thr_inst &&
// adj_para > thr_para` are evaluated at exactly those pairs from the feature rows and the per-segment diameter
// (seg_diameter_kernel): no (n,n) tensor, no 12 elementwise passes over it, and every (cloud, class) subset of the
// model is one segment of a single launch.
//
// seg_diameter_kernel: max_{i != j in segment} ||f_i - f_j||^2 in the expanded form xx_i + xx_j - 2 f_i.f_j the
// reference's cdist uses, Gram blocks on v_mfma_f32_16x16x4_f32.  A workgroup owns 64 rows (16 per wave) and walks the
// 64-row column tiles from its own diagonal tile to the end of the segment (the matrix is symmetric).  Rows are
// zero-padded to a multiple of 16 columns by the caller; rows past the end of the segment are clamped duplicates and
// masked in the epilogue.  Lane group g = lane/16 supplies columns 4g..4g+3 of each 16-column chunk as the k index
// of four consecutive MFMAs (the same permutation on both operands, so the contraction is complete).
using bq_f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

__global__ void row_sqnorm_kernel(int n, int C, const float *__restrict__ f, float *__restrict__ xx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < C; ++c) s = fmaf(f[(long)i * C + c], f[(long)i * C + c], s);
  xx[i] = s;
}

// tile_prefix[s] = number of 64-row tiles of the segments before s (inactive segments own none)
__global__ void seg_tile_prefix_kernel(int S, const int32_t *__restrict__ seg_offsets, const int32_t *__restrict__ seg_cls,
                                       int32_t *__restrict__ tile_prefix) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int run = 0;
  for (int s = 0; s < S; ++s) {
    tile_prefix[s] = run;
    if (seg_cls[s] >= 0) run += (seg_offsets[s + 1] - seg_offsets[s] + 63) / 64;
  }
  tile_prefix[S] = run;
}

__global__ __launch_bounds__(256) void seg_diameter_kernel(int S, int C, const float *__restrict__ f,
                                                           const float *__restrict__ xx,
                                                           const int32_t *__restrict__ seg_offsets,
                                                           const int32_t *__restrict__ tile_prefix,
                                                           unsigned int *__restrict__ dmax2) {
  const int tile = blockIdx.x;
  if (tile >= tile_prefix[S]) return;
  int lo = 0, hi = S;                       // last s with tile_prefix[s] <= tile
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_prefix[mid] <= tile) lo = mid; else hi = mid;
  }
  const int sg = lo;
  const int beg = seg_offsets[sg], end = seg_offsets[sg + 1];
  const int lane = lane_id(), wave = wave_id();
  const int li = lane & 15, lk = lane >> 4;
  const int i0 = beg + (tile - tile_prefix[sg]) * 64 + wave * 16;       // this wave's 16 rows
  const float *arow = f + (long)min(i0 + li, end - 1) * C + 4 * lk;
  float xi[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xi[r] = xx[min(i0 + 4 * lk + r, end - 1)];
  float best = 0.f;
  // the operands of chunk kc + 16 (or of the next column tile's first chunk) are requested before the sixteen MFMAs of
  // chunk kc: a load -> wait -> MFMA sequence per chunk left the matrix pipe idle for the whole L2 round trip (0.8 ms for
  // 72 segments of ~900 rows at C = 128, 7 % of the f32 MFMA rate)
  const int jfirst = beg + (tile - tile_prefix[sg]) * 64;
  auto rowptr = [&](int j0, int t) { return f + (long)min(j0 + 16 * t + li, end - 1) * C + 4 * lk; };
  float4 a_nx = *reinterpret_cast<const float4 *>(arow), b_nx[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) b_nx[t] = *reinterpret_cast<const float4 *>(rowptr(jfirst, t));
  for (int j0 = jfirst; j0 < end; j0 += 64) {
    bq_f32x4 acc[4];
    const float *brow[4], *bnext[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = {0.f, 0.f, 0.f, 0.f};
      brow[t] = rowptr(j0, t);
      bnext[t] = rowptr(j0 + 64 < end ? j0 + 64 : j0, t);
    }
    for (int kc = 0; kc < C; kc += 16) {
      const float4 a = a_nx;
      float4 b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = b_nx[t];
      const bool last = kc + 16 >= C;
      a_nx = *reinterpret_cast<const float4 *>(arow + (last ? 0 : kc + 16));
#pragma unroll
      for (int t = 0; t < 4; ++t) b_nx[t] = *reinterpret_cast<const float4 *>(last ? bnext[t] : brow[t] + kc + 16);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t].w, acc[t], 0, 0, 0);
      }
    }
    // D[i = 4*lk + r][j = li]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int j = j0 + 16 * t + li;
      const float xj = xx[min(j, end - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + 4 * lk + r;
        const float d2 = (xi[r] + xj) - 2.f * acc[t][r];
        if (i < end && j < end && i != j) best = fmaxf(best, d2);
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
  if (lane == 0 && best > 0.f) atomicMax(dmax2 + sg, __float_as_uint(best));   // non-negative floats order as uints
}

// Same-address atomics retire one every ~10 ns on the L2: 65 536 list reservations on ONE counter cost 0.7 ms.  idx is
// cut into BQ_REGIONS equal regions with a counter each (128 bytes apart), workgroups take them round robin.
constexpr int BQ_REGIONS = 64;

struct SimArgs {
  const float *fi, *fp;           // (n, Ci), (n, Cp) feature rows, point order
  const float *dmi2, *dmp2;       // (S) squared diameters
  const int32_t *seg_cls;         // (S) class of the segment, < 0 = inactive
  float thr_i, thr_p;
  int Ci, Cp;
  int exact_only;                 // 1: evaluate every similarity from the rows (GCANET_BQ_EXACT=1: tools/debug/bq_fuzz.py)
};

// adjacency value of M4:210-233 from a squared feature distance: exp(-(d/dmax)^2 / 2), zero on the diagonal, NaN when
// dmax == 0 (the reference's 0/0)
__device__ __forceinline__ float sim_from_dist2(float d2, float dmax, bool self) {
  if (dmax == 0.f) return __builtin_nanf("");
  if (self) return 0.f;
  const float a = sqrtf(d2) / dmax;
  return expf(-(a * a) / 2.f);
}

// ||f_p - f_k||^2 by the 16 lanes of a group: lane `sub` owns the float4 columns sub, sub+16, ... (rows are 64-byte
// multiples, so four rows make one fully coalesced 64-lane load), butterfly sum inside the group.  Every lane of the
// group ends with the same value, and the value is symmetric in (p, k) bit for bit -- the neighbour lists must be.
__device__ __forceinline__ float row_dist2_16(const float4 *__restrict__ prow, const float *__restrict__ f, int C4, int k, int sub) {
  const float4 *b = reinterpret_cast<const float4 *>(f) + (long)k * C4;
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = sub + 16 * u;
    if (c < C4) {
      const float4 x = prow[u], y = b[c];
      float d = x.x - y.x; s = fmaf(d, d, s);
      d = x.y - y.y; s = fmaf(d, d, s);
      d = x.z - y.z; s = fmaf(d, d, s);
      d = x.w - y.w; s = fmaf(d, d, s);
    }
  }
  s += __shfl_xor(s, 8);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 1);
  return s;
}

// One pass: a wave collects the points inside the radius of its point (27 cells; the whole segment if more than ~1900
// lie inside) in LDS, four 16-lane groups then evaluate the two similarities of four candidates at a time and
// compact the list in place; it is ordered ascending, `len` slots of idx are reserved with ONE atomic and written.  The
// lists of different points therefore lie in idx in completion order -- start_len carries the start, as in the
// reference's CSR -- and no count pass / exclusive scan / host round trip is needed.  status: [0] sufficient capacity,
// [1] a list hit the cap, [2] idx too small (lists missing; the caller retries with a larger buffer).
__global__ __launch_bounds__(256) void ballquery_sim_kernel(int n, float radius2, int cap, const float *__restrict__ xyz,
                                                            const int32_t *__restrict__ seg_of,
                                                            const int32_t *__restrict__ seg_offsets, SimArgs sa,
                                                            const BqGrid *__restrict__ g,
                                                            const int32_t *__restrict__ cell_start,
                                                            const float4 *__restrict__ packed, int32_t *__restrict__ idx,
                                                            int capacity, int32_t *__restrict__ start_len,
                                                            int32_t *__restrict__ region, int32_t *__restrict__ status) {
  __shared__ int hits[4][2048];
  const int lane = lane_id(), wave = wave_id();
  const int p = blockIdx.x * 4 + wave;
  if (p >= n) return;
  const int sg = seg_of[p];
  if (sa.seg_cls[sg] < 0) {
    if (lane < 2) start_len[p * 2 + lane] = 0;
    return;
  }
  const float ox = xyz[p * 3], oy = xyz[p * 3 + 1], oz = xyz[p * 3 + 2];
  const float dmi = sqrtf(sa.dmi2[sg]), dmp = sqrtf(sa.dmp2[sg]);
  const int sub = lane & 15, grp = lane >> 4;
  const int Ci4 = sa.Ci / 4, Cp4 = sa.Cp / 4;
  float4 pi[4], pp[4];                                   // this point's rows, the columns lane `sub` owns
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = sub + 16 * u;
    pi[u] = c < Ci4 ? reinterpret_cast<const float4 *>(sa.fi)[(long)p * Ci4 + c] : float4{0.f, 0.f, 0.f, 0.f};
    pp[u] = c < Cp4 ? reinterpret_cast<const float4 *>(sa.fp)[(long)p * Cp4 + c] : float4{0.f, 0.f, 0.f, 0.f};
  }
  // A threshold <= 0 needs no distance: d <= dmax inside a segment, so exp(-(d/dmax)^2 / 2) >= 0.6 > thr for every other
  // point (0 on the diagonal, NaN -- never accepted -- when dmax == 0), and the rows of that feature set are not
  // gathered at all (the reference's similarity_threshold_para is 0.0, M4:1140).  Guard: the diameter comes from the
  // expanded form, whose noise floor is ~1e-6 |f|^2 -- only where it stands clear of that (dmax^2 >= 1e-4 |f_p|^2) is
  // d^2 < 208 dmax^2 (no underflow of the exponential to 0) certain; other points take the exact evaluation.
  auto row_sq = [&](const float4 (&r)[4]) -> float {
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) q = fmaf(r[u].x, r[u].x, fmaf(r[u].y, r[u].y, fmaf(r[u].z, r[u].z, fmaf(r[u].w, r[u].w, q))));
    q += __shfl_xor(q, 8); q += __shfl_xor(q, 4); q += __shfl_xor(q, 2); q += __shfl_xor(q, 1);
    return q;
  };
  const float dmi2v = sa.dmi2[sg], dmp2v = sa.dmp2[sg];
  const bool fast_i = !sa.exact_only && sa.thr_i <= 0.f && dmi2v > 0.f && dmi2v < __builtin_inff() && dmi2v >= 1e-4f * row_sq(pi);
  const bool fast_p = !sa.exact_only && sa.thr_p <= 0.f && dmp2v > 0.f && dmp2v < __builtin_inff() && dmp2v >= 1e-4f * row_sq(pp);
  auto in_radius = [&](int kk) -> bool {
    return sqdist3s(ox, oy, oz, xyz[kk * 3], xyz[kk * 3 + 1], xyz[kk * 3 + 2]) < radius2;
  };
  // keeps the ids[0..m) that pass both similarity thresholds, in place and in order; returns the new count.  Eight
  // candidates per step (two per 16-lane group) so that two row loads are in flight per lane.
  auto sim_filter = [&](int *ids, int m) -> int {
    int kept = 0;
    for (int i0 = 0; i0 < m; i0 += 8) {
      const int ia = i0 + grp, ib = i0 + 4 + grp;
      const int ka = ia < m ? ids[ia] : p, kb = ib < m ? ids[ib] : p;
      float sia, sib, spa, spb;                                          // the two similarities of the two candidates
      if (fast_i) {                                                      // wave-uniform
        sia = ka == p ? 0.f : 1.f; sib = kb == p ? 0.f : 1.f;            // any value in (thr, 1] decides the same
      } else {
        const float dia = row_dist2_16(pi, sa.fi, Ci4, ka, sub), dib = row_dist2_16(pi, sa.fi, Ci4, kb, sub);
        sia = sim_from_dist2(dia, dmi, ka == p); sib = sim_from_dist2(dib, dmi, kb == p);
      }
      if (fast_p) {
        spa = ka == p ? 0.f : 1.f; spb = kb == p ? 0.f : 1.f;
      } else {
        const float dpa = row_dist2_16(pp, sa.fp, Cp4, ka, sub), dpb = row_dist2_16(pp, sa.fp, Cp4, kb, sub);
        spa = sim_from_dist2(dpa, dmp, ka == p); spb = sim_from_dist2(dpb, dmp, kb == p);
      }
      const bool oka = ia < m && sub == 0 && sia > sa.thr_i && spa > sa.thr_p;
      const bool okb = ib < m && sub == 0 && sib > sa.thr_i && spb > sa.thr_p;
      const unsigned long long ma = __ballot(oka), mb = __ballot(okb);
      const unsigned long long lt = (1ull << lane) - 1ull;
      if (oka) ids[kept + __popcll(ma & lt)] = ka;                     // slots never run ahead of the reads
      kept += __popcll(ma);
      if (okb) ids[kept + __popcll(mb & lt)] = kb;
      kept += __popcll(mb);
    }
    return kept;
  };
  // reserve `len` slots in this workgroup's region of idx; returns the start (or -1 and len = 0 when it is full)
  auto reserve = [&](int &len) -> int {
    const int rg = blockIdx.x % BQ_REGIONS, rcap = capacity / BQ_REGIONS;
    int s0 = 0;
    if (lane == 0) {
      s0 = len > 0 ? atomicAdd(region + rg * 32, len) : 0;
      if (len > 0 && s0 + len > rcap) { atomicOr(status + 2, 1); s0 = -1; }
      else s0 += rg * rcap;
    }
    s0 = readlane_i(s0, 0);
    if (s0 < 0) len = 0;
    if (lane == 0) { start_len[p * 2] = s0 < 0 ? 0 : s0; start_len[p * 2 + 1] = len; }
    return s0;
  };
  int cx, cy, cz;
  bq_cell_of(g, ox, oy, oz, cx, cy, cz);
  int rlo = 0, rhi = 0;
  if (lane < 9) {
    const int yy = cy + lane % 3 - 1, zz = cz + lane / 3 - 1;
    if (yy >= 0 && yy < g->dy && zz >= 0 && zz < g->dz) {
      const int rowc = ((sg * g->dz + zz) * g->dy + yy) * g->dx;
      rlo = cell_start[rowc + max(cx - 1, 0)];
      rhi = cell_start[rowc + min(cx + 1, g->dx - 1) + 1];
    }
  }
  int tot = rhi - rlo;
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
  tot = readlane_i(tot, 0);
  int cnt = 0;
  int *hb = hits[wave];
  bool crowded = false;
  {
    // the nine row ranges as ONE candidate stream: stream position t -> range r with pre[r] <= t < pre[r+1]
    int pre = rhi - rlo;                                   // inclusive prefix over lanes 0..8
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) { const int y = __shfl_up(pre, d); if (lane >= d) pre += y; }
    int rpre[9], rbase[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) { rpre[r] = readlane_i(pre, r); rbase[r] = readlane_i(rlo, r) - (r ? readlane_i(pre, r - 1) : 0); }
    for (int base = 0; base < tot; base += 128) {
      if (cnt > 2048 - 128) { crowded = true; break; }          // more points INSIDE the radius than the buffer holds
      const int ta = base + lane, tb = base + 64 + lane;
      int offa = rbase[0], offb = rbase[0];
#pragma unroll
      for (int r = 1; r < 9; ++r) { offa = ta >= rpre[r - 1] ? rbase[r] : offa; offb = tb >= rpre[r - 1] ? rbase[r] : offb; }
      const float4 qa = packed[ta < tot ? ta + offa : 0], qb = packed[tb < tot ? tb + offb : 0];
      const bool ha = ta < tot && sqdist3s(ox, oy, oz, qa.x, qa.y, qa.z) < radius2;
      const bool hb2 = tb < tot && sqdist3s(ox, oy, oz, qb.x, qb.y, qb.z) < radius2;
      const unsigned long long ma = __ballot(ha), mb = __ballot(hb2);
      const unsigned long long lt = (1ull << lane) - 1ull;
      if (ha) hb[cnt + __popcll(ma & lt)] = __float_as_int(qa.w);
      cnt += __popcll(ma);
      if (hb2) hb[cnt + __popcll(mb & lt)] = __float_as_int(qb.w);
      cnt += __popcll(mb);
    }
  }
  if (!crowded) {
    __builtin_amdgcn_wave_barrier();
    cnt = sim_filter(hb, cnt);
    if (cnt >= cap && lane == 0) atomicOr(status + 1, 1);
    int len = min(cnt, cap);
    const int s0 = reserve(len);
    if (len == 0) return;
    __builtin_amdgcn_wave_barrier();
    bq_sort_lds(hb, cnt, lane);
    for (int i = lane; i < len; i += 64) idx[s0 + i] = hb[i];
    return;
  }
  // more than ~1900 points inside the radius: scan the whole segment twice (count, then write; ascending by construction)
  const int start = seg_offsets[sg], end = seg_offsets[sg + 1];
  for (int pass = 0, len = 0, s0 = 0; pass < 2; ++pass) {
    cnt = 0;
    const int limit = pass == 0 ? cap : len;
    for (int base = start; base < end && cnt < limit; base += 64) {
      const int k = base + lane;
      const bool hit = k < end && in_radius(k);
      const unsigned long long mask = __ballot(hit);
      if (hit) hb[__popcll(mask & ((1ull << lane) - 1ull))] = k;
      __builtin_amdgcn_wave_barrier();
      const int m = sim_filter(hb, __popcll(mask));
      __builtin_amdgcn_wave_barrier();
      if (pass == 1 && lane < m && cnt + lane < len) idx[s0 + cnt + lane] = hb[lane];
      cnt += m;
    }
    if (pass == 0) {
      if (cnt >= cap && lane == 0) atomicOr(status + 1, 1);
      len = min(cnt, cap);
      s0 = reserve(len);
      if (len == 0) return;
    }
  }
}

// status[0] = a capacity that holds every list: BQ_REGIONS x the fullest region (+ one cap-sized list of slack each)
__global__ void bq_status_kernel(const int32_t *__restrict__ region, int32_t *__restrict__ status) {
  int m = 0;
  for (int r = 0; r < BQ_REGIONS; ++r) m = max(m, region[r * 32]);
  status[0] = (int)min((long)BQ_REGIONS * (m + 3000L), 2147483647L);
}

// ---------------------------------------------------------------- segment ops
// sec_mean.cu:13-85, roipool.cu:12-32.  OP 0 sec_mean (sum of v/count), 1 min, 2 max,
// 3 global_avg_pool (sum then divide).  Row order is kept sequential per plane -> bit-exact.
template <int OP>
__global__ __launch_bounds__(256) void segment_kernel(int P, int C, const float *__restrict__ inp,
                                                      const int32_t *__restrict__ offsets,
                                                      float *__restrict__ out) {
  const int lane = lane_id();
  const int chunks = (C + 63) / 64;
  const int w = blockIdx.x * 4 + wave_id();
  if (w >= P * chunks) return;
  const int pid = w / chunks, p = (w % chunks) * 64 + lane;
  const int start = offsets[pid], end = offsets[pid + 1];
  if (p >= C) return;
  const float count = (float)(end - start);
  float acc = OP == 1 ? __builtin_inff() : (OP == 2 ? -__builtin_inff() : 0.f);
  for (int i = start; i < end; ++i) {
    const float v = inp[(long)i * C + p];
    if (OP == 0) acc += v / count;
    else if (OP == 1) acc = v < acc ? v : acc;
    else if (OP == 2) acc = v > acc ? v : acc;
    else acc += v;
  }
  if (OP == 3) acc = acc / count;
  out[(long)pid * C + p] = acc;
}

// roipool.cu:46-60: every row of a segment receives d_out / n_points (plain stores: a row
// belongs to one segment)
__global__ __launch_bounds__(256) void avg_pool_bp_kernel(int P, int C, float *__restrict__ d_feats,
                                                          const int32_t *__restrict__ offsets,
                                                          const float *__restrict__ d_out) {
  const int pid = blockIdx.y;
  const int start = offsets[pid], end = offsets[pid + 1];
  const float n_points = (float)(end - start);
  const long total = (long)(end - start) * C;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int p = (int)(e % C);
    d_feats[(long)start * C + e] = d_out[(long)pid * C + p] / n_points;
  }
}

// ---------------------------------------------------------------- IoU / mask labels
// cal_iou_and_masklabel.cu:9-68: intersection counts via an LDS histogram of the labels of
// the proposal's points (integer counts -> exact).
__global__ __launch_bounds__(256) void mask_iou_kernel(int nInstance, const int32_t *__restrict__ proposals_idx,
                                                       const int32_t *__restrict__ proposals_offset,
                                                       const int64_t *__restrict__ instance_labels,
                                                       const int32_t *__restrict__ instance_pointnum,
                                                       const float *__restrict__ mask, float *__restrict__ iou) {
  extern __shared__ int hist[];  // nInstance + 1
  const int pid = blockIdx.x;
  const int start = proposals_offset[pid], end = proposals_offset[pid + 1];
  for (int i = threadIdx.x; i <= nInstance; i += 256) hist[i] = 0;
  __syncthreads();
  for (int i = start + threadIdx.x; i < end; i += 256) {
    if (mask && !(mask[i] > 0.5f)) continue;
    atomicAdd(&hist[nInstance], 1);  // proposal_total
    const int lab = (int)instance_labels[proposals_idx[i]];
    if (lab >= 0 && lab < nInstance) atomicAdd(&hist[lab], 1);
  }
  __syncthreads();
  const int proposal_total = hist[nInstance];
  for (int inst = threadIdx.x; inst < nInstance; inst += 256) {
    const int inter = hist[inst];
    const double den = (double)(float)(proposal_total + instance_pointnum[inst] - inter) + 1e-5;
    iou[(long)pid * nInstance + inst] = (float)((double)(float)inter / den);
  }
}

// cal_iou_and_masklabel.cu:70-104
__global__ __launch_bounds__(256) void mask_label_kernel(int nInstance, float iou_thr,
                                                         const int32_t *__restrict__ proposals_idx,
                                                         const int32_t *__restrict__ proposals_offset,
                                                         const int64_t *__restrict__ instance_labels,
                                                         const int64_t *__restrict__ instance_cls,
                                                         const float *__restrict__ proposals_iou,
                                                         float *__restrict__ mask_label) {
  __shared__ float s_iou[256];
  __shared__ int s_ind[256];
  const int pid = blockIdx.x, tid = threadIdx.x;
  // first maximum with strict '>' starting from 0 == max value, lowest index among equals
  float best = 0.f;
  int bind = 0x7fffffff;
  for (int inst = tid; inst < nInstance; inst += 256) {
    const float v = proposals_iou[(long)pid * nInstance + inst];
    if (v > best && instance_cls[inst] != -100) { best = v; bind = inst; }
  }
  s_iou[tid] = best; s_ind[tid] = bind;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if (tid < off) {
      const float v = s_iou[tid + off];
      const int i2 = s_ind[tid + off];
      if (v > s_iou[tid] || (v == s_iou[tid] && i2 < s_ind[tid])) { s_iou[tid] = v; s_ind[tid] = i2; }
    }
    __syncthreads();
  }
  const float max_iou = s_iou[0];
  const int max_ind = s_ind[0] == 0x7fffffff ? 0 : s_ind[0];
  if (!(max_iou >= iou_thr)) return;
  const int start = proposals_offset[pid], end = proposals_offset[pid + 1];
  for (int i = start + tid; i < end; i += 256)
    mask_label[i] = ((int)instance_labels[proposals_idx[i]] == max_ind) ? 1.f : 0.f;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_voxelize_fp(int M, int maxActive, int C, const float *feats, float *output_feats,
                               const int32_t *rules, int average, void *stream) {
  GCN_REQUIRE(M >= 0 && maxActive >= 0 && C >= 0, "gcn_voxelize_fp: bad shape");
  if (M == 0 || C == 0) return GCN_OK;
  GCN_REQUIRE(feats && output_feats && rules, "gcn_voxelize_fp: null pointer");
  voxelize_fp_kernel<<<cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(M, maxActive, C, feats, output_feats, rules, average);
  return check_launch("voxelize_fp_kernel");
}

GCN_EXPORT int gcn_voxelize_bp(int M, int maxActive, int C, const float *d_output_feats, float *d_feats,
                               const int32_t *rules, int average, void *stream) {
  GCN_REQUIRE(M >= 0 && maxActive >= 0 && C >= 0, "gcn_voxelize_bp: bad shape");
  if (M == 0 || C == 0) return GCN_OK;
  GCN_REQUIRE(d_output_feats && d_feats && rules, "gcn_voxelize_bp: null pointer");
  voxelize_bp_kernel<<<cdiv(M, 4), 256, 0, (hipStream_t)stream>>>(M, maxActive, C, d_output_feats, d_feats, rules, average);
  return check_launch("voxelize_bp_kernel");
}

// uniform-grid workspace: [64-byte header: BqGrid (56 B) + flags | packed (n float4) | cell_start (mc+1) | cursor (mc+1) | cell_of_pt (n) |
// sorted (n) | scan block sums]
struct GridWs {
  BqGrid *g;
  int32_t *flags, *cell_start, *cursor, *cell_of_pt, *sorted, *bsum, *region;
  float4 *packed;
  int max_cells;
};
static long grid_ws_bytes(long n, long max_cells) {
  return 64 + 16 * n + 4 * (2 * (max_cells + 1) + 2 * n + scan_blocks(max_cells + 1) + BQ_REGIONS * 32) + 64;
}
static GridWs grid_ws_carve(void *ws, int n, int max_cells) {
  GridWs w;
  w.g = (BqGrid *)ws;
  w.flags = (int32_t *)((char *)ws + 56);
  w.packed = (float4 *)((char *)ws + 64);
  w.cell_start = (int32_t *)((char *)ws + 64 + 16L * n);
  w.cursor = w.cell_start + max_cells + 1;
  w.cell_of_pt = w.cursor + max_cells + 1;
  w.sorted = w.cell_of_pt + n;
  w.bsum = w.sorted + n;
  w.region = w.bsum + scan_blocks(max_cells + 1);
  w.max_cells = max_cells;
  return w;
}
// counting sort of the points by (segment, z, y, x) cell; everything on the device
static int grid_build(const GridWs &w, int n, const float *xyz, const int32_t *seg_of, int nseg, float radius, hipStream_t st) {
  GCN_HIP(fill_dev(w.g, 0x00, 64, st));
  GCN_HIP(fill_dev(&w.g->bmin[0], 0xff, 12, st));
  GCN_HIP(fill_dev(w.cell_start, 0, sizeof(int32_t) * (size_t)(w.max_cells + 1), st));
  bq_bbox_kernel<<<32, 1024, 0, st>>>(n, xyz, w.g);
  bq_setup_kernel<<<1, 1, 0, st>>>(w.g, radius, nseg, w.max_cells);
  bq_cell_count_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, xyz, seg_of, w.g, w.cell_of_pt, w.cell_start);
  exscan_rows(st, 1, w.max_cells + 1, w.cell_start, w.bsum, w.cursor);     // unused cells are empty: start[ncell..] = n
  bq_cell_fill_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, w.cell_of_pt, w.cursor, w.sorted, xyz, w.packed);
  return GCN_OK;
}

GCN_EXPORT long gcn_ballquery_grid_ws_bytes(int n) {
  if (n < 0) return -1;
  return grid_ws_bytes(n, 4L * n + 4096);
}

GCN_EXPORT int gcn_ballquery_batch_p(int n, int meanActive, float radius, const float *xyz,
                                     const int32_t *batch_idxs, const int32_t *batch_offsets,
                                     const float *adj_inst, float thr_inst, const float *adj_para,
                                     float thr_para, int32_t *idx, int32_t *start_len, int32_t *count_ws,
                                     int nbatch, void *grid_ws, int *total_host, void *stream) {
  GCN_REQUIRE(n >= 0 && meanActive >= 0, "gcn_ballquery_batch_p: bad shape");
  GCN_REQUIRE(total_host, "gcn_ballquery_batch_p: total_host is null");
  GCN_REQUIRE((adj_inst == nullptr) == (adj_para == nullptr), "gcn_ballquery_batch_p: adj_inst and adj_para must both be given or both be NULL");
  *total_host = 0;
  if (n == 0) return GCN_OK;
  GCN_REQUIRE(xyz && batch_idxs && batch_offsets && start_len && count_ws, "gcn_ballquery_batch_p: null pointer");
  GCN_REQUIRE(idx || meanActive == 0, "gcn_ballquery_batch_p: idx is null");
  hipStream_t st = (hipStream_t)stream;
  const int cap = adj_inst ? 3000 : 1000;  // bfs_cluster.cu:54 / bfs_cluster_easy.cu:43
  const float r2 = radius * radius;
  const long thre = (long)n * meanActive;
  if (!adj_inst && grid_ws && n >= 2048 && nbatch >= 1 && radius > 0.f) {
    const GridWs w = grid_ws_carve(grid_ws, n, 4 * n + 4096);
    int grc = grid_build(w, n, xyz, batch_idxs, nbatch, radius, st);
    if (grc) return grc;
    const BqGrid *g = w.g;
    const int32_t *cell_start = w.cell_start, *sorted = w.sorted;
    ballquery_grid_kernel<false><<<cdiv(n, 4), 256, 0, st>>>(n, thre, r2, cap, xyz, batch_idxs, batch_offsets, g, cell_start,
                                                              sorted, idx, start_len, count_ws);
    scan_counts_kernel<<<1, 1024, 0, st>>>(n, count_ws, start_len);
    if (thre > 0)
      ballquery_grid_kernel<true><<<cdiv(n, 4), 256, 0, st>>>(n, thre, r2, cap, xyz, batch_idxs, batch_offsets, g, cell_start,
                                                               sorted, idx, start_len, count_ws);
  } else {
    ballquery_kernel<false><<<cdiv(n, 4), 256, 0, st>>>(n, thre, r2, cap, xyz, batch_idxs, batch_offsets, adj_inst, thr_inst,
                                                         adj_para, thr_para, idx, start_len, count_ws);
    scan_counts_kernel<<<1, 1024, 0, st>>>(n, count_ws, start_len);
    if (thre > 0)
      ballquery_kernel<true><<<cdiv(n, 4), 256, 0, st>>>(n, thre, r2, cap, xyz, batch_idxs, batch_offsets, adj_inst, thr_inst,
                                                          adj_para, thr_para, idx, start_len, count_ws);
  }
  int rc = check_launch("ballquery_kernel");
  if (rc) return rc;
  GCN_HIP(hipMemcpyAsync(total_host, count_ws + n, sizeof(int), hipMemcpyDeviceToHost, st));
  GCN_HIP(hipStreamSynchronize(st));
  return GCN_OK;
}

// for csrc/segdiam.hip: the exhaustive kernel over the tiles tile_prefix assigns (segments it leaves out own none)
void launch_seg_diameter_tiles(int n, int S, int C, const float *feats, const float *xx, const int32_t *seg_offsets,
                               const int32_t *tile_prefix, float *dmax2, hipStream_t st) {
  seg_diameter_kernel<<<n / 64 + S, 256, 0, st>>>(S, C, feats, xx, seg_offsets, tile_prefix, (unsigned int *)dmax2);
}

GCN_EXPORT int gcn_segment_diameter2(int n, int C, const float *feats, const int32_t *seg_offsets,
                                     const int32_t *seg_cls, int S, float *xx_ws, int32_t *tile_ws, float *dmax2,
                                     void *stream) {
  GCN_REQUIRE(n >= 0 && S >= 0 && C > 0 && C % 16 == 0, "gcn_segment_diameter2: C must be a positive multiple of 16 (zero-pad the rows), got C=%d", C);
  if (n == 0 || S == 0) return GCN_OK;
  GCN_REQUIRE(feats && seg_offsets && seg_cls && xx_ws && tile_ws && dmax2, "gcn_segment_diameter2: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(dmax2, 0, sizeof(float) * (size_t)S, st));
  row_sqnorm_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, C, feats, xx_ws);
  seg_tile_prefix_kernel<<<1, 1, 0, st>>>(S, seg_offsets, seg_cls, tile_ws);
  seg_diameter_kernel<<<n / 64 + S, 256, 0, st>>>(S, C, feats, xx_ws, seg_offsets, tile_ws, (unsigned int *)dmax2);
  return check_launch("seg_diameter_kernel");
}

GCN_EXPORT long gcn_ballquery_sim_ws_bytes(int n) {
  if (n < 0) return -1;
  return grid_ws_bytes(n, 128L * n + 4096);
}

GCN_EXPORT int gcn_ballquery_sim(int n, float radius, const float *xyz, const int32_t *seg_of,
                                 const int32_t *seg_offsets, const int32_t *seg_cls, int S, const float *feat_inst,
                                 int Ci, const float *dmax2_inst, float thr_inst, const float *feat_para, int Cp,
                                 const float *dmax2_para, float thr_para, int32_t *idx, int capacity,
                                 int32_t *start_len, int32_t *status, void *grid_ws, void *stream) {
  GCN_REQUIRE(n >= 0 && n <= (1 << 23) && S >= 1 && capacity >= 0, "gcn_ballquery_sim: bad shape");
  GCN_REQUIRE(Ci > 0 && Cp > 0 && Ci % 16 == 0 && Cp % 16 == 0 && Ci <= 256 && Cp <= 256, "gcn_ballquery_sim: feature rows must be zero-padded to a multiple of 16 columns, at most 256 (Ci=%d, Cp=%d)", Ci, Cp);
  GCN_REQUIRE(status, "gcn_ballquery_sim: status is null");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(status, 0, 4 * sizeof(int32_t), st));
  if (n == 0) return GCN_OK;
  GCN_REQUIRE(xyz && seg_of && seg_offsets && seg_cls && feat_inst && dmax2_inst && feat_para && dmax2_para && start_len &&
              grid_ws && (idx || capacity == 0), "gcn_ballquery_sim: null pointer");
  GCN_REQUIRE(radius > 0.f, "gcn_ballquery_sim: radius must be positive");
  const GridWs w = grid_ws_carve(grid_ws, n, 128 * n + 4096);     // segments x cells of edge ~radius: HBM is plentiful
  int rc = grid_build(w, n, xyz, seg_of, S, radius, st);
  if (rc) return rc;
  const char *bq_exact = getenv("GCANET_BQ_EXACT");
  SimArgs sa{feat_inst, feat_para, dmax2_inst, dmax2_para, seg_cls, thr_inst, thr_para, Ci, Cp, (bq_exact && bq_exact[0] == '1') ? 1 : 0};
  GCN_HIP(fill_dev(w.region, 0, sizeof(int32_t) * BQ_REGIONS * 32, st));
  ballquery_sim_kernel<<<cdiv(n, 4), 256, 0, st>>>(n, radius * radius, 3000, xyz, seg_of, seg_offsets, sa, w.g, w.cell_start,
                                                   w.packed, idx, capacity, start_len, w.region, status);
  bq_status_kernel<<<1, 1, 0, st>>>(w.region, status);
  return check_launch("ballquery_sim_kernel");
}

GCN_EXPORT int gcn_sec_op(int op, int P, int C, const float *inp, const int32_t *offsets, float *out, void *stream) {
  GCN_REQUIRE(op >= 0 && op <= 2, "gcn_sec_op: op must be 0 (mean), 1 (min) or 2 (max)");
  GCN_REQUIRE(P >= 0 && C >= 0, "gcn_sec_op: bad shape");
  if (P == 0 || C == 0) return GCN_OK;
  GCN_REQUIRE(inp && offsets && out, "gcn_sec_op: null pointer");
  const int waves = P * ((C + 63) / 64);
  hipStream_t st = (hipStream_t)stream;
  if (op == 0) segment_kernel<0><<<cdiv(waves, 4), 256, 0, st>>>(P, C, inp, offsets, out);
  else if (op == 1) segment_kernel<1><<<cdiv(waves, 4), 256, 0, st>>>(P, C, inp, offsets, out);
  else segment_kernel<2><<<cdiv(waves, 4), 256, 0, st>>>(P, C, inp, offsets, out);
  return check_launch("segment_kernel");
}

GCN_EXPORT int gcn_global_avg_pool_fp(int P, int C, const float *feats, const int32_t *offsets, float *out, void *stream) {
  GCN_REQUIRE(P >= 0 && C >= 0, "gcn_global_avg_pool_fp: bad shape");
  if (P == 0 || C == 0) return GCN_OK;
  GCN_REQUIRE(feats && offsets && out, "gcn_global_avg_pool_fp: null pointer");
  const int waves = P * ((C + 63) / 64);
  segment_kernel<3><<<cdiv(waves, 4), 256, 0, (hipStream_t)stream>>>(P, C, feats, offsets, out);
  return check_launch("segment_kernel<avg>");
}

GCN_EXPORT int gcn_global_avg_pool_bp(int P, int C, float *d_feats, const int32_t *offsets, const float *d_out, void *stream) {
  GCN_REQUIRE(P >= 0 && C >= 0, "gcn_global_avg_pool_bp: bad shape");
  if (P == 0 || C == 0) return GCN_OK;
  GCN_REQUIRE(d_feats && offsets && d_out, "gcn_global_avg_pool_bp: null pointer");
  avg_pool_bp_kernel<<<dim3(64, P), 256, 0, (hipStream_t)stream>>>(P, C, d_feats, offsets, d_out);
  return check_launch("avg_pool_bp_kernel");
}

GCN_EXPORT int gcn_get_mask_iou(int nInstance, int nProposal, const int32_t *proposals_idx,
                                const int32_t *proposals_offset, const int64_t *instance_labels,
                                const int32_t *instance_pointnum, const float *mask_scores_sigmoid,
                                float *proposals_iou, void *stream) {
  GCN_REQUIRE(nInstance >= 0 && nProposal >= 0, "gcn_get_mask_iou: bad shape");
  GCN_REQUIRE(nInstance <= 16000, "gcn_get_mask_iou: nInstance=%d > 16000 unsupported", nInstance);
  if (nInstance == 0 || nProposal == 0) return GCN_OK;
  GCN_REQUIRE(proposals_idx && proposals_offset && instance_labels && instance_pointnum && proposals_iou, "gcn_get_mask_iou: null pointer");
  mask_iou_kernel<<<nProposal, 256, (nInstance + 1) * sizeof(int), (hipStream_t)stream>>>(
      nInstance, proposals_idx, proposals_offset, instance_labels, instance_pointnum, mask_scores_sigmoid, proposals_iou);
  return check_launch("mask_iou_kernel");
}

GCN_EXPORT int gcn_get_mask_label(int nInstance, int nProposal, float iou_thr, const int32_t *proposals_idx,
                                  const int32_t *proposals_offset, const int64_t *instance_labels,
                                  const int64_t *instance_cls, const float *proposals_iou, float *mask_label,
                                  void *stream) {
  GCN_REQUIRE(nInstance >= 0 && nProposal >= 0, "gcn_get_mask_label: bad shape");
  if (nProposal == 0) return GCN_OK;
  GCN_REQUIRE(proposals_idx && proposals_offset && instance_labels && instance_cls && proposals_iou && mask_label, "gcn_get_mask_label: null pointer");
  mask_label_kernel<<<nProposal, 256, 0, (hipStream_t)stream>>>(nInstance, iou_thr, proposals_idx, proposals_offset,
                                                                instance_labels, instance_cls, proposals_iou, mask_label);
  return check_launch("mask_label_kernel");
}
