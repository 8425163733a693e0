// cluster_dev.hip -- hierarchical_aggregation (softgroup/ops/src/hierarchical_aggregation/hierarchical_aggregation.cpp:
// 20-131, a host BFS over CPU tensors in the reference) on DEVICE buffers, for every (cloud, class) segment of
// forward_grouping (M4:1123-1295) in one pass.  SURVEY.md section 8(f) rank 1.
//
// The reference walks the points in index order, starts a breadth-first search at every unvisited point and emits
// the members in dequeue order; clusters smaller than 0.05*mean are dropped, those below 0.3*mean ("kept" fragments)
// are listed before the larger ("primary") ones.  With symmetric neighbour lists (the ball query is symmetric unless a
// list hit the 3000-entry cap -- the caller checks that flag and takes the host routine then) this is:
//   components   lock-free union-find, larger root hooked under the smaller -> the representative of a component is
//                its lowest index = the reference's BFS seed, and seeds in ascending order = its discovery order
//   offsets      four exclusive scans (sizes and counts of kept / primary roots) give every emitted component its
//                slot: segment by segment, kept before primary, ascending seed inside each group
//   BFS order    one workgroup per component replays the reference's queue level by level: the position of a newly
//                reached point in the queue is fixed by its FIRST discoverer (queue rank t, list position pos), so
//                every frontier node proposes key = t<<12|pos with atomicMin, the winners of a node are counted, one
//                block scan turns the counts into queue slots, and the winners are written in list order.  No global
//                synchronisation: a component never leaves its workgroup; components are handed out by an atomic cursor.
#include "common.h"

namespace gcn {

__device__ __forceinline__ int ld_i(const int32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned int ld_u(const unsigned int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i(int32_t *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// parent = the smallest neighbour below the point (lists are ascending, so that is the first entry) -- already a forest
// with parent < child that joins most of every component -- or the point itself
__global__ void cc_init_kernel(int n, const int32_t *__restrict__ nbr, const int32_t *__restrict__ start_len,
                               int32_t *__restrict__ parent, int32_t *__restrict__ csize, unsigned int *__restrict__ key,
                               int32_t *__restrict__ visited, int32_t *__restrict__ counters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int par = i;
    if (start_len[2 * i + 1] > 0) par = min(i, nbr[start_len[2 * i]]);
    parent[i] = par; csize[i] = 0; key[i] = 0xFFFFFFFFu; visited[i] = 0;
  }
  if (i < 8) counters[i] = 0;
}

// Root of x.  FRESH = false reads through the L1: a stale value is an ancestor that was valid earlier, the walk only
// ever moves to smaller indices, and the compare-and-swap in the union is the arbiter (it returns the current parent
// when the presumed root has been hooked meanwhile).
template <bool FRESH>
__device__ __forceinline__ int uf_find(int32_t *parent, int x) {
  for (;;) {
    const int p = FRESH ? ld_i(parent + x) : parent[x];
    if (p == x) return x;
    const int gp = FRESH ? ld_i(parent + p) : parent[p];
    if (gp != p) st_i(parent + x, gp);          // path halving: any ancestor is a valid parent
    x = p;
  }
}

// wave per point, lanes across its neighbour list; the lists are symmetric, so the edges to smaller indices suffice
__global__ __launch_bounds__(256) void cc_union_kernel(int n, const int32_t *__restrict__ nbr, const int32_t *__restrict__ start_len,
                                                       int32_t *parent) {
  const int p = blockIdx.x * 4 + wave_id();
  if (p >= n) return;
  const int s = start_len[2 * p], len = start_len[2 * p + 1];
  for (int pos = lane_id() + 1; pos < len; pos += 64) {          // entry 0 is the initial parent
    const int v = nbr[s + pos];
    if (v >= p) break;                                            // ascending list: nothing smaller follows
    int a = uf_find<false>(parent, p), b = uf_find<false>(parent, v);
    while (a != b) {
      if (a < b) { const int t = a; a = b; b = t; }             // hook the larger root a under the smaller b
      const int old = atomicCAS(parent + a, a, b);
      if (old == a) break;
      a = uf_find<false>(parent, old);
      b = uf_find<false>(parent, b);
    }
  }
}

__global__ void cc_flatten_kernel(int n, int32_t *parent, int32_t *__restrict__ comp, int32_t *__restrict__ csize) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = uf_find<true>(parent, i);
  comp[i] = r;
  atomicAdd(csize + r, 1);
}

// vals (4, n+1): [kept size, primary size, kept count, primary count] of the roots; entry n = 0 (scan total slot)
__global__ void cluster_classify_kernel(int n, const int32_t *__restrict__ comp, const int32_t *__restrict__ csize,
                                        const int32_t *__restrict__ seg_of, const int32_t *__restrict__ seg_cls,
                                        float size_threshold, int32_t *__restrict__ vals) {
  // hierarchical_aggregation.cpp:7-8
  const float class_mean[10] = {-1.f, -1.f, 3917.f, 12056.f, 2303.f, 8331.f, 3948.f, 3166.f, 5629.f, 11719.f};
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int ks = 0, ps = 0;
  if (i < n && comp[i] == i) {
    const int cls = seg_cls[seg_of[i]];
    if (cls >= 0) {
      const int sz = csize[i];
      if (size_threshold >= 0.f) {                       // bfs_cluster.cpp:86-115: one list, clusters of >= threshold points
        if (sz >= size_threshold) ks = sz;
      } else {
        const float mean = class_mean[cls];
        const float low = (float)(0.05 * mean), high = (float)(0.3 * mean);    // hierarchical_aggregation.cpp:60-61
        if (sz < high) { if (sz >= low) ks = sz; }
        else ps = sz;
      }
    }
  }
  const long W = n + 1;
  vals[i] = ks; vals[W + i] = ps; vals[2 * W + i] = ks > 0; vals[3 * W + i] = ps > 0;
}

// every emitted root gets its slot and cluster id; components of two or more points go on the BFS work list
__global__ void cluster_offsets_kernel(int n, const int32_t *__restrict__ comp, const int32_t *__restrict__ csize,
                                       const int32_t *__restrict__ seg_of, const int32_t *__restrict__ seg_offsets,
                                       const int32_t *__restrict__ vals_in, const int32_t *__restrict__ scan,
                                       int32_t *__restrict__ cluster_offsets, int32_t *__restrict__ out,
                                       int32_t *__restrict__ visited, int32_t *__restrict__ work, int32_t *__restrict__ counters) {
  const long W = n + 1;
  const int32_t *EK = scan, *EP = scan + W, *CK = scan + 2 * W, *CP = scan + 3 * W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    const int total = EK[n] + EP[n], ncl = CK[n] + CP[n];
    cluster_offsets[ncl] = total;
    counters[2] = total;
    counters[3] = ncl;
  }
  if (i >= n || comp[i] != i) return;
  const int ks = vals_in[i], ps = vals_in[W + i];
  if (ks == 0 && ps == 0) return;
  const int sg = seg_of[i], a0 = seg_offsets[sg], a1 = seg_offsets[sg + 1];
  int off, id;
  if (ks > 0) { off = EP[a0] + EK[i]; id = CP[a0] + CK[i]; }
  else        { off = EK[a1] + EP[i]; id = CK[a1] + CP[i]; }
  cluster_offsets[id] = off;
  out[off] = i;
  visited[i] = 1;
  const int sz = csize[i];
  if (sz >= 2) {
    const int w = atomicAdd(counters + 0, 1);
    work[3 * w] = i; work[3 * w + 1] = off; work[3 * w + 2] = sz;
  }
}

__global__ __launch_bounds__(1024) void cluster_bfs_kernel(const int32_t *__restrict__ nbr, const int32_t *__restrict__ start_len,
                                                           const int32_t *__restrict__ work, int32_t *counters,
                                                           unsigned int *key, int32_t *visited, int32_t *base, int32_t *out) {
  __shared__ int s_item, s_total, s_wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (;;) {
    __syncthreads();
    if (tid == 0) s_item = atomicAdd(counters + 1, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= ld_i(counters + 0)) return;
    const int o = work[3 * item + 1], m = work[3 * item + 2];
    int lo = 0, hi = 1;                         // the queue is out[o .. o+m); out[o] = seed
    while (lo < hi && hi < m) {
      // propose: first discoverer (queue rank, list position) wins
      for (int t = lo + wave; t < hi; t += 16) {
        const int u = ld_i(out + o + t);
        const int s = start_len[2 * u], len = start_len[2 * u + 1];
        for (int pos = lane; pos < len; pos += 64) {
          const int v = nbr[s + pos];
          if (!ld_i(visited + v)) atomicMin(key + v, ((unsigned int)t << 12) | (unsigned int)pos);
        }
      }
      __syncthreads();
      // winners per frontier node
      for (int t = lo + wave; t < hi; t += 16) {
        const int u = ld_i(out + o + t);
        const int s = start_len[2 * u], len = start_len[2 * u + 1];
        int w = 0;
        for (int b0 = 0; b0 < len; b0 += 64) {
          const int pos = b0 + lane;
          const bool win = pos < len && ld_u(key + nbr[s + pos]) == (((unsigned int)t << 12) | (unsigned int)pos);
          w += __popcll(__ballot(win));
        }
        if (lane == 0) st_i(base + o + t, w);
      }
      __syncthreads();
      // exclusive scan of the winner counts of this level
      const int F = hi - lo, per = (F + 1023) / 1024;
      const int a = min(lo + tid * per, hi), b = min(a + per, hi);
      int sum = 0;
      for (int i = a; i < b; ++i) sum += ld_i(base + o + i);
      int inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(inc, d); if (lane >= d) inc += y; }
      if (lane == 63) s_wtot[wave] = inc;
      __syncthreads();
      int run = inc - sum;
      for (int w = 0; w < wave; ++w) run += s_wtot[w];
      if (tid == 1023) s_total = run + sum;
      for (int i = a; i < b; ++i) { const int c = ld_i(base + o + i); st_i(base + o + i, run); run += c; }
      __syncthreads();
      const int total = s_total;
      // append the winners in (rank, position) order
      for (int t = lo + wave; t < hi; t += 16) {
        const int u = ld_i(out + o + t);
        const int s = start_len[2 * u], len = start_len[2 * u + 1];
        int off = hi + ld_i(base + o + t);
        for (int b0 = 0; b0 < len; b0 += 64) {
          const int pos = b0 + lane;
          const int v = pos < len ? nbr[s + pos] : 0;
          const bool win = pos < len && ld_u(key + v) == (((unsigned int)t << 12) | (unsigned int)pos);
          const unsigned long long mask = __ballot(win);
          const int q = off + __popcll(mask & lt);
          if (win && q < m) { st_i(out + o + q, v); st_i(visited + v, 1); }
          off += __popcll(mask);
        }
      }
      lo = hi;
      hi = min(hi + total, m);
      __syncthreads();
    }
  }
}

// (cluster id, caller's point index) rows
__global__ void cluster_emit_kernel(int n, const int32_t *__restrict__ counters, const int32_t *__restrict__ cluster_offsets,
                                    const int32_t *__restrict__ out, const int32_t *__restrict__ point_index,
                                    int32_t *__restrict__ cluster_idxs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = counters[2], ncl = counters[3];
  if (q >= total) return;
  int lo = 0, hi = ncl;                          // last id with cluster_offsets[id] <= q
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cluster_offsets[mid] <= q) lo = mid; else hi = mid;
  }
  // a queue slot stays unwritten only when the lists were not symmetric (truncated lists: the caller discards the
  // result); never index with it
  const unsigned int v = (unsigned int)out[q];
  cluster_idxs[2 * q] = lo;
  cluster_idxs[2 * q + 1] = v < (unsigned int)n ? point_index[v] : -1;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_cluster_components_ws_bytes(int n) {
  if (n < 0) return -1;
  return 4L * (20L * n + 64 + 4L * scan_blocks(n + 1));
}

GCN_EXPORT int gcn_cluster_components(int n, const int32_t *nbr, const int32_t *start_len, const int32_t *seg_of,
                                      const int32_t *seg_offsets, const int32_t *seg_cls, int S,
                                      const int32_t *point_index, float size_threshold, void *ws, int32_t *cluster_idxs,
                                      int32_t *cluster_offsets, int32_t *counts, void *stream) {
  GCN_REQUIRE(counts, "gcn_cluster_components: counts is null");
  GCN_REQUIRE(n >= 0 && n < (1 << 20) && S >= 1, "gcn_cluster_components: n=%d must be below 2^20 (queue rank is a 20-bit key field)", n);
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(hipMemsetAsync(counts, 0, 2 * sizeof(int32_t), st));
  if (n == 0) return GCN_OK;
  GCN_REQUIRE(start_len && seg_of && seg_offsets && seg_cls && point_index && ws && cluster_idxs && cluster_offsets,
              "gcn_cluster_components: null pointer");
  int32_t *w = (int32_t *)ws;
  int32_t *counters = w;                       // [nwork, cursor, total, ncluster, ...]
  int32_t *parent = w + 16, *csize = parent + n, *comp = csize + n, *visited = comp + n, *base = visited + n;
  unsigned int *key = (unsigned int *)(base + n);
  int32_t *out = (int32_t *)key + n, *work = out + n;           // 3n
  int32_t *vals = work + 3L * n, *scan = vals + 4L * (n + 1);   // 2 x 4(n+1)
  int32_t *bsum = scan + 4L * (n + 1);
  cc_init_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, nbr, start_len, parent, csize, key, visited, counters);
  cc_union_kernel<<<cdiv(n, 4), 256, 0, st>>>(n, nbr, start_len, parent);
  cc_flatten_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, parent, comp, csize);
  cluster_classify_kernel<<<cdiv(n + 1, 256), 256, 0, st>>>(n, comp, csize, seg_of, seg_cls, size_threshold, vals);
  GCN_HIP(hipMemcpyAsync(scan, vals, sizeof(int32_t) * 4 * (size_t)(n + 1), hipMemcpyDeviceToDevice, st));
  exscan_rows(st, 4, n + 1, scan, bsum);
  cluster_offsets_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, comp, csize, seg_of, seg_offsets, vals, scan, cluster_offsets, out,
                                                       visited, work, counters);
  cluster_bfs_kernel<<<512, 1024, 0, st>>>(nbr, start_len, work, counters, key, visited, base, out);
  cluster_emit_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, counters, cluster_offsets, out, point_index, cluster_idxs);
  GCN_HIP(hipMemcpyAsync(counts, counters + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return check_launch("cluster kernels");
}
