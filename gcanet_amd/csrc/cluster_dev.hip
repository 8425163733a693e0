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

// every point directly under the root of its initial tree: the union pass then looks roots up in one or two steps
// instead of walking the chains "point -> its smallest neighbour -> ..." of the initial forest edge by edge
__global__ void cc_compress_kernel(int n, int32_t *parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int r = i;
  for (;;) {                                   // parents only decrease and nobody hooks yet: a plain walk
    const int p = ld_i(parent + r);
    if (p == r) break;
    r = p;
  }
  st_i(parent + i, r);
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
        const bool all_fragments = size_threshold < -1.5f;                     // set aggregation needs the dropped ones too
        if (sz < high) { if (sz >= low || all_fragments) ks = sz; }
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

// COMPACT: `nbr` is CLOBBERED -- the pass that counts a frontier node's winners compacts them to the front of the node's
// own list (a node is in the frontier once and only its own wave reads its list), so the append pass copies winners, one
// per point of the component, instead of scanning every list a third time (gcn_cluster_components_clobber).
template <bool COMPACT>
__global__ __launch_bounds__(1024) void cluster_bfs_kernel(int32_t *nbr, const int32_t *__restrict__ start_len,
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
          const int v = pos < len ? nbr[s + pos] : 0;
          const bool win = pos < len && ld_u(key + v) == (((unsigned int)t << 12) | (unsigned int)pos);
          const unsigned long long mask = __ballot(win);
          if (COMPACT && win) nbr[s + w + __popcll(mask & lt)] = v;  // w + rank <= pos: behind this wave's own reads
          w += __popcll(mask);
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
        const int off0 = ld_i(base + o + t);
        if (COMPACT) {                                             // they sit compacted at the front of the node's list
          const int cnt = (t + 1 < hi ? ld_i(base + o + t + 1) : total) - off0;
          for (int i = lane; i < cnt; i += 64) {
            const int v = nbr[s + i];
            const int q = hi + off0 + i;
            if (q < m) { st_i(out + o + q, v); st_i(visited + v, 1); }
          }
        } else {
          int off = hi + off0;
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


// ---------------------------------------------------------------- set aggregation (using_set_aggr = True, evaluation)
// hierarchical_aggregation.cu:22-196 + the merge of functions.py:52-72: every fragment (component below 0.3*mean points)
// looks for the nearest primary of its (cloud, class) subset; if that centroid lies within 0.01*sqrt(|primary|) it is
// absorbed -- its points are appended to the primary's cluster, fragments in index order, at most 1000 fragments and
// 3000 points per primary.  Input: ALL components of every segment from gcn_cluster_components(size_threshold = -2):
// per segment the fragments in discovery order, then the primaries, members in BFS order (rows hold the sorted-array
// index v of each member).  Centroids are sequential f32 sums in member order, distances (dx*dx + dy*dy) + dz*dz, the
// absorption order is the fragment index order: the oracle's arithmetic and order (oracle/gcanet_oracle.c:
// orc_hier_set_aggr; the reference's own order comes from atomicAdd and is unspecified).
struct SetAggrArgs {
  const int32_t *counts_in;     // (2) device: rows, clusters of the input
  const int32_t *rows;          // (rows,2): (cluster id, v)
  const int32_t *offs;          // (clusters+1)
  const int32_t *seg_of;        // (n)
  const int32_t *seg_cls;       // (S)
  const float *xyz;             // (n,3) shifted coordinates in sorted order
  const int32_t *point_index;   // (n) caller's index of sorted position v
  int n, S;
  int32_t *cfirst, *cend;       // (S) cluster range of a segment (cfirst = INT_MAX when it has none)
  float *cen;                   // (n,3)
  int32_t *owner, *take, *apos, *nfrag, *apts, *oloc, *ooff;   // (n) each
  int32_t *segcl, *segpts, *basecl, *basepts;                  // (S) each
  int32_t *out_idxs, *out_offs, *out_counts;
};

__device__ __forceinline__ bool sa_is_primary(int cls, int sz) {
  const float class_mean[10] = {-1.f, -1.f, 3917.f, 12056.f, 2303.f, 8331.f, 3948.f, 3166.f, 5629.f, 11719.f};
  const float high = (float)(0.3 * class_mean[cls]);
  return !(sz < high);
}
__device__ __forceinline__ bool sa_is_kept(int cls, int sz) {
  const float class_mean[10] = {-1.f, -1.f, 3917.f, 12056.f, 2303.f, 8331.f, 3948.f, 3166.f, 5629.f, 11719.f};
  const float low = (float)(0.05 * class_mean[cls]), high = (float)(0.3 * class_mean[cls]);
  return sz < high && sz >= low;
}

__global__ void sa_init_kernel(SetAggrArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.S) { a.cfirst[i] = 0x7fffffff; a.cend[i] = 0; a.segcl[i] = 0; a.segpts[i] = 0; }
}

__global__ void sa_ranges_kernel(SetAggrArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.counts_in[1]) return;
  const int s = a.seg_of[a.rows[2L * a.offs[c] + 1]];
  atomicMin(a.cfirst + s, c);
  atomicMax(a.cend + s, c + 1);
}

// one workgroup per segment
__global__ __launch_bounds__(256) void sa_segment_kernel(SetAggrArgs a) {
  const int s = blockIdx.x;
  const int c0 = a.cfirst[s], c1 = a.cend[s];
  if (c0 >= c1) return;
  const int cls = a.seg_cls[s];
  // centroids: sequential f32 sums in member (BFS) order, as the reference accumulates them
  for (int c = c0 + threadIdx.x; c < c1; c += 256) {
    float ax = 0.f, ay = 0.f, az = 0.f;
    const int b = a.offs[c], e = a.offs[c + 1];
    for (int q = b; q < e; ++q) {
      const int v = a.rows[2L * q + 1];
      ax += a.xyz[3L * v]; ay += a.xyz[3L * v + 1]; az += a.xyz[3L * v + 2];
    }
    const float sz = (float)(e - b);
    a.cen[3L * c] = ax / sz; a.cen[3L * c + 1] = ay / sz; a.cen[3L * c + 2] = az / sz;
    a.nfrag[c] = 0; a.apts[c] = 0; a.take[c] = 0; a.apos[c] = 0; a.owner[c] = -1; a.oloc[c] = -1; a.ooff[c] = 0;
  }
  __syncthreads();
  // primaries are the tail of the segment's cluster range
  __shared__ int p0_s;
  if (threadIdx.x == 0) {
    int p0 = c1;
    while (p0 > c0 && sa_is_primary(cls, a.offs[p0] - a.offs[p0 - 1])) --p0;
    p0_s = p0;
  }
  __syncthreads();
  const int p0 = p0_s;
  for (int f = c0 + threadIdx.x; f < p0; f += 256) {             // hierarchical_aggregation.cu:22-75
    float nearest = 10000.f;
    int ni = -1;
    for (int i = p0; i < c1; ++i) {
      const float dx = a.cen[3L * i] - a.cen[3L * f], dy = a.cen[3L * i + 1] - a.cen[3L * f + 1],
                  dz = a.cen[3L * i + 2] - a.cen[3L * f + 2];
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < nearest) { nearest = d; ni = i; }
    }
    if (ni >= 0) {
      const int pn = a.offs[ni + 1] - a.offs[ni];
      const float r = (float)(0.01 * sqrtf((float)pn));
      if (nearest < r * r) a.owner[f] = ni;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // absorption in fragment index order with the reference's caps (hierarchical_aggregation.cu:9-12)
    for (int f = c0; f < p0; ++f) {
      const int o = a.owner[f];
      if (o < 0 || a.nfrag[o] >= 1000) continue;
      a.nfrag[o] += 1;
      const int sz = a.offs[f + 1] - a.offs[f];
      int t = 3000 - a.apts[o];
      t = t < 0 ? 0 : (t > sz ? sz : t);
      a.take[f] = t;
      a.apos[f] = a.apts[o];
      a.apts[o] += t;
    }
    // layout inside the segment: kept fragments, then primaries (each with what it absorbed)
    int ncl = 0, npt = 0;
    for (int c = c0; c < c1; ++c) {
      const int sz = a.offs[c + 1] - a.offs[c];
      const bool prim = c >= p0;
      if (!prim && !sa_is_kept(cls, sz)) continue;
      a.oloc[c] = ncl++;
      a.ooff[c] = npt;
      npt += sz + (prim ? a.apts[c] : 0);
    }
    a.segcl[s] = ncl;
    a.segpts[s] = npt;
  }
}

__global__ void sa_scan_kernel(SetAggrArgs a) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int cl = 0, pt = 0;
  for (int s = 0; s < a.S; ++s) {
    a.basecl[s] = cl; a.basepts[s] = pt;
    cl += a.segcl[s]; pt += a.segpts[s];
  }
  a.out_offs[cl] = pt;
  a.out_counts[0] = pt;
  a.out_counts[1] = cl;
}

// one workgroup per input cluster: its own copy (kept fragment / primary) and, for an absorbed fragment, the copy
// behind its owner's points
__global__ __launch_bounds__(256) void sa_emit_kernel(SetAggrArgs a) {
  const int c = blockIdx.x;
  if (c >= a.counts_in[1]) return;
  const int b = a.offs[c], sz = a.offs[c + 1] - b;
  const int s = a.seg_of[a.rows[2L * b + 1]];
  if (a.oloc[c] >= 0) {
    const int id = a.basecl[s] + a.oloc[c], start = a.basepts[s] + a.ooff[c];
    if (threadIdx.x == 0) a.out_offs[id] = start;
    for (int t = threadIdx.x; t < sz; t += 256) {
      a.out_idxs[2L * (start + t)] = id;
      a.out_idxs[2L * (start + t) + 1] = a.point_index[a.rows[2L * (b + t) + 1]];
    }
  }
  const int o = a.owner[c], tk = a.take[c];
  if (o >= 0 && tk > 0) {
    const int id = a.basecl[s] + a.oloc[o];
    const int start = a.basepts[s] + a.ooff[o] + (a.offs[o + 1] - a.offs[o]) + a.apos[c];
    for (int t = threadIdx.x; t < tk; t += 256) {
      a.out_idxs[2L * (start + t)] = id;
      a.out_idxs[2L * (start + t) + 1] = a.point_index[a.rows[2L * (b + t) + 1]];
    }
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_cluster_components_ws_bytes(int n) {
  if (n < 0) return -1;
  return 4L * (20L * n + 64 + 4L * scan_blocks(n + 1));
}

static int cluster_components(int n, int32_t *nbr, bool clobber, const int32_t *start_len, const int32_t *seg_of,
                              const int32_t *seg_offsets, const int32_t *seg_cls, int S, const int32_t *point_index,
                              float size_threshold, void *ws, int32_t *cluster_idxs, int32_t *cluster_offsets,
                              int32_t *counts, void *stream) {
  GCN_REQUIRE(counts, "gcn_cluster_components: counts is null");
  GCN_REQUIRE(n >= 0 && n < (1 << 20) && S >= 1, "gcn_cluster_components: n=%d must be below 2^20 (queue rank is a 20-bit key field)", n);
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(counts, 0, 2 * sizeof(int32_t), st));
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
  cc_compress_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, parent);
  cc_union_kernel<<<cdiv(n, 4), 256, 0, st>>>(n, nbr, start_len, parent);
  cc_flatten_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, parent, comp, csize);
  cluster_classify_kernel<<<cdiv(n + 1, 256), 256, 0, st>>>(n, comp, csize, seg_of, seg_cls, size_threshold, vals);
  GCN_HIP(hipMemcpyAsync(scan, vals, sizeof(int32_t) * 4 * (size_t)(n + 1), hipMemcpyDeviceToDevice, st));
  exscan_rows(st, 4, n + 1, scan, bsum);
  cluster_offsets_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, comp, csize, seg_of, seg_offsets, vals, scan, cluster_offsets, out,
                                                       visited, work, counters);
  if (clobber) cluster_bfs_kernel<true><<<512, 1024, 0, st>>>(nbr, start_len, work, counters, key, visited, base, out);
  else cluster_bfs_kernel<false><<<512, 1024, 0, st>>>(nbr, start_len, work, counters, key, visited, base, out);
  cluster_emit_kernel<<<cdiv(n, 256), 256, 0, st>>>(n, counters, cluster_offsets, out, point_index, cluster_idxs);
  GCN_HIP(hipMemcpyAsync(counts, counters + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return check_launch("cluster kernels");
}

GCN_EXPORT int gcn_cluster_components(int n, const int32_t *nbr, const int32_t *start_len, const int32_t *seg_of,
                                      const int32_t *seg_offsets, const int32_t *seg_cls, int S,
                                      const int32_t *point_index, float size_threshold, void *ws, int32_t *cluster_idxs,
                                      int32_t *cluster_offsets, int32_t *counts, void *stream) {
  return cluster_components(n, const_cast<int32_t *>(nbr), false, start_len, seg_of, seg_offsets, seg_cls, S, point_index,
                            size_threshold, ws, cluster_idxs, cluster_offsets, counts, stream);
}

GCN_EXPORT int gcn_cluster_components_clobber(int n, int32_t *nbr, const int32_t *start_len, const int32_t *seg_of,
                                              const int32_t *seg_offsets, const int32_t *seg_cls, int S,
                                              const int32_t *point_index, float size_threshold, void *ws,
                                              int32_t *cluster_idxs, int32_t *cluster_offsets, int32_t *counts, void *stream) {
  return cluster_components(n, nbr, true, start_len, seg_of, seg_offsets, seg_cls, S, point_index, size_threshold, ws,
                            cluster_idxs, cluster_offsets, counts, stream);
}

GCN_EXPORT long gcn_set_aggregation_ws_bytes(int n, int S) {
  if (n < 0 || S < 1) return -1;
  return 4L * (10L * n + 6L * S + 64);
}

GCN_EXPORT int gcn_set_aggregation(int n, int S, const int32_t *counts_in, const int32_t *rows, const int32_t *offs,
                                   const int32_t *seg_of, const int32_t *seg_cls, const float *xyz,
                                   const int32_t *point_index, void *ws, int32_t *out_idxs, int32_t *out_offs,
                                   int32_t *out_counts, void *stream) {
  GCN_REQUIRE(n >= 1 && S >= 1 && counts_in && rows && offs && seg_of && seg_cls && xyz && point_index && ws && out_idxs &&
              out_offs && out_counts, "gcn_set_aggregation: bad argument");
  hipStream_t st = (hipStream_t)stream;
  SetAggrArgs a{};
  a.counts_in = counts_in; a.rows = rows; a.offs = offs; a.seg_of = seg_of; a.seg_cls = seg_cls; a.xyz = xyz;
  a.point_index = point_index; a.n = n; a.S = S;
  int32_t *w = (int32_t *)ws;
  a.cfirst = w; a.cend = w + S; a.segcl = w + 2 * S; a.segpts = w + 3 * S; a.basecl = w + 4 * S; a.basepts = w + 5 * S;
  int32_t *p = w + 6 * S + 8;
  a.cen = (float *)p; p += 3L * n;
  a.owner = p; p += n; a.take = p; p += n; a.apos = p; p += n; a.nfrag = p; p += n; a.apts = p; p += n;
  a.oloc = p; p += n; a.ooff = p;
  a.out_idxs = out_idxs; a.out_offs = out_offs; a.out_counts = out_counts;
  sa_init_kernel<<<cdiv(S, 256), 256, 0, st>>>(a);
  sa_ranges_kernel<<<cdiv(n, 256), 256, 0, st>>>(a);
  sa_segment_kernel<<<S, 256, 0, st>>>(a);
  sa_scan_kernel<<<1, 64, 0, st>>>(a);
  sa_emit_kernel<<<n, 256, 0, st>>>(a);
  return check_launch("set aggregation kernels");
}
