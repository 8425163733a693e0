// voxelize_dev.hip -- voxelize_idx on the device (SG/src/voxelize/voxelize.cpp:11-165 is a host hash table in the
// reference, 50 ms at N = 100 000 here; SURVEY section 8f rank 2 asks for the (coords, input_map, rule book)
// triple without a host round trip).  Same results as the reference's insertion-ordered hash:
//   voxel ids are numbered by FIRST APPEARANCE in the input, rule rows list their points in input order.
// Sort-based: pack (b,x,y,z) into one u64 key (each component must fit 16 bits: spatial shapes are <= 2^16),
// stable radix sort of (key, i) -> runs of equal keys are voxels with ascending i inside; a voxel's first
// appearance is the i at its run head; sorting the run heads by that i gives the reference's numbering.
// The two radix sorts are rocPRIM device primitives (plumbing); the rest are the small kernels below.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace gcn {

struct VxHeader {          // head of the workspace
  int M, maxActive, bad, pad;
};

__global__ void vx_pack_kernel(const int64_t *__restrict__ coords, int N, int ncol, unsigned long long *__restrict__ keys,
                               int *__restrict__ ids, VxHeader *__restrict__ h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  unsigned long long k = 0;
  bool bad = false;
  for (int j = 0; j < ncol; ++j) {
    const int v = (int)coords[(long)i * ncol + j];          // voxelize.cpp:62-63 truncates to 32 bits too
    bad |= (unsigned)v > 65535u;
    k = (k << 16) | (unsigned long long)(v & 0xffff);
  }
  keys[i] = k;
  ids[i] = i;
  if (bad) h->bad = 1;
}

// head flags of the sorted keys -> seg_head[j] in {0,1}
__global__ void vx_heads_kernel(const unsigned long long *__restrict__ keys, int N, int *__restrict__ flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  flag[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

// seg[j] = inclusive scan of flags - 1; run heads record (first input id, head position)
__global__ void vx_runs_kernel(const int *__restrict__ flag, const int *__restrict__ seg_incl, const int *__restrict__ ids,
                               int N, int *__restrict__ run_first, int *__restrict__ run_pos, int *__restrict__ run_id,
                               VxHeader *__restrict__ h) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  if (flag[j]) {
    const int s = seg_incl[j] - 1;
    run_first[s] = ids[j];
    run_pos[s] = j;
    run_id[s] = s;
  }
  if (j == N - 1) h->M = seg_incl[j];
}

// after sorting runs by first appearance: voxel v <- run run_sorted[v]; rank_of_run[run] = v
__global__ void vx_rank_kernel(const int *__restrict__ run_sorted, int M, int *__restrict__ rank_of_run) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < M) rank_of_run[run_sorted[v]] = v;
}

__global__ void vx_inputmap_kernel(const int *__restrict__ seg_incl, const int *__restrict__ ids, const int *__restrict__ rank_of_run,
                                   const int *__restrict__ run_pos, int N, int M, int mode, int32_t *__restrict__ input_map,
                                   VxHeader *__restrict__ h) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int s = seg_incl[j] - 1;
  input_map[ids[j]] = rank_of_run[s];
  if (mode == 3 || mode == 4) {
    const int nxt = s + 1 < M ? run_pos[s + 1] : N;
    if (j == run_pos[s]) atomicMax(&h->maxActive, nxt - run_pos[s]);
  }
}

__global__ void vx_fill_kernel(const int64_t *__restrict__ coords, int ncol, const int *__restrict__ seg_incl,
                               const int *__restrict__ ids, const int *__restrict__ rank_of_run, const int *__restrict__ run_pos,
                               int N, int M, int mode, int W, int64_t *__restrict__ out_coords, int32_t *__restrict__ out_map) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int s = seg_incl[j] - 1, v = rank_of_run[s];
  const int head = run_pos[s], nxt = s + 1 < M ? run_pos[s + 1] : N;
  const int t = j - head, cnt = nxt - head;
  int32_t *row = out_map + (long)v * W;
  if (mode == 3 || mode == 4) {
    row[1 + t] = ids[j];
    if (t == 0) row[0] = cnt;
  } else if (t == 0) {                       // mode 0 unique / 1 front / 2 back (voxelize.cpp:131-151)
    row[0] = 1;
    row[1] = (mode == 2) ? ids[nxt - 1] : ids[j];
  }
  if (t == 0) {                              // voxelize.cpp:47-55: coords of the first listed input row
    const int src = (mode == 2) ? ids[nxt - 1] : ids[j];
    for (int q = 0; q < ncol; ++q) out_coords[(long)v * ncol + q] = coords[(long)src * ncol + q];
  }
}

struct VxLayout {
  unsigned long long *keys_a, *keys_b;
  int *ids_a, *ids_b, *flag, *seg, *run_first, *run_first_s, *run_pos, *run_id, *run_sorted, *rank;
  void *tmp;
  size_t tmp_bytes, total;
};

static VxLayout vx_layout(char *base, int N) {
  VxLayout L;
  size_t off = 64;
  auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += (bytes + 63) & ~(size_t)63; return p; };
  const size_t n = (size_t)(N > 0 ? N : 1);
  L.keys_a = (unsigned long long *)take(8 * n); L.keys_b = (unsigned long long *)take(8 * n);
  L.ids_a = (int *)take(4 * n); L.ids_b = (int *)take(4 * n);
  L.flag = (int *)take(4 * n); L.seg = (int *)take(4 * n);
  L.run_first = (int *)take(4 * n); L.run_first_s = (int *)take(4 * n);
  L.run_pos = (int *)take(4 * n); L.run_id = (int *)take(4 * n); L.run_sorted = (int *)take(4 * n); L.rank = (int *)take(4 * n);
  size_t t1 = 0, t2 = 0, t3 = 0;
  (void)rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr, n, 0, 64);
  (void)rocprim::radix_sort_pairs(nullptr, t2, (int *)nullptr, (int *)nullptr, (int *)nullptr, (int *)nullptr, n, 0, 32);
  (void)rocprim::inclusive_scan(nullptr, t3, (int *)nullptr, (int *)nullptr, n, rocprim::plus<int>());
  L.tmp_bytes = t1 > t2 ? (t1 > t3 ? t1 : t3) : (t2 > t3 ? t2 : t3);
  L.tmp = take(L.tmp_bytes + 256);
  L.total = off;
  return L;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_voxelize_idx_ws_bytes(int N) {
  if (N < 0) return -1;
  return (long)vx_layout(nullptr, N).total;
}

// Two-call protocol on DEVICE buffers.  Call 1 (output_coords == NULL): fills input_map (N) and returns M and maxActive
// through host ints (one stream synchronisation); the workspace keeps the sorted state.  Call 2 with the SAME ws:
// fills output_coords (M,ncol) i64 and output_map (M, maxActive+1) i32 (zero-filled by the call).
GCN_EXPORT int gcn_voxelize_idx(const int64_t *coords, int N, int ncol, int mode, int32_t *input_map, int *M_host,
                                int *maxActive_host, int64_t *output_coords, int32_t *output_map, void *ws, void *stream) {
  GCN_REQUIRE(M_host && maxActive_host && ws, "gcn_voxelize_idx: null pointer");
  GCN_REQUIRE(N >= 0 && (ncol == 3 || ncol == 4) && mode >= 0 && mode <= 4, "gcn_voxelize_idx: bad arguments (ncol 3|4, mode 0..4)");
  hipStream_t st = (hipStream_t)stream;
  if (N == 0) { *M_host = 0; *maxActive_host = 1; return GCN_OK; }
  GCN_REQUIRE(coords && input_map, "gcn_voxelize_idx: null pointer");
  VxHeader *h = (VxHeader *)ws;
  VxLayout L = vx_layout((char *)ws, N);
  const int nb = cdiv(N, 256);
  if (!output_coords) {
    VxHeader init = {0, 1, 0, 0};
    GCN_HIP(hipMemcpyAsync(h, &init, sizeof(init), hipMemcpyHostToDevice, st));
    vx_pack_kernel<<<nb, 256, 0, st>>>(coords, N, ncol, L.keys_a, L.ids_a, h);
    size_t tb = L.tmp_bytes;
    GCN_HIP(rocprim::radix_sort_pairs(L.tmp, tb, L.keys_a, L.keys_b, L.ids_a, L.ids_b, (size_t)N, 0, 16 * ncol, st));
    vx_heads_kernel<<<nb, 256, 0, st>>>(L.keys_b, N, L.flag);
    tb = L.tmp_bytes;
    GCN_HIP(rocprim::inclusive_scan(L.tmp, tb, L.flag, L.seg, (size_t)N, rocprim::plus<int>(), st));
    vx_runs_kernel<<<nb, 256, 0, st>>>(L.flag, L.seg, L.ids_b, N, L.run_first, L.run_pos, L.run_id, h);
    VxHeader hh;
    GCN_HIP(hipMemcpyAsync(&hh, h, sizeof(hh), hipMemcpyDeviceToHost, st));
    GCN_HIP(hipStreamSynchronize(st));
    if (hh.bad) {
      set_error("gcn_voxelize_idx: a coordinate is outside [0, 65535] (use the host routine)");
      return GCN_EINVAL;
    }
    const int M = hh.M;
    tb = L.tmp_bytes;
    GCN_HIP(rocprim::radix_sort_pairs(L.tmp, tb, L.run_first, L.run_first_s, L.run_id, L.run_sorted, (size_t)M, 0, 32, st));
    vx_rank_kernel<<<cdiv(M, 256), 256, 0, st>>>(L.run_sorted, M, L.rank);
    vx_inputmap_kernel<<<nb, 256, 0, st>>>(L.seg, L.ids_b, L.rank, L.run_pos, N, M, mode, input_map, h);
    GCN_HIP(hipMemcpyAsync(&hh, h, sizeof(hh), hipMemcpyDeviceToHost, st));
    GCN_HIP(hipStreamSynchronize(st));
    *M_host = M;
    *maxActive_host = hh.maxActive;
    return check_launch("voxelize_idx (pass 1)");
  }
  GCN_REQUIRE(output_map, "gcn_voxelize_idx: output_map is null");
  VxHeader hh;
  GCN_HIP(hipMemcpyAsync(&hh, h, sizeof(hh), hipMemcpyDeviceToHost, st));
  GCN_HIP(hipStreamSynchronize(st));
  const int M = hh.M, W = hh.maxActive + 1;
  GCN_HIP(fill_dev(output_map, 0, sizeof(int32_t) * (size_t)M * W, st));
  vx_fill_kernel<<<nb, 256, 0, st>>>(coords, ncol, L.seg, L.ids_b, L.rank, L.run_pos, N, M, mode, W, output_coords, output_map);
  *M_host = M;
  *maxActive_host = hh.maxActive;
  return check_launch("voxelize_idx (pass 2)");
}
