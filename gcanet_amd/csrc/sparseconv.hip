// sparseconv.hip -- the sparse 3-D convolutions of the instance-refinement "tiny U-Net" (softgroup/model/blocks.py:
// 44-143, models/dgcnn-hais-concat-direct-4.py:611-616,1379-1392).  The reference takes them from the third-party
// `spconv` package (not vendored, no version pinned); this file states the same operators on the format
// clusters_voxelization already produces -- features (M,C) f32, coords (M,4) int32 [sample,x,y,z] -- as
// gather -> GEMM on the f32 matrix cores.  SURVEY.md section 8(f) rank 3.
//
//   rule tables   MI355X has 288 GB: the voxel index is a DENSE int32 grid [sample][x][y][z] (210 MB for 200 proposals
//                 at 64^3), so a neighbour lookup is one load -- no hash table, no sort.
//                   submanifold 3x3x3:  nbr (M,27): index of the voxel at offset (dx,dy,dz), k = (dx+1)*9+(dy+1)*3+dz+1
//                   stride-2 2x2x2:     coarse voxels numbered in key order (flag grid + exclusive scan);
//                                       child (M2,8) for the forward, parent slot (M,8) (one entry) for the inverse
//   sc_gather_gemm   out[o, :] = sum_k in[rule[o,k], :] . W[k]   (rule < 0: no contribution).  One kernel serves the
//                 forward of all three conv types and every input gradient (the transposed rule table of a
//                 submanifold conv is its own column reversal; strided and inverse conv are each other's transpose).
//                 A wave owns 64 output rows x 64 output columns: W[k] fragments (64x64) live in 64 VGPRs while the
//                 four 16-row tiles are gathered (one float4 per lane per 16 input columns; lane group g = lane/16
//                 supplies columns 4g..4g+3 as the k index of four consecutive v_mfma_f32_16x16x4_f32).
//   sc_wgrad      dW[k] = sum_o in[rule[o,k], :]^T (x) dOut[o, :]: rows are the MFMA k dimension; 16 accumulator tiles
//                 per wave, f32 atomics into dW at the end.
#include "common.h"

namespace gcn {

using sc_f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

// ---------------------------------------------------------------- rule tables
__global__ void sc_grid_fill_kernel(int M, const int32_t *__restrict__ coords, int D, int32_t *__restrict__ grid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int32_t *c = coords + 4L * i;
  grid[(((long)c[0] * D + c[1]) * D + c[2]) * D + c[3]] = i;
}

__global__ void sc_subm_rules_kernel(int M, const int32_t *__restrict__ coords, int D, const int32_t *__restrict__ grid,
                                     int32_t *__restrict__ nbr) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 27L * M) return;
  const int i = (int)(t / 27), k = (int)(t % 27);
  const int32_t *c = coords + 4L * i;
  const int x = c[1] + k / 9 - 1, y = c[2] + (k / 3) % 3 - 1, z = c[3] + k % 3 - 1;
  int v = -1;
  if (x >= 0 && x < D && y >= 0 && y < D && z >= 0 && z < D) v = grid[(((long)c[0] * D + x) * D + y) * D + z];
  nbr[t] = v;
}

// flag the coarse cells that own a voxel
__global__ void sc_coarse_flag_kernel(int M, const int32_t *__restrict__ coords, int D2, int32_t *__restrict__ flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int32_t *c = coords + 4L * i;
  flag[(((long)c[0] * D2 + (c[1] >> 1)) * D2 + (c[2] >> 1)) * D2 + (c[3] >> 1)] = 1;
}

// after the exclusive scan: rank[cell] = coarse voxel id of a flagged cell.  Writes the coarse coords, child / parent tables.
__global__ void sc_coarse_rules_kernel(int M, const int32_t *__restrict__ coords, int D2, const int32_t *__restrict__ rank,
                                       int32_t *__restrict__ coords2, int32_t *__restrict__ child,
                                       int32_t *__restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int32_t *c = coords + 4L * i;
  const int o = rank[(((long)c[0] * D2 + (c[1] >> 1)) * D2 + (c[2] >> 1)) * D2 + (c[3] >> 1)];
  const int k = ((c[1] & 1) * 2 + (c[2] & 1)) * 2 + (c[3] & 1);
  child[8L * o + k] = i;
  parent[8L * i + k] = o;
  coords2[4L * o] = c[0]; coords2[4L * o + 1] = c[1] >> 1; coords2[4L * o + 2] = c[2] >> 1; coords2[4L * o + 3] = c[3] >> 1;  // same value from every child
}

// ---------------------------------------------------------------- out = sum_k gather(in, rule[:,k]) . W[k]
// W (K, Cin, Cout) row-major.  KREV: use rule column K-1-k with weight k (the transposed rule table of a submanifold
// convolution).  Cin, Cout multiples of 64.
//
// Voxelised surfaces fill ~10-30 % of the 27 neighbour slots, so a 16-row MFMA tile taken from consecutive output rows
// would be mostly zeros.  A wave therefore owns 64 output rows x 64 output columns and per offset k compacts the rows that
// do have a neighbour (ballot) into tiles of 16 (row, source) pairs: the MFMA count follows the pairs, not the rows.
// Layout: lane = output row (64 rows per wave), 64 accumulator VGPRs = the row's 64 output columns.  A tile's 16x64
// product goes through a 4 KB per-wave LDS staging buffer, from which the (at most 16) lanes owning its rows pick their
// row up.  The 64x64 block of W[k] is staged in LDS for the four waves of the workgroup and read as the B operand with
// one ds_read per MFMA (row stride 68 floats keeps the four lane groups on disjoint banks).
// At a few hundred voxels per proposal a launch has only ~130 workgroups, so its duration is the LATENCY of one
// workgroup's 27 offsets: the loop is software pipelined -- while offset k is multiplied, the weight block of the next
// phase is already in flight to registers (written to the other LDS buffer afterwards: one barrier per phase), the
// next offset's pair list is built from a rule entry loaded two offsets ahead, and the source rows of its first tile
// are being gathered.
template <bool KREV>
__global__ __launch_bounds__(256) void sc_gather_gemm_kernel(int Mout, int K, int Cin, int Cout, const float *__restrict__ in,
                                                             const int32_t *__restrict__ rule, const float *__restrict__ W,
                                                             float *__restrict__ out) {
  constexpr int LD = 64 + 4;
  __shared__ float wt[2][64 * LD];
  __shared__ float stage[4][16 * LD];
  __shared__ int psrc[4][2][64];
  const int lane = lane_id(), wave = wave_id();
  const int li = lane & 15, lk = lane >> 4;
  const int r0 = (blockIdx.x * 4 + wave) * 64;                 // may lie past the end: the wave still helps staging W
  const int n0 = blockIdx.y * 64;
  float *stg = stage[wave];
  const int myrow = r0 + lane;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // blockIdx.z selects a contiguous group of offsets (small launches are split so that enough workgroups exist and the
  // latency of a workgroup's offset loop shrinks); group z writes its partial sums to out + z * Mout * Cout
  const int ksplit = gridDim.z;
  const int k_begin = (int)((long)blockIdx.z * K / ksplit), k_end = (int)((long)(blockIdx.z + 1) * K / ksplit);
  const int nc = Cin / 64, P = (k_end - k_begin) * nc;
  out += (long)blockIdx.z * Mout * Cout;
  float acc[64];
#pragma unroll
  for (int j = 0; j < 64; ++j) acc[j] = 0.f;

  auto load_rule = [&](int kk) -> int {
    const int ka = k_begin + kk;
    return (myrow < Mout && ka < k_end) ? rule[(long)myrow * K + (KREV ? K - 1 - ka : ka)] : -1;
  };
  auto load_w = [&](int ph, float4 (&r)[4]) {
    const int k = k_begin + ph / nc, c0 = (ph % nc) * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + 256 * u, row = i >> 4, c4 = (i & 15) * 4;
      r[u] = *reinterpret_cast<const float4 *>(W + ((long)k * Cin + c0 + row) * Cout + n0 + c4);
    }
  };
  auto store_w = [&](int buf, const float4 (&r)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = threadIdx.x + 256 * u, row = i >> 4, c4 = (i & 15) * 4;
      float *d = wt[buf] + row * LD + c4;
      d[0] = r[u].x; d[1] = r[u].y; d[2] = r[u].z; d[3] = r[u].w;
    }
  };
  auto gather = [&](const int *list, int cnt, int j0, int c0, float4 (&a)[4]) {
    const int pa = j0 + li;
    const bool ok = pa < cnt;
    const float *arow = in + (long)(ok ? list[pa] : 0) * Cin + c0 + 4 * lk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = *reinterpret_cast<const float4 *>(arow + 16 * q);
      if (!ok) a[q] = float4{0.f, 0.f, 0.f, 0.f};
    }
  };

  int src_c = load_rule(0), src_n = load_rule(1);
  unsigned long long mask = __ballot(src_c >= 0);
  int c_c = __popcll(mask), pos_c = __popcll(mask & lt);
  if (src_c >= 0) psrc[wave][0][pos_c] = src_c;
  float4 wr[4], a_next[4];
  load_w(0, wr);
  store_w(0, wr);
  __builtin_amdgcn_wave_barrier();
  gather(psrc[wave][0], c_c, 0, 0, a_next);
  __syncthreads();

  for (int ph = 0; ph < P; ++ph) {
    const int k = ph / nc, c0 = (ph % nc) * 64, buf = ph & 1;
    const bool last_chunk = (ph % nc) == nc - 1;
    const int *list = psrc[wave][k & 1];
    if (ph + 1 < P) load_w(ph + 1, wr);
    int src_nn = -1, c_n = 0, pos_n = 0;
    if (last_chunk) {                                           // the next phase starts offset k+1: build its pair list now
      src_nn = load_rule(k + 2);
      const unsigned long long mn = __ballot(src_n >= 0);
      c_n = __popcll(mn);
      pos_n = __popcll(mn & lt);
      if (src_n >= 0) psrc[wave][(k + 1) & 1][pos_n] = src_n;
      __builtin_amdgcn_wave_barrier();
    }
    float4 a[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = a_next[q];
    if (ph + 1 < P) {
      if (last_chunk) gather(psrc[wave][(k + 1) & 1], c_n, 0, 0, a_next);
      else gather(list, c_c, 0, c0 + 64, a_next);
    }
    for (int j0 = 0; j0 < c_c; j0 += 16) {
      if (j0 > 0) gather(list, c_c, j0, c0, a);
      sc_f32x4 d[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) d[t] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // B[kdim = lk][col = li] of step s = W[k][c0 + 16q + 4lk + s][n0 + 16t + li]
        const float *wq = wt[buf] + (16 * q + 4 * lk) * LD + li;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, wq[0 * LD + 16 * t], d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, wq[1 * LD + 16 * t], d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, wq[2 * LD + 16 * t], d[t], 0, 0, 0);
          d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, wq[3 * LD + 16 * t], d[t], 0, 0, 0);
        }
      }
      // D[i = 4*lk + e][j = li] -> staging row i; the lane whose pair sits in slot i adds the row to its accumulators
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < 4; ++t) stg[(4 * lk + e) * LD + 16 * t + li] = d[t][e];
      __builtin_amdgcn_wave_barrier();
      if (src_c >= 0 && pos_c >= j0 && pos_c < j0 + 16) {
        const float4 *sp = reinterpret_cast<const float4 *>(stg + (pos_c - j0) * LD);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 v = sp[j];
          acc[4 * j] += v.x; acc[4 * j + 1] += v.y; acc[4 * j + 2] += v.z; acc[4 * j + 3] += v.w;
        }
      }
    }
    if (ph + 1 < P) store_w(buf ^ 1, wr);                        // last read in phase ph-1, before the previous barrier
    __syncthreads();
    if (last_chunk) { src_c = src_n; c_c = c_n; pos_c = pos_n; src_n = src_nn; }
  }
  if (myrow < Mout) {
    float4 *op = reinterpret_cast<float4 *>(out + (long)myrow * Cout + n0);
#pragma unroll
    for (int j = 0; j < 16; ++j) op[j] = float4{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
  }
}

// out[i] = sum_z part[z][i] (fixed order: deterministic), float4 per thread
__global__ void sc_sum_parts_kernel(long n4, int parts, const float4 *__restrict__ part, float4 *__restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = part[i];
  for (int z = 1; z < parts; ++z) {
    const float4 b = part[(long)z * n4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  out[i] = a;
}

// ---------------------------------------------------------------- dW[k] = sum_o gather(in, rule[o,k])^T (x) dOut[o]
// grid (row chunks, K, (Cin/64)*(Cout/64)); ruleT (K, Mout) is the transposed rule table (coalesced column reads).  A
// wave takes 256 rows at a time, compacts the (o, source) pairs of offset k and feeds them four at a time as the MFMA k
// dimension; the 64x64 block of dW[k] is 16 accumulator tiles.  dW (K, Cin, Cout) must be zero on entry.
__global__ __launch_bounds__(256) void sc_wgrad_kernel(int Mout, int K, int Cin, int Cout, int rows_per_block,
                                                       const float *__restrict__ in, const int32_t *__restrict__ ruleT,
                                                       const float *__restrict__ dout, float *__restrict__ dW) {
  __shared__ int po[4][256], psr[4][256];
  const int lane = lane_id(), wave = wave_id();
  const int li = lane & 15, lk = lane >> 4;
  const int k = blockIdx.y;
  const int c0 = (blockIdx.z / (Cout / 64)) * 64, n0 = (blockIdx.z % (Cout / 64)) * 64;
  const int b0 = blockIdx.x * rows_per_block, b1 = min(b0 + rows_per_block, Mout);
  const int32_t *rk = ruleT + (long)k * Mout;
  int *qo = po[wave], *qs = psr[wave];
  const unsigned long long lt = (1ull << lane) - 1ull;
  sc_f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[a][t] = {0.f, 0.f, 0.f, 0.f};
  for (int o0 = b0 + 256 * wave; o0 < b1; o0 += 1024) {
    int c = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + 64 * u + lane;
      const int src = o < b1 ? rk[o] : -1;
      const unsigned long long mask = __ballot(src >= 0);
      if (src >= 0) { const int pos = c + __popcll(mask & lt); qo[pos] = o; qs[pos] = src; }
      c += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    // operands of step p0+4 are fetched before the MFMAs of step p0 issue (the loop is latency-bound otherwise)
    float an[4], bn[4];
    auto fetch = [&](int p0) {
      const int p = p0 + lk;
      const bool ok = p < c;
      const float *xi = in + (long)(ok ? qs[p] : 0) * Cin + c0 + li;
      const float *dy = dout + (long)(ok ? qo[p] : 0) * Cout + n0 + li;
#pragma unroll
      for (int a = 0; a < 4; ++a) { const float v = xi[16 * a]; an[a] = ok ? v : 0.f; }
#pragma unroll
      for (int t = 0; t < 4; ++t) { const float v = dy[16 * t]; bn[t] = ok ? v : 0.f; }
    };
    if (c > 0) fetch(0);
    for (int p0 = 0; p0 < c; p0 += 4) {
      float av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) { av[a] = an[a]; bv[a] = bn[a]; }
      if (p0 + 4 < c) fetch(p0 + 4);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
    }
  }
  // D[i = c = 4*lk + e][j = n = li]: the four waves add their blocks in LDS, then ONE atomic per element and workgroup
  __shared__ float red[64 * 64];
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float *d = red + (16 * a + 4 * lk + e) * 64 + 16 * t + li;
            *d = (w ? *d : 0.f) + acc[a][t][e];
          }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const float v = red[i];
    if (v != 0.f) atomicAdd(dW + ((long)k * Cin + c0 + (i >> 6)) * Cout + n0 + (i & 63), v);
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_sparse_grid_bytes(int batch, int D) {
  if (batch < 0 || D <= 0) return -1;
  return 4L * batch * D * D * D;
}

GCN_EXPORT int gcn_sparse_subm_rules(int M, const int32_t *coords, int batch, int D, int32_t *grid, int32_t *nbr, void *stream) {
  GCN_REQUIRE(M >= 0 && batch >= 1 && D >= 1, "gcn_sparse_subm_rules: bad shape");
  if (M == 0) return GCN_OK;
  GCN_REQUIRE(coords && grid && nbr, "gcn_sparse_subm_rules: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(grid, 0xff, (size_t)gcn_sparse_grid_bytes(batch, D), st));
  sc_grid_fill_kernel<<<cdiv(M, 256), 256, 0, st>>>(M, coords, D, grid);
  sc_subm_rules_kernel<<<cdiv(27L * M, 256), 256, 0, st>>>(M, coords, D, grid, nbr);
  return check_launch("sc_subm_rules_kernel");
}

GCN_EXPORT long gcn_sparse_coarse_ws_bytes(int batch, int D) {
  if (batch < 0 || D <= 0) return -1;
  const long cells = (long)batch * ((D + 1) / 2) * ((D + 1) / 2) * ((D + 1) / 2) + 1;
  return 4L * (cells + scan_blocks(cells) + 16);
}

GCN_EXPORT int gcn_sparse_coarse_rules(int M, const int32_t *coords, int batch, int D, void *ws, int32_t *coords2,
                                       int32_t *child, int32_t *parent, int32_t *m2_dev, void *stream) {
  GCN_REQUIRE(M >= 0 && batch >= 1 && D >= 1, "gcn_sparse_coarse_rules: bad shape");
  GCN_REQUIRE(m2_dev, "gcn_sparse_coarse_rules: m2_dev is null");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(m2_dev, 0, sizeof(int32_t), st));
  if (M == 0) return GCN_OK;
  GCN_REQUIRE(coords && ws && coords2 && child && parent, "gcn_sparse_coarse_rules: null pointer");
  const int D2 = (D + 1) / 2;
  const long cells = (long)batch * D2 * D2 * D2 + 1;
  GCN_REQUIRE(cells < (1L << 31), "gcn_sparse_coarse_rules: grid too large");
  int32_t *flag = (int32_t *)ws, *bsum = flag + cells;
  GCN_HIP(fill_dev(flag, 0, sizeof(int32_t) * (size_t)cells, st));
  GCN_HIP(fill_dev(child, 0xff, sizeof(int32_t) * 8 * (size_t)M, st));      // at most M coarse voxels
  GCN_HIP(fill_dev(parent, 0xff, sizeof(int32_t) * 8 * (size_t)M, st));
  sc_coarse_flag_kernel<<<cdiv(M, 256), 256, 0, st>>>(M, coords, D2, flag);
  exscan_rows(st, 1, (int)cells, flag, bsum);                 // flag[cells-1] = number of coarse voxels
  sc_coarse_rules_kernel<<<cdiv(M, 256), 256, 0, st>>>(M, coords, D2, flag, coords2, child, parent);
  GCN_HIP(hipMemcpyAsync(m2_dev, flag + cells - 1, sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return check_launch("sc_coarse_rules_kernel");
}

// offsets are split over `parts` workgroup groups when the launch would otherwise have fewer than ~512 workgroups
static int sc_parts(int Mout, int K, int Cout) {
  const long wgs = (long)cdiv(Mout, 256) * (Cout / 64);
  int parts = 1;
  while (parts * 3 <= K && K % (parts * 3) == 0 && wgs * parts < 512) parts *= 3;      // 27 -> 1, 3, 9
  while (parts * 2 <= K && K % (parts * 2) == 0 && wgs * parts < 512) parts *= 2;      // 8 -> 1, 2, 4
  return parts;
}

GCN_EXPORT long gcn_sparse_gather_gemm_ws_floats(int Mout, int K, int Cout) {
  if (Mout < 0 || K < 1 || Cout < 64) return -1;
  const int parts = sc_parts(Mout, K, Cout);
  return parts > 1 ? (long)parts * Mout * Cout : 0;
}

GCN_EXPORT int gcn_sparse_gather_gemm(int Mout, int K, int Cin, int Cout, const float *in, const int32_t *rule, const float *W,
                                      int w_transposed, int k_reversed, float *out, float *ws, void *stream) {
  GCN_REQUIRE(Mout >= 0 && K >= 1 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0,
              "gcn_sparse_gather_gemm: channels must be multiples of 64 (Cin=%d, Cout=%d)", Cin, Cout);
  if (Mout == 0) return GCN_OK;
  GCN_REQUIRE(in && rule && W && out, "gcn_sparse_gather_gemm: null pointer");
  GCN_REQUIRE(!w_transposed, "gcn_sparse_gather_gemm: pass the weight as (K, Cin, Cout) of THIS product (transpose on the caller's side)");
  hipStream_t st = (hipStream_t)stream;
  const int parts = sc_parts(Mout, K, Cout);
  GCN_REQUIRE(parts == 1 || ws, "gcn_sparse_gather_gemm: this shape needs gcn_sparse_gather_gemm_ws_floats() floats of scratch");
  float *dst = parts > 1 ? ws : out;
  const dim3 grid(cdiv(Mout, 256), Cout / 64, parts);
  if (k_reversed) sc_gather_gemm_kernel<true><<<grid, 256, 0, st>>>(Mout, K, Cin, Cout, in, rule, W, dst);
  else sc_gather_gemm_kernel<false><<<grid, 256, 0, st>>>(Mout, K, Cin, Cout, in, rule, W, dst);
  if (parts > 1) {
    const long n4 = (long)Mout * Cout / 4;
    sc_sum_parts_kernel<<<cdiv(n4, 256), 256, 0, st>>>(n4, parts, reinterpret_cast<const float4 *>(ws), reinterpret_cast<float4 *>(out));
  }
  return check_launch("sc_gather_gemm_kernel");
}

GCN_EXPORT int gcn_sparse_wgrad(int Mout, int K, int Cin, int Cout, const float *in, const int32_t *ruleT, const float *dout,
                                float *dW, void *stream) {
  GCN_REQUIRE(Mout >= 0 && K >= 1 && Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % 64 == 0,
              "gcn_sparse_wgrad: channels must be multiples of 64 (Cin=%d, Cout=%d)", Cin, Cout);
  GCN_REQUIRE(dW, "gcn_sparse_wgrad: dW is null");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(fill_dev(dW, 0, sizeof(float) * (size_t)K * Cin * Cout, st));
  if (Mout == 0) return GCN_OK;
  GCN_REQUIRE(in && ruleT && dout, "gcn_sparse_wgrad: null pointer");
  int rows = 1024;                                   // >= ~1000 workgroups; at most 4096 rows each
  while (rows < 4096 && (long)cdiv(Mout, rows) * K * (Cin / 64) * (Cout / 64) > 2048) rows *= 2;
  sc_wgrad_kernel<<<dim3(cdiv(Mout, rows), K, (Cin / 64) * (Cout / 64)), 256, 0, st>>>(Mout, K, Cin, Cout, rows, in, ruleT, dout, dW);
  return check_launch("sc_wgrad_kernel");
}

// ---------------------------------------------------------------- BatchNorm1d (+ReLU) over the voxel rows
// blocks.py's norm_fn = BatchNorm1d(eps 1e-4, momentum 0.1) is always followed by ReLU; in training mode the statistics
// run over the M active voxels.  Two launches each way instead of torch's five (statistics are f32 per thread, f64 across
// threads and workgroups); the running statistics are updated by the apply kernel's first workgroup.
namespace gcn {

// sums (2C doubles, zero on entry): [sum x, sum x^2] per channel.  Thread = one float4 channel group of a row slab.
__global__ __launch_bounds__(256) void bn_stats_kernel(int M, int C, int rows_per_block, const float *__restrict__ x,
                                                       double *__restrict__ sums) {
  __shared__ float part[256][8];                       // per-thread partial sums; reduced over the row slots below
  const int c4n = C / 4, rstep = 256 / c4n;            // C/4 divides 256 (C = 64: 16 rows per pass, C = 128: 8)
  const int c4 = threadIdx.x % c4n;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + threadIdx.x / c4n; r < r1; r += rstep) {
    const float4 v = *reinterpret_cast<const float4 *>(x + (long)r * C + 4 * c4);
    s1[0] += v.x; s1[1] += v.y; s1[2] += v.z; s1[3] += v.w;
    s2[0] = fmaf(v.x, v.x, s2[0]); s2[1] = fmaf(v.y, v.y, s2[1]); s2[2] = fmaf(v.z, v.z, s2[2]); s2[3] = fmaf(v.w, v.w, s2[3]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { part[threadIdx.x][e] = s1[e]; part[threadIdx.x][4 + e] = s2[e]; }
  __syncthreads();
  // thread -> (which sum, channel): add the rstep row slots in double, one global atomic per channel and workgroup
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    double a = 0.0;
    for (int sl = 0; sl < rstep; ++sl) a += (double)part[sl * c4n + c / 4][4 * which + (c & 3)];
    atomicAdd(sums + i, a);
  }
}

// y = [relu]((x - mean) * rstd * gamma + beta); mean_rstd (C,2) saved for backward; running statistics as nn.BatchNorm1d
__global__ __launch_bounds__(256) void bn_apply_kernel(int M, int C, const float *__restrict__ x, const double *__restrict__ sums,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                       int relu, float momentum, float *__restrict__ y,
                                                       float *__restrict__ mean_rstd, float *__restrict__ running_mean,
                                                       float *__restrict__ running_var) {
  __shared__ float s_scale[1024], s_shift[1024];          // y = x * scale + shift per channel (C <= 1024)
  for (int c = threadIdx.x; c < C; c += 256) {
    const double m = sums[c] / M;
    const double var = fmax(sums[C + c] / M - m * m, 0.0);
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    s_scale[c] = rstd * gamma[c];
    s_shift[c] = beta[c] - (float)m * rstd * gamma[c];
  }
  __syncthreads();
  const long per = (long)M * C / 4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
    const int c = (int)((e * 4) % C);
    const float4 v = reinterpret_cast<const float4 *>(x)[e];
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z = fmaf(o[i], s_scale[c + i], s_shift[c + i]);
      o[i] = relu ? fmaxf(z, 0.f) : z;
    }
    reinterpret_cast<float4 *>(y)[e] = float4{o[0], o[1], o[2], o[3]};
  }
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const double m = sums[c] / M;
      const double var = fmax(sums[C + c] / M - m * m, 0.0);
      mean_rstd[2 * c] = (float)m;
      mean_rstd[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        const double unb = M > 1 ? var * M / (M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
}

// acc (2C doubles, zero on entry): [sum g, sum g*xhat] with g = dy * [z > 0]
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(int M, int C, int rows_per_block, const float *__restrict__ dy,
                                                            const float *__restrict__ x, const float *__restrict__ mean_rstd,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            int relu, double *__restrict__ acc) {
  __shared__ float part[256][8];
  const int c4n = C / 4, rstep = 256 / c4n;
  const int c4 = threadIdx.x % c4n;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, M);
  float mean[4], rstd[4], ga[4], be[4], s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    mean[e] = mean_rstd[2 * (4 * c4 + e)]; rstd[e] = mean_rstd[2 * (4 * c4 + e) + 1];
    ga[e] = gamma[4 * c4 + e]; be[e] = beta[4 * c4 + e];
  }
  for (int r = r0 + threadIdx.x / c4n; r < r1; r += rstep) {
    const float4 xv = *reinterpret_cast<const float4 *>(x + (long)r * C + 4 * c4);
    const float4 gv = *reinterpret_cast<const float4 *>(dy + (long)r * C + 4 * c4);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xs[e] - mean[e]) * rstd[e];
      const float z = fmaf(xs[e], rstd[e] * ga[e], be[e] - mean[e] * rstd[e] * ga[e]);     // as bn_apply_kernel forms it
      const float g = (relu && !(z > 0.f)) ? 0.f : gs[e];
      s1[e] += g;
      s2[e] = fmaf(g, xh, s2[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { part[threadIdx.x][e] = s1[e]; part[threadIdx.x][4 + e] = s2[e]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i % C;
    double a = 0.0;
    for (int sl = 0; sl < rstep; ++sl) a += (double)part[sl * c4n + c / 4][4 * which + (c & 3)];
    atomicAdd(acc + i, a);
  }
}

// dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)); dgamma = sum g*xhat, dbeta = sum g (first workgroup)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(int M, int C, const float *__restrict__ dy, const float *__restrict__ x,
                                                           const float *__restrict__ mean_rstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int relu, const double *__restrict__ acc,
                                                           float *__restrict__ dx, float *__restrict__ dgamma,
                                                           float *__restrict__ dbeta) {
  __shared__ float s_mean[1024], s_rstd[1024], s_sc[1024], s_sh[1024], s_mg[1024], s_mgx[1024];
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mean = mean_rstd[2 * c], rstd = mean_rstd[2 * c + 1], ga = gamma[c];
    s_mean[c] = mean; s_rstd[c] = rstd;
    s_sc[c] = rstd * ga; s_sh[c] = beta[c] - mean * rstd * ga;
    s_mg[c] = (float)(acc[c] / M); s_mgx[c] = (float)(acc[C + c] / M);
  }
  __syncthreads();
  const long per = (long)M * C / 4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
    const int c = (int)((e * 4) % C);
    const float4 xv = reinterpret_cast<const float4 *>(x)[e], gv = reinterpret_cast<const float4 *>(dy)[e];
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float xh = (xs[i] - s_mean[c + i]) * s_rstd[c + i];
      const float g = (relu && !(fmaf(xs[i], s_sc[c + i], s_sh[c + i]) > 0.f)) ? 0.f : gs[i];
      o[i] = s_sc[c + i] * (g - s_mg[c + i] - xh * s_mgx[c + i]);
    }
    reinterpret_cast<float4 *>(dx)[e] = float4{o[0], o[1], o[2], o[3]};
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] = (float)acc[c]; dgamma[c] = (float)acc[C + c]; }
}

}  // namespace gcn

static int bn_check(int M, int C, const char *who) {
  GCN_REQUIRE(M >= 1 && C >= 4 && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && C <= 1024, "%s: M=%d, C=%d unsupported (C/4 must divide 256)", who, M, C);
  return GCN_OK;
}

GCN_EXPORT int gcn_bn_relu_fwd(int M, int C, const float *x, const float *gamma, const float *beta, float eps, int relu,
                               float momentum, float *y, float *mean_rstd, float *running_mean, float *running_var,
                               double *sums_ws, void *stream) {
  int rc = bn_check(M, C, "gcn_bn_relu_fwd");
  if (rc) return rc;
  GCN_REQUIRE(x && gamma && beta && y && mean_rstd && sums_ws, "gcn_bn_relu_fwd: null pointer");
  GCN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "gcn_bn_relu_fwd: pass both running buffers or neither");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(sums_ws, sizeof(double) * 2 * C, st));          // skipped inside the pre-zeroed arena
  const int rows = 256;
  bn_stats_kernel<<<cdiv(M, rows), 256, 0, st>>>(M, C, rows, x, sums_ws);
  const int blocks = (int)fmin(4096.0, (double)cdiv((long)M * C / 4, 256));
  bn_apply_kernel<<<blocks, 256, 0, st>>>(M, C, x, sums_ws, gamma, beta, eps, relu, momentum, y, mean_rstd, running_mean, running_var);
  return check_launch("bn_apply_kernel");
}

GCN_EXPORT int gcn_bn_relu_bwd(int M, int C, const float *dy, const float *x, const float *gamma, const float *beta,
                               const float *mean_rstd, int relu, float *dx, float *dgamma, float *dbeta, double *acc_ws,
                               void *stream) {
  int rc = bn_check(M, C, "gcn_bn_relu_bwd");
  if (rc) return rc;
  GCN_REQUIRE(dy && x && gamma && beta && mean_rstd && dx && dgamma && dbeta && acc_ws, "gcn_bn_relu_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(acc_ws, sizeof(double) * 2 * C, st));
  const int rows = 256;
  bn_bwd_reduce_kernel<<<cdiv(M, rows), 256, 0, st>>>(M, C, rows, dy, x, mean_rstd, gamma, beta, relu, acc_ws);
  const int blocks = (int)fmin(4096.0, (double)cdiv((long)M * C / 4, 256));
  bn_bwd_apply_kernel<<<blocks, 256, 0, st>>>(M, C, dy, x, mean_rstd, gamma, beta, relu, acc_ws, dx, dgamma, dbeta);
  return check_launch("bn_bwd_apply_kernel");
}
