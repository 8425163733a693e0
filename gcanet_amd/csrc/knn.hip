// knn.hip -- fused brute-force kNN for gfx950 (CDNA4, wave64).
//
// Replaces KNN_CUDA's three kernels (models/KNN_CUDA/knn_cuda/csrc/cuda/knn.cu:29-183:
// materialised (nr,nq) distance matrix -> per-thread global-memory insertion sort ->
// sqrt) and the pure-torch N x N `knn` / `knn_points_normals` + topk of
// models/dgcnn-hais-concat-direct-4.py:30-90 with ONE kernel that never writes the
// distance matrix.
//
// Design (wave64-first, not a warp tiling):
//   * one wave owns QW queries; its 64 lanes each take one candidate of a 64-wide batch,
//     so a batch's candidate coordinates are one coalesced 256-B load per dimension and
//     are reused for all QW queries from registers;
//   * query coordinates are wave-uniform -> scalar loads, SGPR operands of v_fma;
//   * the running top-k of a query is a SORTED list held one entry per lane (k <= 64) or
//     KPL entries per lane (k <= 64*KPL): insertion = ballot + popcount for the position,
//     one DPP wave_shr:1 to open the slot.  The k-th key is a scalar threshold, so the
//     steady state per (query, batch) is: distance, one v_cmp against an SGPR, one scalar
//     branch.
//   * candidates are visited in ascending index and inserted behind equal keys, so ties
//     resolve to the LOWEST index exactly like the reference's stable insertion sort
//     (knn.cu:125-131) -- bit-exact indices vs oracle/gcanet_oracle.c.
// Arithmetic follows the oracle's contraction convention (explicit fmaf chains; the file
// is compiled with -ffp-contract=off).
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"
#include "knn_topb.h"

namespace gcn {

#define KNN_INF __builtin_inff()

template <int KPL>
struct TopK {
  float key[KPL];
  int idx[KPL];

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      key[s] = KNN_INF;
      idx[s] = 0x7fffffff;
    }
  }
  // key at sorted position kslot*64 + klane (wave-uniform result)
  __device__ __forceinline__ float kth(int kslot, int klane) const {
    float t = KNN_INF;
#pragma unroll
    for (int s = 0; s < KPL; ++s)
      if (s == kslot) t = readlane_f(key[s], klane);
    return t;
  }
  // insert (ckey, cidx) behind all entries with key <= ckey; the last entry falls off
  __device__ __forceinline__ void insert(float ckey, int cidx, int lane) {
    if (KPL == 1) {
      // all-VALU form: entries <= ckey stay; the rest take max(left neighbour, ckey) -- no
      // scalar round trip (ballot/popcount) on the critical path
      // (the list is sorted, so "left neighbour > ckey" is the keep mask shifted by one lane: one
      // compare + two scalar ops instead of a second compare on the DPP-shifted key)
      // Seven VALU per insert: the DPP lane shift is folded into the select (v_cndmask_b32_dpp), which
      // hipcc does not form by itself.  vcc = lanes whose left neighbour is kept (or lane 0): they take
      // the candidate, the others the shifted entry; lanes with key <= ckey keep their own.
      unsigned long long keep;
      float tk;
      int ti;
      asm volatile(
          "v_cmp_ge_f32_e64 %[keep], %[ck], %[key]\n\t"
          "s_lshl_b64 vcc, %[keep], 1\n\t"
          "s_or_b32 vcc_lo, vcc_lo, 1\n\t"
          "v_mov_b32_e32 %[tk], %[ck]\n\t"
          "v_mov_b32_e32 %[ti], %[ci]\n\t"
          "v_cndmask_b32_dpp %[tk], %[key], %[tk], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_dpp %[ti], %[idx], %[ti], vcc wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
          "v_cndmask_b32_e64 %[key], %[tk], %[key], %[keep]\n\t"
          "v_cndmask_b32_e64 %[idx], %[ti], %[idx], %[keep]\n\t"
          : [key] "+v"(key[0]), [idx] "+v"(idx[0]), [keep] "=&s"(keep), [tk] "=&v"(tk), [ti] "=&v"(ti)
          : [ck] "s"(ckey), [ci] "s"(cidx)
          : "vcc", "scc");
      return;
    }
    int pos = 0;
#pragma unroll
    for (int s = 0; s < KPL; ++s) pos += __popcll(__ballot(key[s] <= ckey));
#pragma unroll
    for (int s = KPL - 1; s >= 0; --s) {
      const int lo = s * 64;
      if (pos < lo + 64) {
        float ck = 0.f;
        int ci = 0;
        if (s > 0) {
          ck = readlane_f(key[s - 1], 63);
          ci = readlane_i(idx[s - 1], 63);
        }
        const float sk = wave_shr1_f(ck, key[s]);
        const int si = wave_shr1_i(ci, idx[s]);
        const int lp = pos - lo;
        key[s] = lane < lp ? key[s] : (lane == lp ? ckey : sk);
        idx[s] = lane < lp ? idx[s] : (lane == lp ? cidx : si);
      }
    }
  }
};

// ------------------------------------------------------------------------------------------
// Buffered bitonic top-64 (k <= 64).  Serial list insertion costs ~25 instructions per passing
// candidate and there are ~k(1+ln(N/k)) of them per query; here passing candidates of a 64-wide batch
// are appended to a PENDING register (one entry per lane) by ONE wave-wide ds_permute compaction,
// and only when the pending register would overflow is it bitonic-sorted and merged into the sorted
// list.  The threshold (k-th key) is refreshed at merges only; the scanned fraction then doubles
// between merges, i.e. ~log2(N/k)+1 merges per query instead of hundreds of insertions.
// Entries are (monotone key bits << 32 | index) so u64 order == (key asc, index asc): ties resolve
// to the lowest index exactly as the stable insertion sort of the reference (knn.cu:125-131).
struct KnnArgs {
  const float *ref;    // candidates
  const float *query;
  const float *xx;     // (B, nr) squared norms for the model metrics (ref == query there)
  long ref_sb, ref_sd, ref_sn;  // element (b,d,j) at b*sb + d*sd + j*sn
  long q_sb, q_sd, q_sn;
  int dim, nr, nq, k, step;
  float *dist;   // may be null
  int64_t *ind;
  long o_sb, o_sk, o_sq;  // output element (b,t,q) at b*sb + t*sk + q*sq
  const unsigned char *only;  // (B, nq) or null: when given, only queries with a non-zero byte are searched and written
};

// METRIC 0: KNN_CUDA direct sum of squared differences (knn.cu:73-77), out dist = sqrt
// METRIC 1: in-model expanded form (M4:36-38), key = -pairwise_distance
// METRIC 2: knn_points_normals (M4:62-75), key = p_pd*(1+n_pd)
template <int KPL, int QW, int METRIC, int DIMC>
__global__ __launch_bounds__(256) void knn_select_kernel(KnnArgs a) {
  const int lane = lane_id();
  const int wave = wave_id();
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * QW;
  if (q0 >= a.nq) return;  // wave-uniform
  const int dim = DIMC > 0 ? DIMC : a.dim;

  const float *__restrict__ ref = a.ref + (long)b * a.ref_sb;
  const float *__restrict__ qry = a.query + (long)b * a.q_sb;
  const float *__restrict__ xx = METRIC == 0 ? nullptr : a.xx + (long)b * a.nr;

  int qi[QW];  // clamped query ids (uniform)
#pragma unroll
  for (int q = 0; q < QW; ++q) qi[q] = min(q0 + q, a.nq - 1);
  if (a.only) {  // flagged-only mode: nearly every wave leaves here
    bool any = false;
#pragma unroll
    for (int q = 0; q < QW; ++q) any |= a.only[(long)b * a.nq + qi[q]] != 0;
    if (!any) return;
  }

  TopK<KPL> top[QW];
  TopB topb[QW];     // k <= 64: buffered bitonic selection (KPL == 1)
  int cnt[QW];
  float thr[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    top[q].init();
    topb[q].init();
    cnt[q] = 0;
    thr[q] = KNN_INF;
  }
  const int kslot = (a.k - 1) >> 6, klane = (a.k - 1) & 63;

  float xxi[QW];
  if (METRIC != 0) {
#pragma unroll
    for (int q = 0; q < QW; ++q) xxi[q] = xx[qi[q]];
  }

  for (int base = 0; base < a.nr; base += 64) {
    const int j = base + lane;
    const bool valid = j < a.nr;
    const int jc = valid ? j : a.nr - 1;
    float key[QW];
    if (METRIC == 0) {
      float acc[QW];
#pragma unroll
      for (int q = 0; q < QW; ++q) acc[q] = 0.f;
#pragma unroll 4
      for (int d = 0; d < dim; ++d) {
        const float cv = ref[d * a.ref_sd + jc * a.ref_sn];
#pragma unroll
        for (int q = 0; q < QW; ++q) {
          const float t = cv - qry[d * a.q_sd + qi[q] * a.q_sn];
          acc[q] = fmaf(t, t, acc[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < QW; ++q) key[q] = valid ? acc[q] : KNN_INF;
    } else if (METRIC == 1) {
      float acc[QW];
#pragma unroll
      for (int q = 0; q < QW; ++q) acc[q] = 0.f;
#pragma unroll 4
      for (int d = 0; d < dim; ++d) {
        const float cv = ref[d * a.ref_sd + jc * a.ref_sn];
#pragma unroll
        for (int q = 0; q < QW; ++q) acc[q] = fmaf(qry[d * a.q_sd + qi[q] * a.q_sn], cv, acc[q]);
      }
      const float xxj = xx[jc];
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        const float t = 2.f * acc[q] - xxj;
        const float pd = t - xxi[q];
        key[q] = valid ? -pd : KNN_INF;
      }
    } else {
      float cv[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) cv[d] = ref[d * a.ref_sd + jc * a.ref_sn];
      const float xxj = xx[jc];
#pragma unroll
      for (int q = 0; q < QW; ++q) {
        float dp = 0.f, dn = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) dp = fmaf(qry[d * a.q_sd + qi[q] * a.q_sn], cv[d], dp);
#pragma unroll
        for (int d = 3; d < 6; ++d) dn = fmaf(qry[d * a.q_sd + qi[q] * a.q_sn], cv[d], dn);
        const float p_pd = (xxj - 2.f * dp) + xxi[q];
        const float n_pd = 2.f - 2.f * dn;
        const float pd = p_pd * (1.f + n_pd);
        key[q] = valid ? pd : KNN_INF;
      }
    }

#pragma unroll
    for (int q = 0; q < QW; ++q) {
      if (KPL == 1) {
        const bool pass = key[q] < thr[q];
        const unsigned long long m = __ballot(pass);
        if (m) {
          if (cnt[q] + __popcll(m) > 64) {
            topb[q].merge(cnt[q], lane);
            cnt[q] = 0;
            thr[q] = key_u2f((unsigned int)(__builtin_amdgcn_readlane((int)(unsigned int)(topb[q].lst >> 32), klane)));
          }
          cnt[q] = topb[q].append(m, pass, key[q], base + lane, cnt[q], lane);
        }
        continue;
      }
      unsigned long long m = __ballot(key[q] < thr[q]);
      while (m) {
        const int l = __ffsll((long long)m) - 1;
        m &= m - 1;
        const float ck = readlane_f(key[q], l);
        if (ck < thr[q]) {
          top[q].insert(ck, base + l, lane);
          thr[q] = top[q].kth(kslot, klane);
        }
      }
    }
  }

  // epilogue: sorted ascending by (key, index); position t = s*64 + lane
  if (KPL == 1) {
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      topb[q].merge(cnt[q], lane);
      top[q].key[0] = key_u2f((unsigned int)(topb[q].lst >> 32));
      top[q].idx[0] = (int)(unsigned int)topb[q].lst;
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    if (q0 + q >= a.nq) break;
    if (a.only && a.only[(long)b * a.nq + q0 + q] == 0) continue;
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const int t = s * 64 + lane;
      if (t < a.k && (t % a.step) == 0) {
        const long o = (long)b * a.o_sb + (long)(t / a.step) * a.o_sk + (long)(q0 + q) * a.o_sq;
        a.ind[o] = (int64_t)top[q].idx[s];
        if (a.dist) {
          const float kv = top[q].key[s];
          a.dist[o] = METRIC == 0 ? sqrtf(kv) : (METRIC == 1 ? -kv : -kv);
        }
      }
    }
  }
}

// squared norms in the oracle's order: squares rounded, added left to right (x**2 then sum)
__global__ void sqnorm_kernel(const float *__restrict__ x, float *__restrict__ xx, int C, int N, int cx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= N) return;
  const float *p = x + (long)b * C * N + j;
  float s = 0.f;
  for (int c = 0; c < cx; ++c) {
    const float v = p[(long)c * N];
    const float sq = v * v;
    s = c == 0 ? sq : s + sq;
  }
  xx[(long)b * N + j] = s;
}

template <int METRIC, int DIMC>
static int launch_knn(const KnnArgs &a, int B, hipStream_t st) {
  const int k = a.k;
  if (k <= 64) {
    dim3 grid(cdiv(a.nq, 4 * 8), B);
    knn_select_kernel<1, 8, METRIC, DIMC><<<grid, 256, 0, st>>>(a);
  } else if (k <= 128) {
    dim3 grid(cdiv(a.nq, 4 * 8), B);
    knn_select_kernel<2, 8, METRIC, DIMC><<<grid, 256, 0, st>>>(a);
  } else if (k <= 256) {
    dim3 grid(cdiv(a.nq, 4 * 4), B);
    knn_select_kernel<4, 4, METRIC, DIMC><<<grid, 256, 0, st>>>(a);
  } else {
    dim3 grid(cdiv(a.nq, 4 * 2), B);
    knn_select_kernel<8, 2, METRIC, DIMC><<<grid, 256, 0, st>>>(a);
  }
  return check_launch("knn_select_kernel");
}


// ------------------------------------------------------------------------------------------
// Feature-space kNN (C in {32,64,128}) on the matrix cores.  v_mfma_f32_16x16x4_f32 is bit-for-bit
// a k-ordered fmaf chain (cdna_hip_programming.md section 3), so dot(x_i, x_j) accumulated over
// ascending channels equals the oracle's scalar chain exactly and indices stay bit-exact, while
// the VALU is left to the top-k bookkeeping.
//   workgroup = 4 waves, each wave owns 16 queries (one MFMA row block); the query fragments stay
//   in registers (CC/4 VGPRs).  Candidates stream through LDS in tiles of TC ([channel][cand] f32,
//   the global layout, fetched by LDS-DMA, double buffered) and are shared by the 4 waves.
//   C/D layout: lane = candidate (l&15), register r = query row 4(l>>4)+r: every accumulator
//   register carries FOUR queries (one per 16-lane group), each with its own 64-entry sorted list
//   spread across all 64 lanes -> 16 lists per wave, 32 VGPRs, four waves per SIMD.  (A 32x32x2
//   version with 32 queries per wave ran two waves per SIMD and was 6 % slower: the kernel is bound
//   by the dependent insert chains -- PMC: VALU 42 %, MFMA 31 % of SIMD time.)
// k <= 64: one list entry per lane (KPL = 1, 7-VALU insert); 64 < k <= 128: two (KPL = 2, generic insert);
// larger k uses knn_select_kernel.
//   A[i=l&15][kk=l>>4] = x[4s+kk][q0+i];  B[kk=l>>4][j=l&15] = x[4s+kk][cand j]: the four channel rows a
//   B fetch touches are one LDS row apart (TC=64: same banks), so the DMA writes row r with its
//   16-candidate blocks XOR-permuted by (r&3) and the fetch undoes it (source-side swizzle).
typedef __attribute__((ext_vector_type(4))) float knn_f32x4;

// FLAGGED (the exhaustive stage of knn_filter.hip when its list is long): `only` (B,N) marks the queries to search and
// write; a wave none of whose 16 queries is marked only keeps the tile stream going, a workgroup without marked queries
// leaves at once, and the whole launch is a no-op unless *gate > gate_min.
template <int CC, int TC, int KPL, bool FLAGGED = false>   // KPL list registers per query: k <= 64 * KPL
__global__ __launch_bounds__(KPL == 1 ? 512 : 256, KPL == 1 ? 4 : 3) void knn_mfma16_kernel(const float *__restrict__ x, const float *__restrict__ xxg,
                                                            int N, int k, int step, int kout,
                                                            int64_t *__restrict__ ind, float *__restrict__ val,
                                                            const unsigned char *__restrict__ only = nullptr,
                                                            const unsigned int *__restrict__ gate = nullptr,
                                                            unsigned int gate_min = 0) {
  constexpr int ROWS = CC + 1;           // + one row of squared norms
  constexpr int RPP = 256 / TC;          // rows per 1-KiB DMA piece
  constexpr int PIECES = (ROWS + RPP - 1) / RPP;
  constexpr int CPR = TC / 4;            // 16-B chunks per row
  constexpr int NCB = TC / 16;           // 16-candidate column blocks per tile
  extern __shared__ __attribute__((aligned(1024))) float tile[];  // 2 buffers of PIECES KiB

  const int lane = lane_id(), wave = wave_id();
  const int lc = lane & 15, lg = lane >> 4;
  // Workgroups go to the 8 XCDs round robin by their linear id; cloud = id % B keeps a cloud's candidate rows (2 MB at
  // C = 64) in ONE XCD's L2 when B is a multiple of 8 instead of streaming every cloud through all eight.
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  const int b = lin % (int)gridDim.y;
  const int NW = (int)(blockDim.x >> 6);             // waves (16 queries each) sharing a candidate tile
  const int q0 = ((lin / (int)gridDim.y) * NW + wave) * 16;
  const float *xb = x + (long)b * CC * N;
  const float *xxb = xxg + (long)b * N;

  bool act = true;                                   // wave-uniform: this wave has queries to search
  if (FLAGGED) {
    if (*gate <= gate_min) return;
    const bool mine = lane < 16 && q0 + lane < N && only[(long)b * N + q0 + lane] != 0;
    act = __ballot(mine) != 0ull;
    if (!__syncthreads_or(act ? 1 : 0)) return;
  }

  const int qa = min(q0 + lc, N - 1);
  float afrag[CC / 4];
#pragma unroll
  for (int s = 0; s < CC / 4; ++s) afrag[s] = xb[(long)(4 * s + lg) * N + qa];
  float xxq[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xxq[r] = xxb[min(q0 + 4 * lg + r, N - 1)];

  TopK<KPL> top[4][4];                   // [r][g]: query q0 + 4g + r
  float thrv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int g = 0; g < 4; ++g) top[r][g].init();
    thrv[r] = KNN_INF;
  }
  const int kslot = (k - 1) >> 6, klane = (k - 1) & 63;

  const int ntiles = (N + TC - 1) / TC;
  auto issue_tile = [&](int t, int buf) {
    int j0 = t * TC;
    if (j0 + TC > N) j0 = N - TC;                   // tail tile: shifted back (N >= TC, N % 4 == 0 by dispatch)
    for (int p = wave; p < PIECES; p += NW) {
      const int row = p * RPP + lane / CPR;
      const int chunk = lane % CPR;
      const int sc = TC == 64 ? (chunk ^ ((row & 3) << 2)) : chunk;
      const int j = j0 + sc * 4;
      const float *src = row < CC ? xb + (long)row * N + j : xxb + j;
      if (row < ROWS)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(tile + buf * PIECES * 256 + p * 256), 16, 0, 0);
    }
  };

  issue_tile(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue_tile(t + 1, buf ^ 1);
    const float *tb = tile + buf * PIECES * 256;
    const int j0 = t * TC;
    const int jbase = (j0 + TC > N) ? N - TC : j0;
    // CC <= 64: the B operands of a column block are fetched as ONE batch of CC/4 LDS reads ahead of its MFMA chain
    // (block cb+1 while block cb computes) instead of a read -> wait -> MFMA pair per step with the LDS latency on the
    // dependent chain (-2 %); at CC = 128 the 2 x 32 extra registers cost the fourth wave per SIMD (2.1 -> 3.6 ms)
    constexpr bool PREFETCH_B = CC <= 64;
    float bv[PREFETCH_B ? CC / 4 : 1];
    if (!FLAGGED || act) {
    if (PREFETCH_B) {
      const float *bcol0 = tb + lg * TC + (TC == 64 ? (((0 ^ lg) << 4) + lc) : lc);
#pragma unroll
      for (int s = 0; s < CC / 4; ++s) bv[PREFETCH_B ? s : 0] = bcol0[4 * s * TC];
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      knn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (PREFETCH_B) {
        float bcur[PREFETCH_B ? CC / 4 : 1];
#pragma unroll
        for (int s = 0; s < CC / 4; ++s) bcur[PREFETCH_B ? s : 0] = bv[PREFETCH_B ? s : 0];
        if (cb + 1 < NCB) {
          const float *bcol = tb + lg * TC + (TC == 64 ? ((((cb + 1) ^ lg) << 4) + lc) : ((cb + 1) * 16 + lc));
#pragma unroll
          for (int s = 0; s < CC / 4; ++s) bv[PREFETCH_B ? s : 0] = bcol[4 * s * TC];
        }
#pragma unroll
        for (int s = 0; s < CC / 4; ++s)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], bcur[PREFETCH_B ? s : 0], acc, 0, 0, 0);
      } else {
        const float *bcol = tb + lg * TC + (TC == 64 ? (((cb ^ lg) << 4) + lc) : (cb * 16 + lc));
#pragma unroll
        for (int s = 0; s < CC / 4; ++s)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[s], bcol[4 * s * TC], acc, 0, 0, 0);
      }
      const int j = jbase + cb * 16 + lc;
      const float xxj = tb[CC * TC + cb * 16 + lc];
      const bool fresh = j >= j0 && j < N;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tt = 2.f * acc[r] - xxj;
        const float pd = tt - xxq[r];
        const float key = fresh ? -pd : KNN_INF;
        const unsigned long long m = __ballot(key < thrv[r]);
        if (m) {
          const int cb0 = jbase + cb * 16;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            unsigned int mg = (unsigned int)(m >> (16 * g)) & 0xffffu;
            while (mg) {
              const int l = __ffs((int)mg) - 1;
              mg &= mg - 1;
              top[r][g].insert(readlane_f(key, l + 16 * g), cb0 + l, lane);
            }
          }
          float t0, t1, t2, t3;
          if (KPL == 1) {
            t0 = readlane_f(top[r][0].key[0], klane); t1 = readlane_f(top[r][1].key[0], klane);
            t2 = readlane_f(top[r][2].key[0], klane); t3 = readlane_f(top[r][3].key[0], klane);
          } else {
            t0 = top[r][0].kth(kslot, klane); t1 = top[r][1].kth(kslot, klane);
            t2 = top[r][2].kth(kslot, klane); t3 = top[r][3].kth(kslot, klane);
          }
          thrv[r] = lg == 0 ? t0 : (lg == 1 ? t1 : (lg == 2 ? t2 : t3));
        }
      }
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (FLAGGED && !act) return;

#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int q = q0 + 4 * g + r;
#pragma unroll
      for (int sl = 0; sl < KPL; ++sl) {
        const int t = sl * 64 + lane;
        if (q < N && t < k && (t % step) == 0 && (!FLAGGED || only[(long)b * N + q] != 0)) {
          const long o = ((long)b * N + q) * kout + t / step;
          ind[o] = (int64_t)top[r][g].idx[sl];
          if (val) val[o] = -top[r][g].key[sl];
        }
      }
    }
}

template <int CC, int TC>
static int launch_knn_mfma16(const float *x, const float *xx, int B, int N, int k, int step, int kout, int64_t *ind,
                             float *val, hipStream_t st) {
  constexpr int PIECES = (CC + 1 + 256 / TC - 1) / (256 / TC);
  const int lds = 2 * PIECES * 1024;
  if (k <= 64) {
    GCN_HIP(hipFuncSetAttribute((const void *)knn_mfma16_kernel<CC, TC, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    knn_mfma16_kernel<CC, TC, 1><<<dim3(cdiv(N, 128), B), 512, lds, st>>>(x, xx, N, k, step, kout, ind, val);
  } else {   // 64 < k <= 128 (the reference's default k = 80): two list registers per query
    GCN_HIP(hipFuncSetAttribute((const void *)knn_mfma16_kernel<CC, TC, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    knn_mfma16_kernel<CC, TC, 2><<<dim3(cdiv(N, 64), B), 256, lds, st>>>(x, xx, N, k, step, kout, ind, val);
  }
  return check_launch("knn_mfma16_kernel");
}


// ------------------------------------------------------------------------------------------
// Row-wise top-k (largest, sorted descending) of short rows (NK <= 128, k <= 64): the `torch.topk(dist, 30)` over
// the 120 key-point similarities of every point in OFFSET_PRED_MODULE (M4:421-422; torch's radix-select kernel
// takes 0.42 ms at B*N = 65536).  One wave per row: lane l holds columns l and l+64 as u64 keys (order-preserving
// value bits << 32 | ~column, so ties go to the LOWER column), two 64-wide bitonic sorts + one bitonic merge.
template <bool BF16>
__global__ __launch_bounds__(256) void topk_rows_kernel(const void *__restrict__ x, long R, int NK, int k,
                                                        float *__restrict__ vals, int64_t *__restrict__ idx) {
  const int lane = lane_id();
  const long row = (long)blockIdx.x * 4 + wave_id();
  if (row >= R) return;
  auto load = [&](int c) -> u64 {
    if (c >= NK) return 0ull;
    float v;
    if (BF16) v = __uint_as_float(((unsigned int)reinterpret_cast<const unsigned short *>(x)[row * NK + c]) << 16);
    else v = reinterpret_cast<const float *>(x)[row * NK + c];
    return ((u64)key_f2u(v) << 32) | (unsigned int)(0xFFFFFFFFu - (unsigned)c);
  };
  u64 a = load(lane), b = load(lane + 64);
  // descending sorts == ascending sorts of the complemented keys
  a = ~a; b = ~b;
#pragma unroll
  for (int sz = 2; sz <= 64; sz <<= 1)
#pragma unroll
    for (int j = sz >> 1; j >= 1; j >>= 1) {
      a = TopB::cex(a, lane, j, (lane & sz) == 0);
      b = TopB::cex(b, lane, j, (lane & sz) == 0);
    }
  const u64 br = shfl_u64(b, 63 - lane);
  u64 m = br < a ? br : a;                     // the 64 smallest complemented keys, bitonic
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) m = TopB::cex(m, lane, j, true);
  m = ~m;
  if (lane < k) {
    vals[row * k + lane] = key_u2f((unsigned int)(m >> 32));
    idx[row * k + lane] = (int64_t)(0xFFFFFFFFu - (unsigned int)m);
  }
}


// ------------------------------------------------------------------------------------------
// Self kNN of 3-D clouds (C = 3, or xyz+normal with the knn_points_normals metric) with spatial pruning.
// The brute-force kernel above evaluates all N candidates per query; in 3-D the k-th neighbour of a point lies
// within a few percent of the cloud's extent.  Here every cloud is sorted along a Morton curve, cut into tiles of
// 64 consecutive points with their bounding boxes, and a wave (8 queries, adjacent on the curve) walks the tiles
// outward from its own: a tile is skipped when, for each of the 8 queries, a conservative lower bound of the
// metric over the box exceeds that query's current k-th key.  Visited candidates get exactly the arithmetic of
// knn_select_kernel and entries are ordered by (key, ORIGINAL index) in the buffered bitonic lists, so indices
// and distances are bit-identical to the brute-force kernel -- only ~15-25 of the 128 tiles are evaluated at
// N = 8192, k = 64.  Lower bound: box distance^2 * (1 - 1e-6) - 2e-6 (xx_q + max xx_tile)  (the expanded-form
// distance can undershoot the true one by a few ulp of the squared norms), times the smallest possible normal
// factor 3 - 2 max|n|^2 for the normal metric (pruning is disabled when that is not positive).
struct TileKnnArgs {
  const float *xs;     // (B, C, N) sorted along the curve
  const float *xxs;    // (B, N)
  const int *perm;     // (B, N) original index of sorted position
  const float *bbox;   // (B, T, 8): lo xyz, hi xyz, max xx, pad
  int N, T, k, step;
  long o_sb, o_sk, o_sq;       // output element (b, t, q) at b*o_sb + t*o_sk + q*o_sq
  const unsigned int *box;   // (B, 8) per-cloud boxes; slot 6 = max |n|^2 bits of the cloud (METRIC 2)
  int64_t *ind;
  float *val;
};

__device__ __forceinline__ unsigned int morton_expand10(unsigned int v) {
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}
__device__ __forceinline__ unsigned int f2ord_u(float f) {
  const unsigned int u = __float_as_uint(f);
  return u ^ ((unsigned int)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ord2f_u(unsigned int u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

__global__ void tile_box_init_kernel(unsigned int *box, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) box[i] = (i & 7) < 3 ? 0xFFFFFFFFu : 0u;      // lo = +max (ordered), hi = lowest, |n|^2 = 0
}

// per-cloud bounding box of the xyz channels (ordered-uint atomics), and the cloud's max |n|^2 (slot 6)
__global__ void tile_bbox_kernel(const float *__restrict__ x, long sb, long sd, long sn, int N, int with_normals,
                                 unsigned int *__restrict__ box) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f}, nn = 0.f;
  const bool ok = j < N;
  if (ok) {
    for (int d = 0; d < 3; ++d) v[d] = x[b * sb + d * sd + j * sn];
    if (with_normals) for (int d = 3; d < 6; ++d) { const float t = x[b * sb + d * sd + j * sn]; nn = fmaf(t, t, nn); }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float lo = ok ? v[d] : __builtin_inff(), hi = ok ? v[d] : -__builtin_inff();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (lane_id() == 0) { atomicMin(box + b * 8 + d, f2ord_u(lo)); atomicMax(box + b * 8 + 3 + d, f2ord_u(hi)); }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) nn = fmaxf(nn, __shfl_xor(nn, o));
  if (lane_id() == 0) atomicMax(box + b * 8 + 6, __float_as_uint(nn));
}

__global__ void tile_morton_kernel(const float *__restrict__ x, long sb, long sd, long sn, int N, const unsigned int *__restrict__ box,
                                   unsigned long long *__restrict__ keys, int *__restrict__ vals) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  unsigned int code = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float lo = ord2f_u(box[b * 8 + d]), hi = ord2f_u(box[b * 8 + 3 + d]);
    const float ext = hi - lo;
    const float t = ext > 0.f ? (x[b * sb + d * sd + j * sn] - lo) / ext : 0.f;
    const unsigned int q = (unsigned int)fminf(fmaxf(t * 1023.f, 0.f), 1023.f);
    code |= morton_expand10(q) << d;
  }
  keys[(long)b * N + j] = ((unsigned long long)b << 32) | code;
  vals[(long)b * N + j] = j;
}

// one wave per tile: gather the sorted rows and reduce the tile's box
__global__ __launch_bounds__(256) void tile_gather_kernel(const float *__restrict__ x, long sb, long sd, long sn,
                                                          const float *__restrict__ xx, int C, int N, int T, const int *__restrict__ perm, float *__restrict__ xs,
                                                          float *__restrict__ xxs, float *__restrict__ bbox) {
  const int lane = lane_id();
  const int t = blockIdx.x * 4 + wave_id(), b = blockIdx.y;
  if (t >= T) return;
  const int p = t * 64 + lane;
  const bool ok = p < N;
  const int j = perm[(long)b * N + min(p, N - 1)];
  float v3[3];
  for (int d = 0; d < C; ++d) {
    const float v = x[b * sb + d * sd + j * sn];
    if (ok) xs[((long)b * C + d) * N + p] = v;
    if (d < 3) v3[d] = v;
  }
  const float xj = xx ? xx[(long)b * N + j] : 0.f;
  if (ok) xxs[(long)b * N + p] = xj;
  float red[7];
#pragma unroll
  for (int d = 0; d < 3; ++d) { red[d] = ok ? v3[d] : __builtin_inff(); red[3 + d] = ok ? v3[d] : -__builtin_inff(); }
  red[6] = ok ? xj : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
    for (int d = 0; d < 3; ++d) { red[d] = fminf(red[d], __shfl_xor(red[d], o)); red[3 + d] = fmaxf(red[3 + d], __shfl_xor(red[3 + d], o)); }
    red[6] = fmaxf(red[6], __shfl_xor(red[6], o));
  }
  if (lane < 7) {
    float out = red[0];
#pragma unroll
    for (int d = 1; d < 7; ++d) out = lane == d ? red[d] : out;
    bbox[((long)b * T + t) * 8 + lane] = out;
  }
}

template <int METRIC, int DIMC>   // METRIC 0: KNN_CUDA direct differences, 1: expanded form (DIMC = 3); 2: knn_points_normals (6)
__global__ __launch_bounds__(256) void knn_tiles_kernel(TileKnnArgs a) {
  constexpr int QW = 8;
  const int lane = lane_id(), wave = wave_id();
  const int b = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * QW;
  if (q0 >= a.N) return;
  const int N = a.N;
  const float *xs = a.xs + (long)b * DIMC * N;
  const float *xxs = a.xxs + (long)b * N;
  const int *perm = a.perm + (long)b * N;
  const float *bbox = a.bbox + (long)b * a.T * 8;
  int qi[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) qi[q] = min(q0 + q, N - 1);
  float qv[QW][DIMC], xxi[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) {
#pragma unroll
    for (int d = 0; d < DIMC; ++d) qv[q][d] = xs[(long)d * N + qi[q]];
    xxi[q] = xxs[qi[q]];
  }
  // lane q (< 8) keeps query q's position / norm / threshold for the box test
  const int ql = min(q0 + (lane & 7), N - 1);
  const float vx = xs[ql], vy = xs[(long)N + ql], vz = xs[2L * N + ql], vxx = xxs[ql];
  float thr_v = KNN_INF;
  TopB topb[QW];
  int cnt[QW];
  u64 thr64[QW];
#pragma unroll
  for (int q = 0; q < QW; ++q) { topb[q].init(); cnt[q] = 0; thr64[q] = ~0ull; }
  const int klane = a.k - 1;
  const int home = q0 >> 6;
  // METRIC 2: lower bound of the normal factor (1 + n_pd) = 3 - 2 n_i.n_j >= 3 - 2 max|n|^2; <= 0 disables pruning
  float fac_lb = 1.f;
  if (METRIC == 2) fac_lb = (3.f - 2.f * __uint_as_float(a.box[b * 8 + 6])) * (1.f - 1e-5f) - 1e-5f;
  const bool prune = METRIC != 2 || fac_lb > 0.f;
  const int span = max(home, a.T - 1 - home);
  for (int i = 0; i <= 2 * span; ++i) {
    const int off = (i + 1) >> 1;
    const int t = (i & 1) ? home + off : home - off;
    if (t < 0 || t >= a.T) continue;
    if (prune) {
      const float *bb = bbox + (long)t * 8;
      const float dx = fmaxf(fmaxf(bb[0] - vx, vx - bb[3]), 0.f), dy = fmaxf(fmaxf(bb[1] - vy, vy - bb[4]), 0.f),
                  dz = fmaxf(fmaxf(bb[2] - vz, vz - bb[5]), 0.f);
      float lb = fmaf(dz, dz, fmaf(dy, dy, dx * dx)) * (1.f - 1e-6f) - 2e-6f * (vxx + bb[6]);
      lb = fmaxf(lb, 0.f);
      if (METRIC == 2) lb *= fac_lb;
      const unsigned long long need = __ballot(lane < QW && q0 + lane < N && lb <= thr_v);
      if (!need) continue;
    }
    const int j = t * 64 + lane;
    const bool valid = j < N;
    const int jc = valid ? j : N - 1;
    float cv[DIMC];
#pragma unroll
    for (int d = 0; d < DIMC; ++d) cv[d] = xs[(long)d * N + jc];
    const float xxj = xxs[jc];
    const int oidx = perm[jc];
    float key[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      if (METRIC == 0) {                        // KNN_CUDA: direct differences (knn.cu:73-77)
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DIMC; ++d) { const float t2 = cv[d] - qv[q][d]; acc = fmaf(t2, t2, acc); }
        key[q] = acc;
      } else if (METRIC == 1) {
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DIMC; ++d) acc = fmaf(qv[q][d], cv[d], acc);
        const float tt = 2.f * acc - xxj;
        key[q] = -(tt - xxi[q]);
      } else {
        float dp = 0.f, dn = 0.f;
#pragma unroll
        for (int d = 0; d < 3; ++d) dp = fmaf(qv[q][d], cv[d], dp);
#pragma unroll
        for (int d = 3; d < 6; ++d) dn = fmaf(qv[q][d], cv[d], dn);
        const float p_pd = (xxj - 2.f * dp) + xxi[q];
        const float n_pd = 2.f - 2.f * dn;
        key[q] = p_pd * (1.f + n_pd);
      }
    }
#pragma unroll
    for (int q = 0; q < QW; ++q) {
      const u64 c64 = ((u64)key_f2u(key[q]) << 32) | (unsigned int)oidx;
      const bool pass = valid && c64 < thr64[q];
      const unsigned long long m = __ballot(pass);
      if (m) {
        if (cnt[q] + __popcll(m) > 64) {
          topb[q].merge(cnt[q], lane);
          cnt[q] = 0;
          const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(topb[q].lst >> 32), klane);
          const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)topb[q].lst, klane);
          thr64[q] = ((u64)hi << 32) | lo;
          const float tk = hi == 0xFFFFFFFFu ? KNN_INF : key_u2f(hi);   // list not full yet (short home tile): no bound
          thr_v = (lane & 7) == q ? tk : thr_v;
        }
        cnt[q] = topb[q].append(m, pass, key[q], oidx, cnt[q], lane);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < QW; ++q) {
    if (q0 + q >= N) break;
    topb[q].merge(cnt[q], lane);
    const int oq = perm[q0 + q];
    if (lane < a.k && (lane % a.step) == 0) {
      const long o = (long)b * a.o_sb + (long)(lane / a.step) * a.o_sk + (long)oq * a.o_sq;
      a.ind[o] = (int64_t)(unsigned int)topb[q].lst;
      const float kv = key_u2f((unsigned int)(topb[q].lst >> 32));
      if (a.val) a.val[o] = METRIC == 0 ? sqrtf(kv) : -kv;
    }
  }
}

struct TileWs {
  unsigned int *box;            // (B, 8)
  unsigned long long *keys_a, *keys_b;
  int *vals_a, *perm;
  float *xs, *xxs, *bbox;
  void *tmp;
  size_t tmp_bytes, total;
};
static TileWs tile_ws_layout(char *base, int B, int C, int N) {
  TileWs L;
  size_t off = 0;
  auto take = [&](size_t bytes) { char *p = base ? base + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
  const size_t n = (size_t)B * N;
  const int T = (N + 63) / 64;
  L.box = (unsigned int *)take(32 * (size_t)B);
  L.keys_a = (unsigned long long *)take(8 * n); L.keys_b = (unsigned long long *)take(8 * n);
  L.vals_a = (int *)take(4 * n); L.perm = (int *)take(4 * n);
  L.xs = (float *)take(4 * n * C); L.xxs = (float *)take(4 * n); L.bbox = (float *)take(32 * (size_t)B * T);
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *)nullptr, (unsigned long long *)nullptr, (int *)nullptr,
                                  (int *)nullptr, n > 0 ? n : 1, 0, 64);
  L.tmp_bytes = tb;
  L.tmp = take(tb + 256);
  L.total = off;
  return L;
}

// kernel_metric: 0 KNN_CUDA, 1 expanded form, 2 points+normals.  x element (b,d,j) at b*sb + d*sd + j*sn.
static int run_knn_tiles(const float *x, long sb, long sd, long sn, const float *xx, int B, int C, int N, int k, int step,
                         int kernel_metric, long o_sb, long o_sk, long o_sq, int64_t *ind, float *val, void *ws,
                         hipStream_t st) {
  TileWs L = tile_ws_layout((char *)ws, B, C, N);
  const int T = (N + 63) / 64;
  tile_box_init_kernel<<<cdiv(B * 8, 256), 256, 0, st>>>(L.box, B * 8);
  const dim3 gp(cdiv(N, 256), B);
  tile_bbox_kernel<<<gp, 256, 0, st>>>(x, sb, sd, sn, N, kernel_metric == 2 ? 1 : 0, L.box);
  tile_morton_kernel<<<gp, 256, 0, st>>>(x, sb, sd, sn, N, L.box, L.keys_a, L.vals_a);
  size_t tb = L.tmp_bytes;
  int bits = 32;
  for (int v = B - 1; v > 0; v >>= 1) ++bits;
  GCN_HIP(rocprim::radix_sort_pairs(L.tmp, tb, L.keys_a, L.keys_b, L.vals_a, L.perm, (size_t)B * N, 0, bits, st));
  tile_gather_kernel<<<dim3(cdiv(T, 4), B), 256, 0, st>>>(x, sb, sd, sn, xx, C, N, T, L.perm, L.xs, L.xxs, L.bbox);
  TileKnnArgs a{};
  a.xs = L.xs; a.xxs = L.xxs; a.perm = L.perm; a.bbox = L.bbox;
  a.N = N; a.T = T; a.k = k; a.step = step; a.o_sb = o_sb; a.o_sk = o_sk; a.o_sq = o_sq; a.ind = ind; a.val = val;
  a.box = L.box;
  const dim3 grid(cdiv(N, 32), B);
  if (kernel_metric == 2) knn_tiles_kernel<2, 6><<<grid, 256, 0, st>>>(a);
  else if (kernel_metric == 1) knn_tiles_kernel<1, 3><<<grid, 256, 0, st>>>(a);
  else knn_tiles_kernel<0, 3><<<grid, 256, 0, st>>>(a);
  return check_launch("knn_tiles_kernel");
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_knn_cuda(const float *ref, const float *query, int B, int dim, int nr, int nq, int k,
                            int point_major, float *dist, int64_t *ind, void *tile_ws, void *stream) {
  GCN_REQUIRE(ref && query && dist && ind, "gcn_knn_cuda: null pointer");
  GCN_REQUIRE(B >= 0 && dim >= 1 && nr >= 1 && nq >= 0, "gcn_knn_cuda: bad shape B=%d dim=%d nr=%d nq=%d", B, dim, nr, nq);
  GCN_REQUIRE(k >= 1 && k <= nr && k <= 512, "gcn_knn_cuda: need 1 <= k <= min(nr,512), got k=%d nr=%d", k, nr);
  if (B == 0 || nq == 0) return GCN_OK;
  KnnArgs a{};
  a.ref = ref; a.query = query; a.xx = nullptr;
  a.dim = dim; a.nr = nr; a.nq = nq; a.k = k; a.step = 1;
  a.ref_sb = (long)dim * nr; a.q_sb = (long)dim * nq;
  if (point_major) {
    a.ref_sd = 1; a.ref_sn = dim; a.q_sd = 1; a.q_sn = dim;
    a.o_sk = 1; a.o_sq = k;
  } else {
    a.ref_sd = nr; a.ref_sn = 1; a.q_sd = nq; a.q_sn = 1;
    a.o_sk = nq; a.o_sq = 1;
  }
  a.o_sb = (long)k * nq;
  a.dist = dist; a.ind = ind;
  hipStream_t st = (hipStream_t)stream;
  // a 3-D cloud searched against itself at filter-friendly sizes: threshold + filter + re-rank (knn_normal.hip), then the
  // exhaustive kernel for the few queries it flags
  if (tile_ws && ref == query && nr == nq && dim == 3 && knn_normal_supported(B, nr, k)) {
    const unsigned char *flag = nullptr;
    int rc = run_knn_normal(0, ref, a.ref_sb, a.ref_sd, a.ref_sn, nullptr, B, 3, nr, k, 1, a.o_sb, a.o_sk, a.o_sq, ind, dist,
                            tile_ws, &flag, st);
    if (rc) return rc;
    a.only = flag;
    return launch_knn<0, 3>(a, B, st);
  }
  // a 3-D cloud searched against itself: the Morton-tiled kernel (same results, box pruning)
  if (tile_ws && ref == query && nr == nq && dim == 3 && k <= 64 && nr >= 512)
    return run_knn_tiles(ref, a.ref_sb, a.ref_sd, a.ref_sn, nullptr, B, 3, nr, k, 1, 0, a.o_sb, a.o_sk, a.o_sq, ind, dist, tile_ws, st);
  if (dim == 3) return launch_knn<0, 3>(a, B, st);
  return launch_knn<0, 0>(a, B, st);
}

namespace gcn {
// Exhaustive exact search of the queries marked in `flag` on the f32 matrix cores (channel-major x_cm), a no-op unless
// *gate > gate_min (knn_filter.hip, long fallback lists).
int launch_knn_mfma16_flagged(const float *x_cm, const float *xx, const unsigned char *flag, const unsigned int *gate,
                              unsigned int gate_min, int B, int N, int C, int k, int step, int kout, int64_t *idx, hipStream_t st) {
#define GCN_KNN_FL(CCV, TCV)                                                                                       \
  {                                                                                                                 \
    constexpr int PIECES = (CCV + 1 + 256 / TCV - 1) / (256 / TCV);                                                 \
    const int lds = 2 * PIECES * 1024;                                                                              \
    GCN_HIP(hipFuncSetAttribute((const void *)knn_mfma16_kernel<CCV, TCV, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    knn_mfma16_kernel<CCV, TCV, 1, true><<<dim3(cdiv(N, 128), B), 512, lds, st>>>(x_cm, xx, N, k, step, kout, idx, nullptr, flag, gate, gate_min); \
  }
  if (k > 64 || N < 64 || (N % 4) != 0) { set_error("launch_knn_mfma16_flagged: unsupported shape"); return GCN_EINVAL; }
  if (C == 32) GCN_KNN_FL(32, 64) else if (C == 64) GCN_KNN_FL(64, 64) else if (C == 128) GCN_KNN_FL(128, 32)
  else { set_error("launch_knn_mfma16_flagged: unsupported channel count"); return GCN_EINVAL; }
#undef GCN_KNN_FL
  return check_launch("knn_mfma16_kernel<flagged>");
}

int launch_knn_flagged(const float *x_pm, const float *xx, const unsigned char *flag, int B, int N, int C, int k, int step,
                       int kout, int64_t *idx, hipStream_t st) {
  KnnArgs a{};
  a.ref = x_pm; a.query = x_pm; a.xx = xx; a.only = flag;
  a.dim = C; a.nr = N; a.nq = N; a.k = k; a.step = step;
  a.ref_sb = a.q_sb = (long)C * N;
  a.ref_sd = a.q_sd = 1; a.ref_sn = a.q_sn = C;          // point-major rows
  a.o_sb = (long)N * kout; a.o_sk = 1; a.o_sq = kout;
  a.dist = nullptr; a.ind = idx;
  return launch_knn<1, 0>(a, B, st);
}
}  // namespace gcn

GCN_EXPORT long gcn_knn_tiles_ws_bytes(int B, int C, int N) {
  if (B < 0 || C < 1 || N < 1) return -1;
  size_t need = tile_ws_layout(nullptr, B, C, N).total;
  if ((C == 6 || C == 3) && knn_normal_supported(B, N, 1)) need = std::max(need, knn_normal_ws_bytes(B, N));
  return (long)need;
}

GCN_EXPORT int gcn_knn_normal_supported(int B, int N, int k2) { return knn_normal_supported(B, N, k2) ? 1 : 0; }

GCN_EXPORT int gcn_knn_model(const float *x, int B, int C, int N, int k1, int k2, int metric,
                             int64_t *idx, float *val, float *xx_ws, void *tile_ws, void *stream) {
  GCN_REQUIRE(x && idx && xx_ws, "gcn_knn_model: null pointer");
  GCN_REQUIRE(metric == 0 || metric == 1, "gcn_knn_model: metric must be 0 (knn) or 1 (knn_points_normals)");
  GCN_REQUIRE(B >= 0 && C >= 1 && N >= 1, "gcn_knn_model: bad shape B=%d C=%d N=%d", B, C, N);
  GCN_REQUIRE(metric == 0 || C >= 6, "gcn_knn_model: knn_points_normals needs C >= 6, got %d", C);
  GCN_REQUIRE(k1 >= 1 && k1 <= k2 && k2 <= N && k2 <= 512, "gcn_knn_model: need 1 <= k1 <= k2 <= min(N,512), got k1=%d k2=%d N=%d", k1, k2, N);
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int step = k2 / k1;
  const int kout = (k2 + step - 1) / step;
  dim3 g(cdiv(N, 256), B);
  sqnorm_kernel<<<g, 256, 0, st>>>(x, xx_ws, C, N, metric == 0 ? C : 3);
  int rc = check_launch("sqnorm_kernel");
  if (rc) return rc;
  KnnArgs a{};
  a.ref = x; a.query = x; a.xx = xx_ws;
  a.dim = C; a.nr = N; a.nq = N; a.k = k2; a.step = step;
  a.ref_sb = a.q_sb = (long)C * N;
  a.ref_sd = a.q_sd = N; a.ref_sn = a.q_sn = 1;
  a.o_sb = (long)N * kout; a.o_sk = 1; a.o_sq = kout;
  a.dist = val; a.ind = idx;
  // xyz + normal clouds at filter-friendly sizes: threshold + filter + re-rank (knn_normal.hip), then the exhaustive
  // kernel for the few queries it flags
  if (metric == 1 && C == 6 && tile_ws && knn_normal_supported(B, N, k2)) {
    const unsigned char *flag = nullptr;
    rc = run_knn_normal(1, x, (long)C * N, N, 1, xx_ws, B, C, N, k2, step, (long)N * kout, 1, kout, idx, val, tile_ws, &flag, st);
    if (rc) return rc;
    a.only = flag;
    return launch_knn<2, 6>(a, B, st);
  }
  if (metric == 0 && C == 3 && tile_ws && knn_normal_supported(B, N, k2)) {      // the same scheme for `knn` on xyz
    const unsigned char *flag = nullptr;
    rc = run_knn_normal(2, x, (long)C * N, N, 1, xx_ws, B, C, N, k2, step, (long)N * kout, 1, kout, idx, val, tile_ws, &flag, st);
    if (rc) return rc;
    a.only = flag;
    return launch_knn<1, 3>(a, B, st);
  }
  // 3-D clouds: Morton-tiled kernel with bounding-box pruning (identical results, ~5x fewer candidates)
  const bool tiled = tile_ws && k2 <= 64 && N >= 512 && ((metric == 1 && C == 6) || (metric == 0 && C == 3));
  if (tiled)
    return run_knn_tiles(x, (long)C * N, N, 1, xx_ws, B, C, N, k2, step, metric == 1 ? 2 : 1, (long)N * kout, 1, kout, idx, val,
                         tile_ws, st);
  if (metric == 1) return launch_knn<2, 6>(a, B, st);
  if (C == 3) return launch_knn<1, 3>(a, B, st);
  if (k2 <= 128 && N >= 64 && (N % 4) == 0) {  // matrix-core path (bit-identical dot products)
    if (C == 32) return launch_knn_mfma16<32, 64>(x, xx_ws, B, N, k2, step, kout, idx, val, st);
    if (C == 64) return launch_knn_mfma16<64, 64>(x, xx_ws, B, N, k2, step, kout, idx, val, st);
    if (C == 128) return launch_knn_mfma16<128, 32>(x, xx_ws, B, N, k2, step, kout, idx, val, st);
  }
  return launch_knn<1, 0>(a, B, st);
}

GCN_EXPORT int gcn_topk_rows(const void *x, int dtype, long R, int NK, int k, float *vals, int64_t *idx, void *stream) {
  GCN_REQUIRE(x && vals && idx, "gcn_topk_rows: null pointer");
  GCN_REQUIRE(dtype == 0 || dtype == 1, "gcn_topk_rows: dtype must be 0 (f32) or 1 (bf16)");
  GCN_REQUIRE(R >= 0 && NK >= 1 && NK <= 128 && k >= 1 && k <= 64 && k <= NK, "gcn_topk_rows: need NK <= 128, k <= min(64, NK)");
  if (R == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 1) topk_rows_kernel<true><<<cdiv(R, 4), 256, 0, st>>>(x, R, NK, k, vals, idx);
  else topk_rows_kernel<false><<<cdiv(R, 4), 256, 0, st>>>(x, R, NK, k, vals, idx);
  return check_launch("topk_rows_kernel");
}
