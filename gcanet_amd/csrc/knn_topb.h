// knn_topb.h -- buffered bitonic top-64 list shared by the kNN kernels (knn.hip, knn_filter.hip)
#pragma once
#include "common.h"

#include <type_traits>

namespace gcn {

typedef unsigned long long u64;
#define TOPB_SENT 0xFFFFFFFF7FFFFFFFull

__device__ __forceinline__ unsigned int key_f2u(float x) {
  const unsigned int u = __float_as_uint(x + 0.0f);          // -0 -> +0
  return u ^ ((unsigned int)((int)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_u2f(unsigned int u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}
__device__ __forceinline__ u64 shfl_u64(u64 v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned int)v);
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(unsigned int)(v >> 32));
  return ((u64)(unsigned int)hi << 32) | (unsigned int)lo;
}

struct TopB {
  u64 lst, pnd;

  __device__ __forceinline__ void init() { lst = TOPB_SENT; pnd = TOPB_SENT; }

  // one compare-exchange stage with partner lane^j; `asc_block` = this lane's block sorts ascending
  __device__ __forceinline__ static u64 cex(u64 v, int lane, int j, bool asc_block) {
    const u64 o = shfl_u64(v, lane ^ j);
    const bool lower = (lane & j) == 0;
    const bool take_min = lower == asc_block;
    const bool o_lt = o < v;
    return (o_lt == take_min) ? o : v;
  }

  // merge the first `cnt` pending entries into the sorted list
  __device__ __forceinline__ void merge(int cnt, int lane) {
    u64 p = lane < cnt ? pnd : TOPB_SENT;
#pragma unroll
    for (int sz = 2; sz <= 64; sz <<= 1)
#pragma unroll
      for (int j = sz >> 1; j >= 1; j >>= 1) p = cex(p, lane, j, (lane & sz) == 0);
    const u64 r = shfl_u64(p, 63 - lane);       // descending copy of the sorted pending entries
    u64 m = r < lst ? r : lst;                  // the 64 smallest of the union, as a bitonic sequence
#pragma unroll
    for (int j = 32; j >= 1; j >>= 1) m = cex(m, lane, j, true);
    lst = m;
  }

  // sort the first `cnt` pending entries into the (empty) list
  __device__ __forceinline__ void sort_pending(int cnt, int lane) {
    u64 p = lane < cnt ? pnd : TOPB_SENT;
#pragma unroll
    for (int sz = 2; sz <= 64; sz <<= 1)
#pragma unroll
      for (int j = sz >> 1; j >= 1; j >>= 1) p = cex(p, lane, j, (lane & sz) == 0);
    lst = p;
  }

  // append the candidates of the lanes in `mask` (key, idx); returns the new pending count
  __device__ __forceinline__ int append(unsigned long long mask, bool pass, float key, int idx, int cnt, int lane) {
    const int p = __popcll(mask);
    const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0));
    // full permutation: passing lanes -> [cnt, cnt+p), the others -> the complement (their data is ignored)
    const int dest = pass ? cnt + rank : ((cnt + p + lane - rank) & 63);
    const int lo = __builtin_amdgcn_ds_permute(dest << 2, idx);
    const int hi = __builtin_amdgcn_ds_permute(dest << 2, (int)key_f2u(key));
    const bool in = (unsigned int)(lane - cnt) < (unsigned int)p;
    pnd = in ? (((u64)(unsigned int)hi << 32) | (unsigned int)lo) : pnd;
    return cnt + p;
  }
};


// ascending bitonic sort of 128 entries held two per lane: `a` = position lane, `b` = position lane + 64
__device__ __forceinline__ void sort128(u64 &a, u64 &b, int lane) {
#pragma unroll
  for (int sz = 2; sz <= 64; sz <<= 1)
#pragma unroll
    for (int j = sz >> 1; j >= 1; j >>= 1) {
      a = TopB::cex(a, lane, j, (lane & sz) == 0);
      b = TopB::cex(b, lane, j, ((lane + 64) & sz) == 0);          // sz == 64: the upper half sorts descending
    }
  {                                                                  // sz = 128, j = 64: partner sits in the same lane
    const u64 lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
  }
#pragma unroll
  for (int j = 32; j >= 1; j >>= 1) {
    a = TopB::cex(a, lane, j, true);
    b = TopB::cex(b, lane, j, true);
  }
}

// The k smallest of up to 512 candidates held 8 per lane as (monotone integer key kf[bt], index cj[bt]), batch bt valid
// when bt*64 < total (empty slots carry 0xFFFFFFFF), sorted by (key, index): positions 0..63 in tb.lst, and -- for
// 64 < k <= 128 -- positions 64..127 in tb.pnd (buf: a per-wave LDS buffer of 128 entries, used only then).  The k-th
// smallest key VALUE comes from an MSB-first search with wave-wide counts (scalar unit), then ONE bitonic sort of the
// candidates at or below it instead of a sort + merge per batch of 64.  Returns the k-th smallest key (integer image).
// Used by the re-rank kernels of knn_filter.hip / knn_normal.hip.
__device__ __forceinline__ unsigned int rank_candidates(const unsigned int (&kf)[8], const int (&cj)[8], int total, int k, int lane,
                                                        TopB &tb, u64 *buf = nullptr) {
  const int nb = (total + 63) >> 6;
  auto kth_key = [&](auto nbc) -> unsigned int {
    constexpr int NB = decltype(nbc)::value;
    unsigned int pk = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const unsigned int trial = pk | (1u << bit);
      int c = 0;
#pragma unroll
      for (int bt = 0; bt < NB; ++bt) c += __popcll(__ballot(kf[bt] < trial));
      pk = c >= k ? pk : trial;
    }
    return pk;
  };
  unsigned int pk;
  switch (nb) {
    case 1: pk = kth_key(std::integral_constant<int, 1>{}); break;
    case 2: pk = kth_key(std::integral_constant<int, 2>{}); break;
    case 3: pk = kth_key(std::integral_constant<int, 3>{}); break;
    case 4: pk = kth_key(std::integral_constant<int, 4>{}); break;
    case 5: pk = kth_key(std::integral_constant<int, 5>{}); break;
    case 6: pk = kth_key(std::integral_constant<int, 6>{}); break;
    case 7: pk = kth_key(std::integral_constant<int, 7>{}); break;
    default: pk = kth_key(std::integral_constant<int, 8>{}); break;
  }
  tb.init();
  int npend = 0;
  int nle = 0, nlt = 0;
#pragma unroll
  for (int bt = 0; bt < 8; ++bt)
    if (bt * 64 < total) {
      nle += __popcll(__ballot(kf[bt] <= pk));
      nlt += __popcll(__ballot(kf[bt] < pk));
    }
  // candidates strictly below the k-th key all belong to the result; of those EQUAL to it the lowest indices fill
  // the remaining k - nlt places (ties -> lowest index, as the reference's stable insertion).  Usually nle == k and
  // the index bound is the maximum.
  int jmax = 0x7fffffff;
  if (nle > k) {                                             // wave-uniform, rare: an exact tie straddles the k-th place
    const int need = k - nlt;
    int pj = 0;
    for (int bit = 15; bit >= 0; --bit) {                    // largest pj with fewer than `need` tied indices below it
      const int trial = pj | (1 << bit);
      int c = 0;
#pragma unroll
      for (int bt = 0; bt < 8; ++bt)
        if (bt * 64 < total) c += __popcll(__ballot(kf[bt] == pk && cj[bt] < trial));
      pj = c >= need ? pj : trial;
    }
    jmax = pj;                                               // the need-th smallest tied index
  }
  if (k <= 64) {
#pragma unroll
    for (int bt = 0; bt < 8; ++bt)
      if (bt * 64 < total) {
        const bool pass = kf[bt] < pk || (kf[bt] == pk && cj[bt] <= jmax);
        const unsigned long long m = __ballot(pass);
        if (m) npend = tb.append(m, pass, key_u2f(kf[bt]), cj[bt], npend, lane);
      }
    tb.sort_pending(npend, lane);
    return pk;
  }
  // 64 < k <= 128: the (exactly k) passing candidates are compacted into the LDS buffer, two entries per lane are sorted
  buf[lane] = TOPB_SENT;
  buf[lane + 64] = TOPB_SENT;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int bt = 0; bt < 8; ++bt)
    if (bt * 64 < total) {
      const bool pass = kf[bt] < pk || (kf[bt] == pk && cj[bt] <= jmax);
      const unsigned long long m = __ballot(pass);
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0));
      if (pass && npend + rank < 128) buf[npend + rank] = ((u64)kf[bt] << 32) | (unsigned int)cj[bt];
      npend += __popcll(m);
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  tb.lst = buf[lane];
  tb.pnd = buf[lane + 64];
  sort128(tb.lst, tb.pnd, lane);
  return pk;
}

// exact kNN in the model's expanded form for the queries whose flag byte is set (the safety net of knn_filter.hip);
// x_pm (B,N,C) point-major, xx (B,N), flag (B,N), idx (B,N,kout).  Implemented in knn.hip on knn_select_kernel.
int launch_knn_mfma16_flagged(const float *x_cm, const float *xx, const unsigned char *flag, const unsigned int *gate,
                              unsigned int gate_min, int B, int N, int C, int k, int step, int kout, int64_t *idx, hipStream_t st);
int launch_knn_flagged(const float *x_pm, const float *xx, const unsigned char *flag, int B, int N, int C, int k, int step,
                       int kout, int64_t *idx, hipStream_t st);

// knn_points_normals by threshold + filter + re-rank (knn_normal.hip).  The caller runs the flagged queries (flag byte
// set) through knn_select_kernel afterwards.
bool knn_normal_supported(int B, int N, int k);
size_t knn_normal_ws_bytes(int B, int N);
// The candidate row of slot i of the N/8-row sample the threshold passes use: a hash of the slot number, i.e. rows drawn
// (with replacement) independently of the cloud's storage order.  A strided sample (round 2: row 8 i + 3) meets a cloud
// stored cluster by cluster in rotation at the same few clusters for every query; and one row of every 8 consecutive
// ones still ties the slot number -- hence the lane that keeps the slot's value -- to the cluster: every query of
// bench.blob_clouds then overshoots its threshold (tools/debug/knn_flagged_why.py).
__host__ __device__ __forceinline__ int knn_sample_row(int i, int N) {
  unsigned int h = (unsigned int)i + 0x9E3779B9u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;        // murmur3 finaliser
  return (int)(((unsigned long long)h * (unsigned int)N) >> 32);
}
int run_knn_normal(int metric, const float *x, long sb, long sd, long sn, const float *xx, int B, int C, int N, int k, int step,
                   long o_sb, long o_sk, long o_sq, int64_t *idx, float *val, void *ws, const unsigned char **flag_out,
                   hipStream_t st);

}  // namespace gcn
