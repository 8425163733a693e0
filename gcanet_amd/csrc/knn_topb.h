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


// The k smallest of up to 512 candidates held 8 per lane as (monotone integer key kf[bt], index cj[bt]), batch bt valid
// when bt*64 < total (empty slots carry 0xFFFFFFFF), sorted by (key, index) into tb.lst.  The k-th smallest key VALUE
// comes from an MSB-first search with wave-wide counts (scalar unit), then ONE bitonic sort of the candidates at or
// below it instead of a sort + merge per batch of 64.  Used by the re-rank kernels of knn_filter.hip / knn_normal.hip.
__device__ __forceinline__ void rank_candidates(const unsigned int (&kf)[8], const int (&cj)[8], int total, int k, int lane,
                                                TopB &tb) {
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
  if (nle > 64) {                                            // wave-uniform, rare: an exact tie straddles the k-th place
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
#pragma unroll
  for (int bt = 0; bt < 8; ++bt)
    if (bt * 64 < total) {
      const bool pass = kf[bt] < pk || (kf[bt] == pk && cj[bt] <= jmax);
      const unsigned long long m = __ballot(pass);
      if (m) npend = tb.append(m, pass, key_u2f(kf[bt]), cj[bt], npend, lane);
    }
  tb.sort_pending(npend, lane);
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
int run_knn_normal(int metric, const float *x, long sb, long sd, long sn, const float *xx, int B, int C, int N, int k, int step,
                   long o_sb, long o_sk, long o_sq, int64_t *idx, float *val, void *ws, const unsigned char **flag_out,
                   hipStream_t st);

}  // namespace gcn
