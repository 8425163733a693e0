// knn_normal.hip -- knn_points_normals (models/dgcnn-hais-concat-direct-4.py:50-90: xyz + normal clouds, key =
// |p_i - p_j|^2 (expanded form) * (3 - 2 n_i.n_j)) as THRESHOLD + FILTER + RE-RANK, all in the reference's own f32
// arithmetic, for gfx950.
//
// Why.  The key costs 11 VALU operations; what made the full scan (knn.hip: knn_select_kernel<1,8,2,6>, 0.71 ms at
// B=8, N=8192, k=64) slow is the sorted-list bookkeeping behind it: ~k ln(N/k) inserts and half a dozen bitonic merges
// per query.  Box pruning (knn_tiles_kernel) does not help when the normals are incoherent: the factor 3 - 2 n.n spans
// [1,5] and more than half of the Morton tiles survive.  Here nothing is inserted during the N^2 pass:
//   1. prep       (B,6,N) channel-major -> 32-byte rows {x y z nx ny nz |xyz|^2 0}.
//   2. threshold  knnn_sample_kernel: for every query the m-th smallest key over a 1-in-8 pseudo-random sample of the
//                 candidates (m ~ k/8 + 6 sqrt(k/8)): a value tau with, almost surely, at least k keys under it.
//                 Four lanes share a query (a quarter of the sample each, three smallest per group of samples kept in
//                 registers), the order statistic comes from an MSB-first search on the monotone integer image.
//   3. filter     knnn_filter_kernel: key(q,j) <= tau_q for ALL pairs -> one bit per pair (B*N*N/8 bytes).  Lane = two
//                 queries (packed f32 arithmetic: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 are exact IEEE per
//                 component), the candidate is wave-uniform and arrives through the scalar cache (s_load_dwordx8); the
//                 test result enters the lane's bitmap word by v_cmp + v_addc (w = 2w + carry).  8.5 instructions per
//                 64 pairs instead of 13 + list upkeep.
//   4. re-rank    knnn_rerank_kernel: one wave per query expands its ~3k bits, re-evaluates those keys (bitwise the
//                 same expression as the filter), and sorts by (key, index): lowest index wins ties, as in the oracle.
//      Clouds of any size 1024 <= N <= 16384 (rows padded to a multiple of 1024 with far-away points whose key exceeds
//      every threshold), k <= 128 (two sorted entries per lane beyond 64).
//   5. fallback   a query with fewer than k or more than 512 bits is flagged and searched exhaustively by
//                 knn_select_kernel in its flagged-only mode (knn.hip).
// Every key is the oracle's expression (oracle/gcanet_oracle.c:model_pd, metric 1), so the result is exact by
// construction: all candidates at or below tau are kept, at least k of them exist, the k-th smallest key is <= tau.
#include "common.h"
#include "knn_topb.h"

namespace gcn {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int KNNN_CAP = 512;
constexpr int KNNN_STRIDE = 8;       // the sample holds N/8 hashed candidate rows (knn_topb.h: knn_sample_row)
constexpr int KNNN_CS = 8;           // candidate ranges per query block in the filter pass

// M = 1: oracle/gcanet_oracle.c:model_pd metric 1 (M4:62-75): q = query i, c = candidate j; rows {x y z nx | ny nz xx 0}
// M = 0: KNN_CUDA's squared distance (knn.cu:73-77), differences candidate - query summed by fmaf in x, y, z order
// M = 2: the in-model `knn` on a 3-D cloud (M4:36-38, metric 0 of model_pd): -(fl(fl(2 dot - xx_j) - xx_i)); rows carry xx
template <int M>
__device__ __forceinline__ float knnn_key(const float4 q0, const float4 q1, const float4 c0, const float4 c1) {
  if (M == 0) {
    const float tx = c0.x - q0.x, ty = c0.y - q0.y, tz = c0.z - q0.z;
    float acc = fmaf(tx, tx, 0.f);
    acc = fmaf(ty, ty, acc);
    acc = fmaf(tz, tz, acc);
    return acc;
  }
  if (M == 2) {
    float d3 = fmaf(q0.x, c0.x, 0.f);
    d3 = fmaf(q0.y, c0.y, d3);
    d3 = fmaf(q0.z, c0.z, d3);
    const float t = fmaf(2.f, d3, -c1.z);                    // fl(2 dot - xx_j): 2 dot is exact
    return q1.z - t;                                         // -(t - xx_i)
  }
  float dp = fmaf(q0.x, c0.x, 0.f);
  dp = fmaf(q0.y, c0.y, dp);
  dp = fmaf(q0.z, c0.z, dp);
  float dn = fmaf(q0.w, c0.w, 0.f);
  dn = fmaf(q1.x, c1.x, dn);
  dn = fmaf(q1.y, c1.y, dn);
  const float p_pd = fmaf(-2.f, dp, c1.z) + q1.z;          // fl(fl(xx_j - 2 dp) + xx_i): 2 dp is exact
  const float n_pd = fmaf(-2.f, dn, 2.f);                  // fl(2 - 2 dn)
  return p_pd * (1.f + n_pd);
}

// ------------------------------------------------------------------ 1. rows
__global__ __launch_bounds__(256) void knnn_prep_kernel(const float *__restrict__ x, const float *__restrict__ xx,
                                                        float *__restrict__ rows, int N, int Np, int C, long sb, long sd, long sn) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (j >= Np) return;
  if (j >= N) {       // padding candidate: far away, finite in every metric's key (> any threshold), never a NaN
    float4 *o = reinterpret_cast<float4 *>(rows + ((long)b * Np + j) * 8);
    o[0] = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    o[1] = make_float4(0.f, 0.f, 3e36f, 0.f);
    return;
  }
  const float *p = x + (long)b * sb + (long)j * sn;         // element (b, d, j) at b*sb + d*sd + j*sn
  float4 r0, r1;
  r0.x = p[0]; r0.y = p[sd]; r0.z = p[2 * sd];
  r0.w = C >= 6 ? p[3 * sd] : 0.f;
  r1.x = C >= 6 ? p[4 * sd] : 0.f; r1.y = C >= 6 ? p[5 * sd] : 0.f;
  r1.z = xx ? xx[(long)b * N + j] : 0.f; r1.w = 0.f;
  float4 *o = reinterpret_cast<float4 *>(rows + ((long)b * Np + j) * 8);
  o[0] = r0;
  o[1] = r1;
}

// ------------------------------------------------------------------ 2. thresholds from the strided sample
// workgroup = 64 queries; lane l of wave w: query 16 w + (l & 15), sample quarter l >> 4.  The N/8 sample rows sit in
// LDS; the four quarters of a wave read rows one apart (different banks), every 16-lane group the same address.
template <int M>
__global__ __launch_bounds__(256) void knnn_sample_kernel(const float *__restrict__ rows, float *__restrict__ tau, int N,
                                                          int Np, int m_rank) {
  extern __shared__ __attribute__((aligned(16))) float4 smp[];      // (N/8, 2)
  const int lane = lane_id(), wave = wave_id();
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;              // cloud = id % B: one cloud per XCD at 8 clouds
  const int b = lin % (int)gridDim.y;
  const int q = (lin / (int)gridDim.y) * 64 + wave * 16 + (lane & 15);
  const int part = lane >> 4;
  const int ns = N / KNNN_STRIDE;
  const float4 *rb = reinterpret_cast<const float4 *>(rows + (long)b * Np * 8);
  for (int i = threadIdx.x; i < 2 * ns; i += 256) smp[i] = rb[(long)knn_sample_row(i >> 1, N) * 2 + (i & 1)];
  const int qc = min(q, N - 1);
  const float4 q0 = rb[(long)qc * 2], q1 = rb[(long)qc * 2 + 1];
  __syncthreads();
  const int gs = (ns + 31) / 32;                                     // samples per (quarter, group); slots >= ns are empty
  float t0[8], t1[8], t2[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    t0[g] = __builtin_inff(); t1[g] = __builtin_inff(); t2[g] = __builtin_inff();
    const int base = (part * 8 + g) * gs;
    for (int s = 0; s < gs; ++s) {
      int ss = s + part;
      ss = ss >= gs ? ss - gs : ss;
      const int si = min(base + ss, ns - 1);
      const float4 c0 = smp[si * 2], c1 = smp[si * 2 + 1];
      const float key = base + ss < ns ? knnn_key<M>(q0, q1, c0, c1) : __builtin_inff();
      const float x1 = fmaxf(t0[g], key);
      t0[g] = fminf(t0[g], key);
      const float x2 = fmaxf(t1[g], x1);
      t1[g] = fminf(t1[g], x1);
      t2[g] = fminf(t2[g], x2);
    }
  }
  // m-th smallest of the 96 values the query's four lanes kept; rounded UP to the next 2^8 boundary of the integer
  // image, so at least m sample keys are <= tau
  unsigned int k0[8], k1[8], k2[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) { k0[g] = key_f2u(t0[g]); k1[g] = key_f2u(t1[g]); k2[g] = key_f2u(t2[g]); }
  unsigned int p = 0;
  for (int bit = 31; bit >= 8; --bit) {
    const unsigned int trial = p | (1u << bit);
    int c = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) c += (k0[g] < trial ? 1 : 0) + (k1[g] < trial ? 1 : 0) + (k2[g] < trial ? 1 : 0);
    c += __shfl_xor(c, 16);
    c += __shfl_xor(c, 32);
    p = c >= m_rank ? p : trial;
  }
  if (part == 0 && q < N) tau[(long)b * N + q] = key_u2f(p | 0xffu);
}

// ------------------------------------------------------------------ 3. filter: one bit per (query, candidate)
// workgroup = 4 waves x 128 queries (lane: q and q + 64) x one of KNNN_CS candidate ranges.  The candidate row is
// wave-uniform: the loads below are scalar (s_load_dwordx8 through the scalar cache).
template <int M>
__global__ __launch_bounds__(256) void knnn_filter_kernel(const float *__restrict__ rows, const float *__restrict__ tau,
                                                          unsigned int *__restrict__ bitmap, int N, int Np, int B) {
  const int lane = lane_id(), wave = wave_id();
  const int lin = blockIdx.x;
  const int b = lin % B;
  const int rest = lin / B;
  const int range = rest % KNNN_CS;
  const int qa = (rest / KNNN_CS) * 512 + wave * 128 + lane;
  if (qa - lane >= N) return;
  const int NW = Np / 32;
  const float *rb = rows + (long)b * Np * 8;
  const int qA = min(qa, N - 1), qB = min(qa + 64, N - 1);
  const float4 a0 = reinterpret_cast<const float4 *>(rb)[(long)qA * 2], a1 = reinterpret_cast<const float4 *>(rb)[(long)qA * 2 + 1];
  const float4 b0 = reinterpret_cast<const float4 *>(rb)[(long)qB * 2], b1 = reinterpret_cast<const float4 *>(rb)[(long)qB * 2 + 1];
  const f32x2 qx = {a0.x, b0.x}, qy = {a0.y, b0.y}, qz = {a0.z, b0.z}, qnx = {a0.w, b0.w}, qny = {a1.x, b1.x},
              qnz = {a1.y, b1.y}, qxx = {a1.z, b1.z};
  const float tauA = tau[(long)b * N + qA], tauB = tau[(long)b * N + qB];
  const f32x2 zero2 = {0.f, 0.f}, two2 = {2.f, 2.f}, one2 = {1.f, 1.f}, m2 = {-2.f, -2.f};
  const int per = Np / KNNN_CS;                // candidates of this range: a multiple of 128 (Np % 1024 == 0)
  const int j0 = range * per;
  for (int w4 = 0; w4 < per / 128; ++w4) {
    unsigned int wa[4], wb[4];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww) {
      unsigned int ua = 0, ub = 0;
      for (int i8 = 0; i8 < 4; ++i8) {
        // eight candidate rows into scalar registers first (one wait), then the arithmetic
        const int jb = __builtin_amdgcn_readfirstlane(j0 + w4 * 128 + ww * 32 + i8 * 8);
        const float4 *cb = reinterpret_cast<const float4 *>(rb + (long)jb * 8);
        float4 r0[8], r1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { r0[u] = cb[2 * u]; r1[u] = cb[2 * u + 1]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const f32x2 cx = {r0[u].x, r0[u].x}, cy = {r0[u].y, r0[u].y}, cz = {r0[u].z, r0[u].z};
          f32x2 key;
          if (M == 0) {                                      // six packed operations + the test
            const f32x2 tx = cx - qx, ty = cy - qy, tz = cz - qz;
            key = __builtin_elementwise_fma(tx, tx, zero2);
            key = __builtin_elementwise_fma(ty, ty, key);
            key = __builtin_elementwise_fma(tz, tz, key);
          } else if (M == 2) {
            const f32x2 cxx = {r1[u].z, r1[u].z};
            f32x2 d3 = __builtin_elementwise_fma(qx, cx, zero2);
            d3 = __builtin_elementwise_fma(qy, cy, d3);
            d3 = __builtin_elementwise_fma(qz, cz, d3);
            key = qxx - __builtin_elementwise_fma(two2, d3, -cxx);
          } else {
            const f32x2 cnx = {r0[u].w, r0[u].w}, cny = {r1[u].x, r1[u].x}, cnz = {r1[u].y, r1[u].y}, cxx = {r1[u].z, r1[u].z};
            f32x2 dp = __builtin_elementwise_fma(qx, cx, zero2);
            dp = __builtin_elementwise_fma(qy, cy, dp);
            dp = __builtin_elementwise_fma(qz, cz, dp);
            f32x2 dn = __builtin_elementwise_fma(qnx, cnx, zero2);
            dn = __builtin_elementwise_fma(qny, cny, dn);
            dn = __builtin_elementwise_fma(qnz, cnz, dn);
            const f32x2 p_pd = __builtin_elementwise_fma(m2, dp, cxx) + qxx;
            const f32x2 n_pd = __builtin_elementwise_fma(m2, dn, two2);
            key = p_pd * (one2 + n_pd);
          }
          // w = 2 w + (key <= tau): the first candidate of a word ends in bit 31 (reversed on store)
          asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ua) : "v"(key.x), "v"(tauA) : "vcc");
          asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ub) : "v"(key.y), "v"(tauB) : "vcc");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      wa[ww] = __brev(ua);
      wb[ww] = __brev(ub);
    }
    const int wi = (j0 + w4 * 128) / 32;
    if (qa < N) *reinterpret_cast<uint4 *>(bitmap + ((long)b * N + qa) * NW + wi) = make_uint4(wa[0], wa[1], wa[2], wa[3]);
    if (qa + 64 < N) *reinterpret_cast<uint4 *>(bitmap + ((long)b * N + qa + 64) * NW + wi) = make_uint4(wb[0], wb[1], wb[2], wb[3]);
  }
}

// ------------------------------------------------------------------ 4. re-rank
template <int M>
__global__ __launch_bounds__(256) void knnn_rerank_kernel(const float *__restrict__ rows, const unsigned int *__restrict__ bitmap,
                                                          unsigned char *__restrict__ flag, int64_t *__restrict__ idx,
                                                          float *__restrict__ val, int N, int Np, int k, int step, long o_sb,
                                                          long o_sk, long o_sq) {
  __shared__ __attribute__((aligned(16))) unsigned short cand_s[4][KNNN_CAP];     // later the 128-entry sort buffer (k > 64)
  const int lane = lane_id(), wave = wave_id();
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  const int b = lin % (int)gridDim.y;
  const int q = (lin / (int)gridDim.y) * 4 + wave;
  if (q >= N) return;
  const int NW = Np / 32;
  const unsigned int *bm = bitmap + ((long)b * N + q) * NW;
  unsigned int words[8];
  int cnt = 0;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int wi = lane + 64 * w;
    words[w] = wi < NW ? bm[wi] : 0u;
    cnt += __popc(words[w]);
  }
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int y = __shfl_up(incl, d);
    if (lane >= d) incl += y;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const bool bad = total < k || total > KNNN_CAP;          // wave-uniform
  if (lane == 0) flag[(long)b * N + q] = bad ? 1 : 0;
  if (bad) return;                                          // knn_select_kernel (flagged-only mode) searches this query

  unsigned short *cand = cand_s[wave];
  int pos = incl - cnt;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    unsigned int word = words[w];
    const int base = (lane + 64 * w) * 32;
    while (word) {
      const int p = __ffs((int)word) - 1;
      word &= word - 1;
      cand[pos++] = (unsigned short)(base + p);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  const float4 *rb = reinterpret_cast<const float4 *>(rows + (long)b * Np * 8);
  const float4 q0 = rb[(long)q * 2], q1 = rb[(long)q * 2 + 1];
  unsigned int kf[8];
  int cj[8];
#pragma unroll
  for (int bt = 0; bt < 8; ++bt) {
    kf[bt] = 0xFFFFFFFFu;
    cj[bt] = q;
    if (bt * 64 < total) {                                   // wave-uniform
      const int c = bt * 64 + lane;
      const bool valid = c < total;
      const int j = valid ? (int)cand[c] : q;
      const float key = knnn_key<M>(q0, q1, rb[(long)j * 2], rb[(long)j * 2 + 1]);
      kf[bt] = valid ? key_f2u(key) : 0xFFFFFFFFu;
      cj[bt] = j;
    }
  }
  TopB tb;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the candidate list is dead: its LDS becomes the sort buffer
  __builtin_amdgcn_wave_barrier();
  rank_candidates(kf, cj, total, k, lane, tb, reinterpret_cast<u64 *>(cand));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int pos = lane + 64 * h;
    const u64 e = h ? tb.pnd : tb.lst;
    if (pos < k && (pos % step) == 0) {
      const long o = (long)b * o_sb + (long)(pos / step) * o_sk + (long)q * o_sq;    // output element (b, t, q)
      idx[o] = (int64_t)(unsigned int)e;
      const float kv = key_u2f((unsigned int)(e >> 32));
      if (val) val[o] = M == 0 ? sqrtf(kv) : -kv;
    }
  }
}

static size_t knnn_align(size_t v) { return (v + 255) & ~(size_t)255; }

static int knnn_np(int N) { return (N + 1023) / 1024 * 1024; }

bool knn_normal_supported(int B, int N, int k) {
  return B >= 1 && N >= 1024 && N <= 16384 && k >= 1 && k <= 128 && k <= N;
}

size_t knn_normal_ws_bytes(int B, int N) {
  const size_t n = (size_t)B * N, np = (size_t)B * knnn_np(N);
  return knnn_align(np * 32) + knnn_align(n * 4) + knnn_align(n) + knnn_align(n * (size_t)(knnn_np(N) / 8));
}

// metric 1: x (B,6,N) xyz + normal, xx (B,N) |xyz|^2 in the oracle's order (knn_points_normals); metric 2: the in-model
// `knn` on a 3-D cloud (needs xx as well); metric 0: the first three
// channels, squared Euclidean distance by differences (KNN_CUDA; val = sqrt).  x element (b, d, j) at b*sb + d*sd + j*sn;
// output element (b, t, q) at b*o_sb + t*o_sk + q*o_sq.  Writes idx/val of every query the filter settled and a flag
// byte per query (returned through *flag_out) for the exhaustive fallback the caller launches.
int run_knn_normal(int metric, const float *x, long sb, long sd, long sn, const float *xx, int B, int C, int N, int k, int step,
                   long o_sb, long o_sk, long o_sq, int64_t *idx, float *val, void *ws, const unsigned char **flag_out,
                   hipStream_t st) {
  char *base = (char *)ws;
  const int Np = knnn_np(N);
  const size_t n = (size_t)B * N, np = (size_t)B * Np;
  float *rows = (float *)base; base += knnn_align(np * 32);
  float *tau = (float *)base; base += knnn_align(n * 4);
  unsigned char *flag = (unsigned char *)base; base += knnn_align(n);
  unsigned int *bitmap = (unsigned int *)base;
  knnn_prep_kernel<<<dim3(cdiv(Np, 256), B), 256, 0, st>>>(x, xx, rows, N, Np, metric == 1 ? 6 : 3, sb, sd, sn);
  const double mu = (double)k / KNNN_STRIDE;
  int m = (int)(mu + 6.0 * __builtin_sqrt(mu) + 2.0);
  if (m > 96) m = 96;
  const int lds = (N / KNNN_STRIDE) * 32;
  const dim3 gs(cdiv(N, 64), B), gf((Np / 512) * KNNN_CS * B), gr(cdiv(N, 4), B);
#define KNNN_RUN(MV)                                                                                               \
  {                                                                                                                 \
    GCN_HIP(hipFuncSetAttribute((const void *)knnn_sample_kernel<MV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); \
    knnn_sample_kernel<MV><<<gs, 256, lds, st>>>(rows, tau, N, Np, m);                                              \
    knnn_filter_kernel<MV><<<gf, 256, 0, st>>>(rows, tau, bitmap, N, Np, B);                                        \
    knnn_rerank_kernel<MV><<<gr, 256, 0, st>>>(rows, bitmap, flag, idx, val, N, Np, k, step, o_sb, o_sk, o_sq);     \
  }
  if (metric == 1) KNNN_RUN(1) else if (metric == 2) KNNN_RUN(2) else KNNN_RUN(0)
#undef KNNN_RUN
  *flag_out = flag;
  return check_launch("knnn_rerank_kernel");
}

}  // namespace gcn
