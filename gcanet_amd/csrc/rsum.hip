// rsum.hip -- the transposed graph aggregation of the closed-form EdgeConv backward,
//     r[b,m,:] = sum over the edges (n -> m), i.e. idx[b,n,j] == m, of x[b,n,:]      indeg[b,m] = their number,
// as STAGE + SORT + GATHER for the shapes of the model (C in {64,128,256}, N <= 16384), gfx950.
//
// reverse_sum_lds_kernel (edgeconv.hip) lets every destination partition scan the cloud's whole edge list and add
// the matching source rows into LDS with ds_add_u64 -- bound by the LDS atomic rate (3 lane-operations per clock and
// CU: 268 M of them at B=8, N=8192, k=64, C=64 = 0.15 ms) plus the scans: 0.35 ms.  Here no accumulator is shared:
//   1. (absmax_kernel)     S from max|x| and N*k such that no sum of rint(x * 2^S) can overflow 64 bits (integer sums
//                          are order independent: r is bitwise reproducible).
//   2. rsum_file_kernel    one pass over idx: every edge is FILED under its destination partition (R = 16384/C rows):
//                          a per-workgroup LDS counter per partition hands out the slot in that workgroup's segment of
//                          the partition's staging area (plain 4-byte stores {row in partition, source n}); segments
//                          hold twice the mean, the excess of hub graphs goes to a per-cloud overflow list.
//   3. rsum_gather_kernel  one workgroup per (cloud, partition): LDS counting sort of its ~64 R entries by destination
//                          row (histogram = the in-degrees, scan, scatter of the 16-bit source ids), then every wave
//                          takes whole destination rows and GATHERS: 16 source rows in flight and the next 16 already issued,
//                          lane = channel, each f32 converted to 64-bit fixed point by two 32-bit converts (fixed64),
//                          integer adds in registers, one store per element; bound by L2 read bandwidth (256-byte rows:
//                          pre-converted 512-byte integer rows were slower, 0.16 ms vs 0.1).  A partition with more entries than the
//                          LDS list holds, or a cloud with overflow entries, takes the accumulate-in-LDS path instead
//                          (ds_add_u64 as before) -- same result.
#include "common.h"

namespace gcn {

constexpr int RS_SORT_CAP = 32768;           // 16-bit source ids one partition may sort in LDS (64 KB)

struct RsumArgs {
  const float *x;              // (B,N,C)
  const int64_t *idx;          // (B,N,k)
  const unsigned int *absmax;  // max |x| bits (ws header, written by absmax_kernel)
  unsigned int *ovf_cnt;       // (B) zeroed by the host
  int *counts;                 // (B,P,T)
  unsigned int *stag;          // (B,P,T,cap): (row in partition << 16) | n
  unsigned int *ovf;           // (B, N*k): (m << 16) | n
  float *r, *indeg;
  int B, N, C, k, P, T, cap, rshift, tile_rows;
};

__device__ __forceinline__ int rsum_shift(const unsigned int *absmax_bits, int N, int k) {
  const float mx = __uint_as_float(*absmax_bits);
  int ex = 0;
  if (mx > 0.f) (void)frexpf(mx, &ex);                   // mx < 2^ex
  int S = 62 - ex - (64 - __clzll((long long)N * k));
  S = S > 50 - ex ? 50 - ex : S;                         // every |x * 2^S| < 2^50: inside the magic-number window
  return S < 0 ? 0 : (S > 40 ? 40 : S);
}

// Fixed-point conversion by the "magic number" addition: fma(x, 2^S, 1.5 * 2^52) is x * 2^S rounded to the nearest integer
// (ties to even) sitting in the low mantissa bits -- bits(result) - bits(1.5 * 2^52) is that integer for |x * 2^S| < 2^51
// (here < 2^43: S leaves 19 bits for the N*k addends).  The subtraction is done once per sum, not per addend.
constexpr long long RS_MAGIC_BITS = 0x4338000000000000LL;     // bit pattern of 1.5 * 2^52
__device__ __forceinline__ long long magic_bits(float x, double dscale) {
  return __double_as_longlong(fma((double)x, dscale, 6755399441055744.0));
}

// rint(v) of an f32 holding an integer-valued or half-integer-valued product x * 2^S (|v| < 2^62) as a 64-bit integer:
// hi = floor(v / 2^32) and lo = v - hi * 2^32 are both exact in f32 (24 significant bits), so two 32-bit converts do it
__device__ __forceinline__ long long fixed64(float v) {
  const float r = rintf(v);
  const float hi = floorf(r * 2.3283064365386963e-10f);
  const float lo = fmaf(hi, -4294967296.f, r);
  return ((long long)(int)hi << 32) + (long long)(unsigned int)lo;
}

__global__ __launch_bounds__(256) void rsum_file_kernel(RsumArgs a) {
  __shared__ int bcnt[256];                                     // one counter per destination partition (P <= 256)
  bcnt[threadIdx.x] = 0;
  __syncthreads();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int r0 = tile * a.tile_rows, r1 = min(r0 + a.tile_rows, a.N);
  const long e0 = (long)r0 * a.k, e1 = (long)r1 * a.k;
  const int64_t *ib = a.idx + (long)b * a.N * a.k;
  const unsigned int rmask = (1u << a.rshift) - 1u;
  for (long e = e0 + threadIdx.x; e < e1; e += 256) {
    const unsigned int m = (unsigned int)ib[e];
    const unsigned int n = (unsigned int)(e / a.k);
    const int part = (int)(m >> a.rshift);
    const int slot = atomicAdd(&bcnt[part], 1);
    if (slot < a.cap) {
      a.stag[(((long)b * a.P + part) * a.T + tile) * a.cap + slot] = ((m & rmask) << 16) | n;
    } else {
      const unsigned int o = atomicAdd(a.ovf_cnt + b, 1u);
      a.ovf[(long)b * a.N * a.k + o] = (m << 16) | n;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < a.P) a.counts[((long)b * a.P + threadIdx.x) * a.T + tile] = min(bcnt[threadIdx.x], a.cap);
}

template <int CW, int U = 16>   // channels per lane: C = 64 * CW; U source rows in flight (and U more issued)
__global__ __launch_bounds__(1024) void rsum_gather_kernel(RsumArgs a) {
  constexpr int C = 64 * CW;
  constexpr int R = 16384 / C;
  extern __shared__ unsigned long long dyn[];                  // sorted u16 ids (fast path) or [R][C] sums (LDS path)
  __shared__ int hist[256], offs[256], cursor[256];
  __shared__ int total_s;
  const int lane = lane_id(), wave = wave_id();
  const int b = blockIdx.x % a.B, part = blockIdx.x / a.B, m0 = part * R;
  for (int i = threadIdx.x; i < R; i += 1024) hist[i] = 0;
  __syncthreads();
  const long seg0 = ((long)b * a.P + part) * a.T;
  const unsigned int novf = a.ovf_cnt[b];
  // 1. in-degrees of the partition's rows
  for (int t = wave; t < a.T; t += 16) {
    const int cnt = a.counts[seg0 + t];
    const unsigned int *sg = a.stag + (seg0 + t) * a.cap;
    for (int i = lane; i < cnt; i += 64) atomicAdd(&hist[sg[i] >> 16], 1);
  }
  if (novf) {
    const unsigned int *ov = a.ovf + (long)b * a.N * a.k;
    for (unsigned int i = threadIdx.x; i < novf; i += 1024) {
      const unsigned int row = (ov[i] >> 16) - (unsigned int)m0;
      if (row < (unsigned int)R) atomicAdd(&hist[row], 1);
    }
  }
  __syncthreads();
  if (wave == 0) {                                             // exclusive scan of up to 256 counts by one wave
    int run = 0;
    for (int base = 0; base < R; base += 64) {
      const int v = base + lane < R ? hist[base + lane] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
      }
      if (base + lane < R) { offs[base + lane] = run + incl - v; cursor[base + lane] = run + incl - v; }
      run += __shfl(incl, 63);
    }
    if (lane == 0) total_s = run;
  }
  __syncthreads();
  const int total = total_s;
  const int S = rsum_shift(a.absmax, a.N, a.k);
  const double inv = ldexp(1.0, -S);
  const float *xb = a.x + (long)b * a.N * C;
  const float scale = ldexpf(1.f, S);
  const double dscale = ldexp(1.0, S);
  float *rb = a.r + ((long)b * a.N + m0) * C;
  if (a.indeg)
    for (int i = threadIdx.x; i < R; i += 1024) a.indeg[(long)b * a.N + m0 + i] = (float)hist[i];

  if (total <= RS_SORT_CAP) {
    // 2. counting sort of the source ids by destination row
    unsigned short *sorted = reinterpret_cast<unsigned short *>(dyn);
    for (int t = wave; t < a.T; t += 16) {
      const int cnt = a.counts[seg0 + t];
      const unsigned int *sg = a.stag + (seg0 + t) * a.cap;
      for (int i = lane; i < cnt; i += 64) {
        const unsigned int e = sg[i];
        sorted[atomicAdd(&cursor[e >> 16], 1)] = (unsigned short)(e & 0xffffu);
      }
    }
    if (novf) {
      const unsigned int *ov = a.ovf + (long)b * a.N * a.k;
      for (unsigned int i = threadIdx.x; i < novf; i += 1024) {
        const unsigned int e = ov[i], row = (e >> 16) - (unsigned int)m0;
        if (row < (unsigned int)R) sorted[atomicAdd(&cursor[row], 1)] = (unsigned short)(e & 0xffffu);
      }
    }
    __syncthreads();
    // 3. a wave takes whole destination rows: gather the source rows, 16 loads in flight and the next 16 issued before
    //    the current ones are added (the kernel is bound by the latency of these L2 reads, not by their bandwidth)
    for (int row = wave; row < R; row += 16) {
      const int cnt = hist[row], off = offs[row];
      long long acc[CW];
#pragma unroll
      for (int w = 0; w < CW; ++w) acc[w] = 0;
      float cur[U][CW], nxt[U][CW];
      // ids of the row: one LDS read per 64 entries (lane = entry), then a v_readlane per entry
      int ids = 0;
      auto fetch = [&](int e, float (&v)[U][CW]) {
        if ((e & 63) == 0) ids = (int)sorted[off + min(e + lane, cnt - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int n = __builtin_amdgcn_readlane(ids, ((e & 63) + u) & 63);
#pragma unroll
          for (int w = 0; w < CW; ++w) v[u][w] = xb[(unsigned int)n * (unsigned int)C + (unsigned int)(w * 64 + lane)];
        }
      };
      if (cnt > 0) fetch(0, cur);
      for (int e = 0; e < cnt; e += U) {
        if (e + U < cnt) fetch(e + U, nxt);
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (e + u < cnt) {
#pragma unroll
            for (int w = 0; w < CW; ++w) acc[w] += magic_bits(cur[u][w], dscale);
          }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int w = 0; w < CW; ++w) cur[u][w] = nxt[u][w];
      }
      // every addend carried the bit pattern of the magic constant: take cnt of them out again
#pragma unroll
      for (int w = 0; w < CW; ++w) acc[w] -= (long long)cnt * RS_MAGIC_BITS;
#pragma unroll
      for (int w = 0; w < CW; ++w) rb[(long)row * C + w * 64 + lane] = (float)((double)acc[w] * inv);
    }
  } else {
    // LDS path: more entries than the sort list holds (hub destinations): add the rows up in LDS, half the
    // partition's rows at a time (64 KB of 64-bit sums)
    unsigned long long *qacc = dyn;
    constexpr int RH = R / 2;
    for (int half = 0; half < 2; ++half) {
      const unsigned int lo = (unsigned int)(half * RH);
      __syncthreads();
      for (int i = threadIdx.x; i < RH * C; i += 1024) qacc[i] = 0ull;
      __syncthreads();
      auto add_row = [&](unsigned int row, unsigned int n) {
        if (row - lo < (unsigned int)RH) {                       // wave-uniform
#pragma unroll
          for (int w = 0; w < CW; ++w)
            atomicAdd(&qacc[(row - lo) * C + w * 64 + lane], (unsigned long long)fixed64(xb[(long)n * C + w * 64 + lane] * scale));
        }
      };
      for (int t = wave; t < a.T; t += 16) {
        const int cnt = a.counts[seg0 + t];
        const unsigned int *sg = a.stag + (seg0 + t) * a.cap;
        for (int i = 0; i < cnt; ++i) {
          const unsigned int e = sg[i];
          add_row(e >> 16, e & 0xffffu);
        }
      }
      if (novf) {
        const unsigned int *ov = a.ovf + (long)b * a.N * a.k;
        for (unsigned int i = wave; i < novf; i += 16) {
          const unsigned int e = ov[i], row = (e >> 16) - (unsigned int)m0;
          if (row < (unsigned int)R) add_row(row, e & 0xffffu);
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < RH * C; i += 1024) rb[(long)lo * C + i] = (float)((double)(long long)qacc[i] * inv);
    }
  }
}

struct RsumWs { size_t counts, stag, ovf, total; int P, T, cap, tile_rows, rshift; };

static RsumWs rsum_layout(int B, int N, int C, int k) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  RsumWs w{};
  const int R = 16384 / C;
  w.rshift = C == 64 ? 8 : (C == 128 ? 7 : 6);                  // log2 of the R = 16384 / C rows of a partition
  w.P = N / R;
  w.tile_rows = 128;
  w.T = (N + 127) / 128;
  const long mean = (long)w.tile_rows * k / w.P;
  w.cap = (int)(2 * mean < 64 ? 64 : 2 * mean);
  size_t o = 256;                                               // zeroed header: max |x| bits, overflow counters
  w.counts = o; o += al(sizeof(int) * (size_t)B * w.P * w.T);
  w.stag = o; o += al(sizeof(unsigned int) * (size_t)B * w.P * w.T * w.cap);
  w.ovf = o; o += al(sizeof(unsigned int) * (size_t)B * N * k);
  w.total = o;
  return w;
}

bool rsum_staged_supported(int B, int N, int C, int k) {
  return B >= 1 && B <= 60 && (C == 64 || C == 128 || C == 256) && N <= 16384 && N % (16384 / C) == 0 && N / (16384 / C) <= 256 && k >= 1 &&
         (long)N * k < (1L << 31);
}

size_t rsum_staged_ws_bytes(int B, int N, int C, int k) { return rsum_layout(B, N, C, k).total; }

// ws header [0,4) must already hold max |x| bits (absmax_kernel) and [16, 16+4B) zeros
int run_reverse_sum_staged(const float *x, const int64_t *idx, int B, int N, int C, int k, float *r, float *indeg, void *ws,
                           hipStream_t st) {
  const RsumWs w = rsum_layout(B, N, C, k);
  char *base = (char *)ws;
  RsumArgs a{};
  a.x = x; a.idx = idx; a.absmax = (const unsigned int *)base; a.ovf_cnt = (unsigned int *)(base + 16);
  a.counts = (int *)(base + w.counts); a.stag = (unsigned int *)(base + w.stag);
  a.ovf = (unsigned int *)(base + w.ovf); a.r = r; a.indeg = indeg;
  a.B = B; a.N = N; a.C = C; a.k = k; a.P = w.P; a.T = w.T; a.cap = w.cap; a.rshift = w.rshift; a.tile_rows = w.tile_rows;
  rsum_file_kernel<<<dim3(w.T, B), 256, 0, st>>>(a);
  const int ldsb = 8192 * 8;
  if (C == 64) {
    GCN_HIP(hipFuncSetAttribute((const void *)rsum_gather_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    rsum_gather_kernel<1><<<w.P * B, 1024, ldsb, st>>>(a);
  } else if (C == 128) {
    GCN_HIP(hipFuncSetAttribute((const void *)rsum_gather_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    rsum_gather_kernel<2><<<w.P * B, 1024, ldsb, st>>>(a);
  } else {                                                      // C = 256: 8 rows in flight (16 waves share the CU's registers)
    GCN_HIP(hipFuncSetAttribute((const void *)rsum_gather_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    rsum_gather_kernel<4, 8><<<w.P * B, 1024, ldsb, st>>>(a);
  }
  return check_launch("rsum_gather_kernel");
}

}  // namespace gcn
