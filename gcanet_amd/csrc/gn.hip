// gn.hip -- GroupNorm (+ReLU) for POINT-MAJOR activations (B, N, C), forward and backward.
//
// The reference runs its per-point heads as Conv1d(1x1) + GroupNorm + ReLU on (B, C, N) tensors
// (M4:556-603,644-699).  torch's GroupNorm launches one workgroup per (sample, group) for the
// statistics -- 16..64 workgroups on a 256-CU part -- and the channel-major layout forces a transpose
// in front of every row gather.  Here activations stay point-major (one point = one contiguous row,
// which is also what the per-point GEMMs want) and the normalisation is two HBM-bound passes:
//   stats : every workgroup streams a slab of rows with 16-B loads, f64 atomics per (sample, group)
//   apply : y = ReLU((x - mu) * rstd * gamma + beta), 16-B loads/stores
// backward is the same shape: one reduction pass (S1 = sum gamma*g, S2 = sum gamma*g*xhat per
// (sample, group); dgamma, dbeta per channel) and one apply pass.
// dtype 0 = f32, 1 = bf16 (statistics and arithmetic always in f32).
#include "common.h"

#include <type_traits>

namespace gcn {

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned int)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

template <bool BF16>
__device__ __forceinline__ void load4(const void *p, long i, float (&v)[4]) {
  if (BF16) {
    const ushort4 u = *reinterpret_cast<const ushort4 *>(reinterpret_cast<const unsigned short *>(p) + i);
    v[0] = bf2f(u.x); v[1] = bf2f(u.y); v[2] = bf2f(u.z); v[3] = bf2f(u.w);
  } else {
    const float4 f = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p) + i);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
}
template <bool BF16>
__device__ __forceinline__ void store4(void *p, long i, const float (&v)[4]) {
  if (BF16) {
    ushort4 u;
    u.x = f2bf(v[0]); u.y = f2bf(v[1]); u.z = f2bf(v[2]); u.w = f2bf(v[3]);
    *reinterpret_cast<ushort4 *>(reinterpret_cast<unsigned short *>(p) + i) = u;
  } else {
    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// A workgroup owns rows [r0, r1) of one sample.  Thread t owns the 4 channels c4 = (t % (C/4))*4 of
// rows t/(C/4), t/(C/4) + 256/(C/4), ... (C/4 divides 256 or is a multiple of it).
struct Slab {
  int c4, row0, rstep, nc4, reps;  // reps > 1 when C/4 > 256: thread also owns c4 + 1024*i
};
__device__ __forceinline__ Slab make_slab(int C, int nt = 256) {
  Slab s;
  const int q = C / 4;
  if (q <= nt) {
    s.c4 = (threadIdx.x % q) * 4; s.row0 = threadIdx.x / q; s.rstep = nt / q; s.nc4 = q; s.reps = 1;
  } else {
    s.c4 = threadIdx.x * 4; s.row0 = 0; s.rstep = 1; s.nc4 = q; s.reps = (q + nt - 1) / nt;   // callers skip c >= C
  }
  return s;
}

// threads per workgroup of the backward reduction pass: same row slabs, twice the waves of a 256-thread workgroup -- the
// 512 workgroups of a (8, 8192, C) tensor left 1.8 waves per SIMD resident, 72 % of their cycles waiting on memory (SQ
// counters): 31 -> 29 us per launch on average.  The forward statistics pass (one tensor) got slower that way: 256.
constexpr int GN_RT = 512;
constexpr int GN_ST = 256;

template <bool BF16>
__global__ __launch_bounds__(GN_ST) void gn_stats_kernel(const void *__restrict__ x, int N, int C, int G, int rows_per_block,
                                                       double *__restrict__ gsum) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < 2 * G; i += GN_ST) sm[i] = 0.0;
  __syncthreads();
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const Slab s = make_slab(C, GN_ST);
  const int cpg = C / G;
  const int lpg = cpg / 4;
  const bool slots = s.reps == 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0;     // see gn_bwd_reduce_kernel
  for (int rep = 0; rep < s.reps; ++rep) {
    const int c = s.c4 + rep * (GN_ST * 4);
    if (c >= C) break;                                          // C/4 a multiple of 256 but not of GN_ST
    float a1 = 0.f, a2 = 0.f;
    // four rows in flight per thread (8-byte loads: one row at a time leaves the memory system idle -- 3 TB/s)
    int r = r0 + s.row0;
    for (; r + 7 * s.rstep < r1; r += 8 * s.rstep) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) load4<BF16>(x, ((long)b * N + r + u * s.rstep) * C + c, v[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a1 += v[u][i]; a2 = fmaf(v[u][i], v[u][i], a2); }
    }
    for (; r < r1; r += s.rstep) {
      float v[4];
      load4<BF16>(x, ((long)b * N + r) * C + c, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) { a1 += v[i]; a2 = fmaf(v[i], v[i], a2); }
    }
    // 4 consecutive channels share a group (cpg % 4 == 0)
    if (slots) {                 // group sums by butterflies over the lpg lanes of a group, one slot per segment (no LDS atomics)
      double d1 = (double)a1, d2 = (double)a2;
      for (int off = 1; off < lpg; off <<= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
      if ((threadIdx.x % lpg) == 0) {
        sm[2 * G + (threadIdx.x / lpg) * 2] = d1;
        sm[2 * G + (threadIdx.x / lpg) * 2 + 1] = d2;
      }
    } else {
      atomicAdd(&sm[(c / cpg) * 2], (double)a1);
      atomicAdd(&sm[(c / cpg) * 2 + 1], (double)a2);
    }
  }
  __syncthreads();
  if (slots && (int)threadIdx.x < 2 * G) {             // group g = segment index mod G; fixed order
    const int g = threadIdx.x >> 1, j = threadIdx.x & 1, nseg = GN_ST / lpg;
    double t = 0.0;
    for (int q = g; q < nseg; q += G) t += sm[2 * G + q * 2 + j];
    sm[threadIdx.x] = t;          // (threads < 2G only read segment slots and write their own sm[t])
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * G) atomicAdd(gsum + (long)b * G * 2 + threadIdx.x, sm[threadIdx.x]);
}

template <bool BF16>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void *__restrict__ x, const double *__restrict__ gsum,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int N, int C, int G, float eps, int relu, void *__restrict__ y,
                                                       float *__restrict__ mean_rstd) {
  // mean / rstd of the sample's groups once per workgroup (the f64 divide + sqrt per 4 elements made this kernel
  // arithmetic-bound), then 16 bytes per lane and iteration
  __shared__ float mean_s[64], rstd_s[64];
  const int b = blockIdx.y;
  const int cpg = C / G;
  const double cnt = (double)cpg * N;
  for (int g = threadIdx.x; g < G; g += 256) {
    const double m = gsum[((long)b * G + g) * 2] / cnt;
    double var = gsum[((long)b * G + g) * 2 + 1] / cnt - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (g < 64) { mean_s[g] = mean; rstd_s[g] = rstd; }
    if (mean_rstd && blockIdx.x == 0) {
      mean_rstd[((long)b * G + g) * 2] = mean;
      mean_rstd[((long)b * G + g) * 2 + 1] = rstd;
    }
  }
  __syncthreads();
  constexpr int V = BF16 ? 8 : 4;                              // elements per lane and iteration
  const bool wide = (C % V) == 0 && (cpg % V) == 0 && G <= 64;
  if (wide) {
    const long per = (long)N * C / V;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
      const int c = (int)((e * V) % C);
      const int g = c / cpg;
      const float mean = mean_s[g], rstd = rstd_s[g];
#pragma unroll
      for (int h = 0; h < V / 4; ++h) {
        float v[4];
        load4<BF16>(x, (long)b * N * C + e * V + 4 * h, v);
        const float4 ga = *reinterpret_cast<const float4 *>(gamma + c + 4 * h);
        const float4 be = *reinterpret_cast<const float4 *>(beta + c + 4 * h);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float z = (v[i] - mean) * rstd * gg[i] + bb[i];
          v[i] = (relu && !(z > 0.f)) ? 0.f : z;
        }
        store4<BF16>(y, (long)b * N * C + e * V + 4 * h, v);
      }
    }
    return;
  }
  const long per = (long)N * C / 4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
    const int c = (int)((e * 4) % C);
    const int g = c / cpg;
    float mean, rstd;
    if (G <= 64) { mean = mean_s[g]; rstd = rstd_s[g]; }
    else {
      const double m = gsum[((long)b * G + g) * 2] / cnt;
      double var = gsum[((long)b * G + g) * 2 + 1] / cnt - m * m;
      if (var < 0.0) var = 0.0;
      mean = (float)m; rstd = (float)(1.0 / sqrt(var + (double)eps));
    }
    float v[4];
    load4<BF16>(x, (long)b * N * C + e * 4, v);
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 be = *reinterpret_cast<const float4 *>(beta + c);
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float z = (v[i] - mean) * rstd * gg[i] + bb[i];
      v[i] = (relu && !(z > 0.f)) ? 0.f : z;
    }
    store4<BF16>(y, (long)b * N * C + e * 4, v);
  }
}

// apply + max over the points of each sample: out[b,c] = max_n ReLU(GN(x))[b,n,c] and its arg-max row, without
// writing the (B,N,C) activation (M4:510-513: the 1024-channel global feature is only ever used through its max).
// Thread = 4 channels of a row slab (as the backward reduction); best (value, lowest row) per channel goes to a
// u64 atomicMax (value bits << 32 | ~row) -- values are >= 0 after ReLU, or order-mapped when relu == 0.
template <bool BF16>
__global__ __launch_bounds__(256) void gn_apply_max_kernel(const void *__restrict__ x, const double *__restrict__ gsum,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           int N, int C, int G, float eps, int relu, int rows_per_block,
                                                           unsigned long long *__restrict__ best,
                                                           float *__restrict__ mean_rstd) {
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const Slab s = make_slab(C);
  const int cpg = C / G;
  const double cnt = (double)cpg * N;
  for (int rep = 0; rep < s.reps; ++rep) {
    const int c = s.c4 + rep * 1024;
    const int g = c / cpg;
    const double m = gsum[((long)b * G + g) * 2] / cnt;
    double var = gsum[((long)b * G + g) * 2 + 1] / cnt - m * m;
    if (var < 0.0) var = 0.0;
    const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (mean_rstd && blockIdx.x == 0 && s.row0 == 0 && (c % cpg) == 0) {
      mean_rstd[((long)b * G + g) * 2] = mean;
      mean_rstd[((long)b * G + g) * 2 + 1] = rstd;
    }
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 be = *reinterpret_cast<const float4 *>(beta + c);
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    float bv[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    int br[4] = {0, 0, 0, 0};
    auto take = [&](const float (&v)[4], int r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float z = (v[i] - mean) * rstd * gg[i] + bb[i];
        z = (relu && !(z > 0.f)) ? 0.f : z;
        if (BF16) z = bf2f(f2bf(z));                    // the value the (B,N,C) bf16 tensor would have held
        if (z > bv[i]) { bv[i] = z; br[i] = r; }
      }
    };
    int r = r0 + s.row0;
    for (; r + 3 * s.rstep < r1; r += 4 * s.rstep) {    // four rows in flight per thread, consumed in row order
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) load4<BF16>(x, ((long)b * N + r + u * s.rstep) * C + c, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) take(v[u], r + u * s.rstep);
    }
    for (; r < r1; r += s.rstep) {
      float v[4];
      load4<BF16>(x, ((long)b * N + r) * C + c, v);
      take(v, r);
    }
    if (r0 + s.row0 < r1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned int u = __float_as_uint(bv[i] + 0.0f);
        u ^= ((unsigned int)((int)u >> 31) | 0x80000000u);          // order-preserving for any sign
        atomicMax(best + (long)b * C + c + i, ((unsigned long long)u << 32) | (unsigned int)(0xFFFFFFFFu - (unsigned)br[i]));
      }
    }
  }
}

__global__ void gn_max_unpack_kernel(const unsigned long long *__restrict__ best, long n, float *__restrict__ vals,
                                     int64_t *__restrict__ arg) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = best[i];
  unsigned int u = (unsigned int)(k >> 32);
  u ^= ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  vals[i] = __uint_as_float(u);
  arg[i] = (int64_t)(0xFFFFFFFFu - (unsigned int)k);
}

// backward pass 1: S (B,G,2) = [sum gamma*g, sum gamma*g*xhat], dgamma/dbeta (C) (accumulated, pre-zeroed)
template <bool BF16>
__global__ __launch_bounds__(GN_RT) void gn_bwd_reduce_kernel(const void *__restrict__ dy, const void *__restrict__ x,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ mean_rstd, int N, int C, int G,
                                                            int relu, int rows_per_block, double *__restrict__ part_s,
                                                            float *__restrict__ part_c) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < 2 * G; i += GN_RT) sm[i] = 0.0;
  {
    float *cs0 = reinterpret_cast<float *>(sm + 2 * G);
    for (int i = threadIdx.x; i < 2 * C; i += GN_RT) cs0[i] = 0.f;
  }
  __syncthreads();
  const int b = blockIdx.y;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const Slab s = make_slab(C, GN_RT);
  const int cpg = C / G;
  const int lpg = cpg / 4;                                      // lanes (4-channel groups) per GroupNorm group
  // slot scheme: one row-lane tile slot per thread, group sums by wave butterflies (needs one channel group per thread and
  // a power-of-two lpg <= 64, true for every layer of the model; the host sizes the LDS accordingly)
  const bool slots = s.reps == 1 && lpg <= 64 && (lpg & (lpg - 1)) == 0;
  for (int rep = 0; rep < s.reps; ++rep) {
    const int c = s.c4 + rep * (GN_RT * 4);
    if (c >= C) break;                                          // C/4 a multiple of 256 but not of GN_RT
    const int g = c / cpg;
    const float mean = mean_rstd[((long)b * G + g) * 2], rstd = mean_rstd[((long)b * G + g) * 2 + 1];
    const float4 ga = *reinterpret_cast<const float4 *>(gamma + c);
    const float4 be = *reinterpret_cast<const float4 *>(beta + c);
    const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    float s1 = 0.f, s2 = 0.f;
    auto accum = [&](const float (&xv)[4], const float (&gv)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xh = (xv[i] - mean) * rstd;
        const float z = xh * gg[i] + bb[i];
        const float gz = (relu && !(z > 0.f)) ? 0.f : gv[i];
        dg[i] = fmaf(gz, xh, dg[i]);
        db[i] += gz;
        s1 = fmaf(gg[i], gz, s1);
        s2 = fmaf(gg[i] * gz, xh, s2);
      }
    };
    int r = r0 + s.row0;
    for (; r + 7 * s.rstep < r1; r += 8 * s.rstep) {           // eight rows (sixteen loads) in flight per thread
      float xv[8][4], gv[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long o = ((long)b * N + r + u * s.rstep) * C + c;
        load4<BF16>(x, o, xv[u]);
        load4<BF16>(dy, o, gv[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) accum(xv[u], gv[u]);
    }
    for (; r < r1; r += s.rstep) {
      float xv[4], gv[4];
      const long o = ((long)b * N + r) * C + c;
      load4<BF16>(x, o, xv);
      load4<BF16>(dy, o, gv);
      accum(xv, gv);
    }
    if (slots) {
      // every thread owns one slot of the [row lane][2][C] tile (16-byte stores, no atomics: ds_add_f32 from 512 threads
      // x 8 values was 10 of the kernel's 29 us); the group sums meet by butterflies over the lpg lanes of a group
      float *tile = reinterpret_cast<float *>(sm + 2 * G + 2 * GN_RT);      // [RL][2][C]
      const int rl = threadIdx.x / s.nc4;
      *reinterpret_cast<float4 *>(tile + ((long)rl * 2) * C + c) = make_float4(dg[0], dg[1], dg[2], dg[3]);
      *reinterpret_cast<float4 *>(tile + ((long)rl * 2 + 1) * C + c) = make_float4(db[0], db[1], db[2], db[3]);
      double d1 = (double)s1, d2 = (double)s2;
      for (int off = 1; off < lpg; off <<= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
      if ((threadIdx.x % lpg) == 0) {
        double *seg = sm + 2 * G;                                            // [GN_RT / lpg][2]  (<= 2 GN_RT doubles)
        seg[(threadIdx.x / lpg) * 2] = d1;
        seg[(threadIdx.x / lpg) * 2 + 1] = d2;
      }
    } else {
      // generic shapes: combine the row slices of the workgroup with LDS atomics
      float *cs = reinterpret_cast<float *>(sm + 2 * G);   // [2][C]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        atomicAdd(&cs[c + i], dg[i]);
        atomicAdd(&cs[C + c + i], db[i]);
      }
      atomicAdd(&sm[g * 2], (double)s1);
      atomicAdd(&sm[g * 2 + 1], (double)s2);
    }
  }
  __syncthreads();
  // per-workgroup partials, folded by fold_partials_kernel (common.h): 512 workgroups adding to the same 2C + 2G addresses
  // cost ~25 us of contention per launch, whatever the tensor size (and made the sums order-dependent)
  const long blk = (long)b * gridDim.x + blockIdx.x;
  if (slots) {
    const int RL = GN_RT / s.nc4, nseg = GN_RT / lpg;
    if ((int)threadIdx.x < 2 * G) {                     // group g = segment index mod G; fixed order
      const double *seg = sm + 2 * G;
      const int g = threadIdx.x >> 1, j = threadIdx.x & 1;
      double t = 0.0;
      for (int q = g; q < nseg; q += G) t += seg[q * 2 + j];
      part_s[blk * 2 * G + threadIdx.x] = t;
    }
    const float *tile = reinterpret_cast<const float *>(sm + 2 * G + 2 * GN_RT);
    for (int i = threadIdx.x; i < 2 * C; i += GN_RT) {
      const int which = i / C, c = i - which * C;
      float t = 0.f;
      for (int rl = 0; rl < RL; ++rl) t += tile[((long)rl * 2 + which) * C + c];
      part_c[blk * 2 * C + i] = t;
    }
    return;
  }
  if ((int)threadIdx.x < 2 * G) part_s[blk * 2 * G + threadIdx.x] = sm[threadIdx.x];
  {
    const float *cs = reinterpret_cast<const float *>(sm + 2 * G);
    for (int i = threadIdx.x; i < 2 * C; i += GN_RT) part_c[blk * 2 * C + i] = cs[i];
  }
}


// backward pass 2: dx = rstd * (gamma*g - S1/M - xhat*S2/M)
template <bool BF16>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const void *__restrict__ dy, const void *__restrict__ x,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ mean_rstd, const double *__restrict__ S,
                                                           int N, int C, int G, int relu, void *__restrict__ dx) {
  const int b = blockIdx.y;
  const int cpg = C / G;
  const float invM = 1.f / ((float)cpg * (float)N);
  constexpr int V = BF16 ? 8 : 4;                              // 16 bytes per lane and iteration
  const bool wide = (C % V) == 0 && (cpg % V) == 0;
  auto body = [&](auto wc) {
    constexpr int W = decltype(wc)::value;
    const long per = (long)N * C / W;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
      const int c = (int)((e * W) % C);
      const int g = c / cpg;
      const float mean = mean_rstd[((long)b * G + g) * 2], rstd = mean_rstd[((long)b * G + g) * 2 + 1];
      const float m1 = (float)S[((long)b * G + g) * 2] * invM, m2 = (float)S[((long)b * G + g) * 2 + 1] * invM;
      float xv[W / 4][4], gv[W / 4][4];
#pragma unroll
      for (int h = 0; h < W / 4; ++h) {
        load4<BF16>(x, (long)b * N * C + e * W + 4 * h, xv[h]);
        load4<BF16>(dy, (long)b * N * C + e * W + 4 * h, gv[h]);
      }
#pragma unroll
      for (int h = 0; h < W / 4; ++h) {
        const float4 ga = *reinterpret_cast<const float4 *>(gamma + c + 4 * h);
        const float4 be = *reinterpret_cast<const float4 *>(beta + c + 4 * h);
        const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float xh = (xv[h][i] - mean) * rstd;
          const float z = xh * gg[i] + bb[i];
          const float gz = (relu && !(z > 0.f)) ? 0.f : gv[h][i];
          o[i] = rstd * (gg[i] * gz - m1 - xh * m2);
        }
        store4<BF16>(dx, (long)b * N * C + e * W + 4 * h, o);
      }
    }
  };
  if (wide) body(std::integral_constant<int, V>{});
  else body(std::integral_constant<int, 4>{});
}

// ---------------------------------------------------------------- backward of  max_n [ReLU](GroupNorm(x))
// The gradient of the pooled output reaches ONE row per (sample, channel): dy is zero except at (b, arg[b,c], c).
// GroupNorm's input gradient is then  dx = rstd*gamma*g*[n == arg] + A[b,g] + Bx[b,g]*x  with per-(sample, group)
// constants -- the two group sums run over B*C values, not B*N*C.  Three launches: the sums (tiny), the dense affine
// map (reads x, writes dx: two tensor passes instead of a zero fill, a scatter and the five passes of the generic
// reduce + apply), and the B*C sparse corrections.
template <bool BF16>
__device__ __forceinline__ float load1(const void *p, long i) {
  return BF16 ? bf2f(reinterpret_cast<const unsigned short *>(p)[i]) : reinterpret_cast<const float *>(p)[i];
}

template <bool BF16>
__global__ __launch_bounds__(1024) void gn_max_bwd_sums_kernel(const void *__restrict__ x, const float *__restrict__ gamma,
                                                               const float *__restrict__ beta, const float *__restrict__ mean_rstd,
                                                               const float *__restrict__ dout, const int64_t *__restrict__ arg,
                                                               int B, int N, int C, int G, int relu, float *__restrict__ AB,
                                                               float *__restrict__ sp, float *__restrict__ dgamma,
                                                               float *__restrict__ dbeta) {
  __shared__ double p1[1024], p2[1024];
  const int cpg = C / G;
  const double M = (double)cpg * (double)N;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int c = c0 + threadIdx.x;
    const bool live = c < C;
    const float ga = live ? gamma[c] : 0.f, be = live ? beta[c] : 0.f;
    const int g = live ? c / cpg : 0;
    float dg = 0.f, db = 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {
      // the kernel is one workgroup of dependent gathers: issue the loads of eight samples before using any
      long nrow[8];
      float xv8[8], go8[8], mean8[8], rstd8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = min(b0 + u, B - 1);
        nrow[u] = live ? arg[(long)bb * C + c] : 0;
        go8[u] = live ? dout[(long)bb * C + c] : 0.f;
        mean8[u] = mean_rstd[((long)bb * G + g) * 2];
        rstd8[u] = mean_rstd[((long)bb * G + g) * 2 + 1];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = min(b0 + u, B - 1);
        xv8[u] = live ? load1<BF16>(x, ((long)bb * N + nrow[u]) * C + c) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + u;
        if (b >= B) break;                                       // block-uniform
        double s1 = 0.0, s2 = 0.0;
        if (live) {
          const float mean = mean8[u], rstd = rstd8[u];
          const float xh = (xv8[u] - mean) * rstd;
          const float z = xh * ga + be;
          const float gz = (relu && !(z > 0.f)) ? 0.f : go8[u];
          dg = fmaf(gz, xh, dg);
          db += gz;
          s1 = (double)(ga * gz);
          s2 = (double)(ga * gz) * (double)xh;
          sp[(long)b * C + c] = rstd * ga * gz;
        }
        // group sums in a fixed order (deterministic): whole waves per group -> butterfly inside the wave, then one
        // thread per group adds the wave partials; otherwise one thread per group walks its channels
        const bool by_wave = cpg % 64 == 0;
        if (by_wave) {
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
          }
          if (lane_id() == 0) { p1[wave_id()] = s1; p2[wave_id()] = s2; }
        } else {
          p1[threadIdx.x] = s1;
          p2[threadIdx.x] = s2;
        }
        __syncthreads();
        const int g0 = c0 / cpg, g1 = min(G, (min(C, c0 + 1024) + cpg - 1) / cpg);
        if ((int)threadIdx.x < g1 - g0) {
          const int gg = g0 + threadIdx.x;
          int lo = max(gg * cpg, c0) - c0, hi = min((gg + 1) * cpg, c0 + 1024) - c0;
          if (by_wave) { lo /= 64; hi /= 64; }
          double a1 = 0.0, a2 = 0.0;
          for (int i = lo; i < hi; ++i) { a1 += p1[i]; a2 += p2[i]; }
          // a group may straddle chunks (cpg > 1024): accumulate in AB as raw sums first (f32 pairs hold them below)
          double *acc = reinterpret_cast<double *>(AB) + ((long)b * G + gg) * 2;   // AB doubles as (B,G,2) f64 scratch
          if (max(gg * cpg, c0) == gg * cpg) { acc[0] = a1; acc[1] = a2; } else { acc[0] += a1; acc[1] += a2; }
          if (min((gg + 1) * cpg, c0 + 1024) == (gg + 1) * cpg) {               // group complete: the affine constants
            const double mu = (double)mean_rstd[((long)b * G + gg) * 2], rs = (double)mean_rstd[((long)b * G + gg) * 2 + 1];
            const double Bg = (-(rs * rs)) * acc[1] / M;
            const double Ag = (-(rs * acc[0])) / M - Bg * mu;
            float *o = AB + ((long)B * G * 2) * 2 + ((long)b * G + gg) * 2;       // floats behind the f64 scratch
            o[0] = (float)Ag;
            o[1] = (float)Bg;
          }
        }
        __syncthreads();
      }
    }
    if (live) { dgamma[c] = dg; dbeta[c] = db; }
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void gn_max_bwd_dense_kernel(const void *__restrict__ x, const float *__restrict__ ab, int N,
                                                               int C, int G, void *__restrict__ dx) {
  const int b = blockIdx.y;
  const long per = (long)N * C / 4;
  const int cpg = C / G;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
    const int g = (int)((e * 4) % C) / cpg;
    const float A = ab[((long)b * G + g) * 2], Bx = ab[((long)b * G + g) * 2 + 1];
    float xv[4], o[4];
    load4<BF16>(x, (long)b * N * C + e * 4, xv);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(Bx, xv[i], A);
    store4<BF16>(dx, (long)b * N * C + e * 4, o);
  }
}

template <bool BF16>
__global__ void gn_max_bwd_sparse_kernel(const void *__restrict__ x, const float *__restrict__ ab, const float *__restrict__ sp,
                                         const int64_t *__restrict__ arg, int B, int N, int C, int G, void *__restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C) return;
  const int b = (int)(i / C), c = (int)(i % C), g = c / (C / G);
  const long o = ((long)b * N + arg[i]) * C + c;
  const float v = fmaf(ab[((long)b * G + g) * 2 + 1], load1<BF16>(x, o), ab[((long)b * G + g) * 2]) + sp[i];
  if (BF16) reinterpret_cast<unsigned short *>(dx)[o] = f2bf(v);
  else reinterpret_cast<float *>(dx)[o] = v;
}

static int gn_check(const char *who, int B, int N, int C, int G, int dtype) {
  GCN_REQUIRE(dtype == 0 || dtype == 1, "%s: dtype must be 0 (f32) or 1 (bf16)", who);
  GCN_REQUIRE(B >= 0 && N >= 1 && C >= 4 && G >= 1 && C % G == 0, "%s: bad shape", who);
  GCN_REQUIRE((C / G) % 4 == 0, "%s: C/G must be a multiple of 4", who);
  GCN_REQUIRE((C / 4 <= 256 && 256 % (C / 4) == 0) || (C / 4) % 256 == 0, "%s: C=%d unsupported (C/4 must divide 256 or be a multiple of it)", who, C);
  return GCN_OK;
}

// dynamic LDS of gn_bwd_reduce_kernel: [2G] f64 group sums, [2 GN_RT] f64 segment sums (2 per lpg lanes), then the larger
// of the slot tile (GN_RT x 8 f32) and the atomic path's [2][C] f32
static size_t gn_reduce_lds(int C, int G) {
  const size_t fl = (size_t)GN_RT * 8 > (size_t)2 * C ? (size_t)GN_RT * 8 : (size_t)2 * C;
  return sizeof(double) * (2 * (size_t)G + 2 * GN_RT) + sizeof(float) * fl;
}

static int slab_rows(int N, int B) {
  int blocks = (512 + B - 1) / B;  // ~2 workgroups per CU in total
  int rows = (N + blocks - 1) / blocks;
  return rows < 8 ? 8 : rows;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_gn_fwd(const void *x, int dtype, const float *gamma, const float *beta, int B, int N, int C, int G,
                          float eps, int relu, void *y, float *mean_rstd, double *gsum_ws, void *stream) {
  int rc = gn_check("gcn_gn_fwd", B, N, C, G, dtype);
  if (rc) return rc;
  GCN_REQUIRE(x && gamma && beta && y && gsum_ws, "gcn_gn_fwd: null pointer");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(gsum_ws, sizeof(double) * 2 * B * G, st));
  const int rows = slab_rows(N, B);
  const dim3 g1(cdiv(N, rows), B);
  const int g2 = (int)(((long)N * C / 4 + 255) / 256 > 2048 ? 2048 : ((long)N * C / 4 + 255) / 256);
  if (dtype == 1) {
    gn_stats_kernel<true><<<g1, GN_ST, sizeof(double) * (2 * G + 2 * GN_ST), st>>>(x, N, C, G, rows, gsum_ws);
    gn_apply_kernel<true><<<dim3(g2, B), 256, 0, st>>>(x, gsum_ws, gamma, beta, N, C, G, eps, relu, y, mean_rstd);
  } else {
    gn_stats_kernel<false><<<g1, GN_ST, sizeof(double) * (2 * G + 2 * GN_ST), st>>>(x, N, C, G, rows, gsum_ws);
    gn_apply_kernel<false><<<dim3(g2, B), 256, 0, st>>>(x, gsum_ws, gamma, beta, N, C, G, eps, relu, y, mean_rstd);
  }
  return check_launch("gn_fwd");
}

GCN_EXPORT int gcn_gn_apply(const void *x, int dtype, const double *gsum, const float *gamma, const float *beta, int B,
                            int N, int C, int G, float eps, int relu, void *y, float *mean_rstd, void *stream) {
  int rc = gn_check("gcn_gn_apply", B, N, C, G, dtype);
  if (rc) return rc;
  GCN_REQUIRE(x && gsum && gamma && beta && y, "gcn_gn_apply: null pointer");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const int g2 = (int)(((long)N * C / 4 + 255) / 256 > 2048 ? 2048 : ((long)N * C / 4 + 255) / 256);
  if (dtype == 1) gn_apply_kernel<true><<<dim3(g2, B), 256, 0, st>>>(x, gsum, gamma, beta, N, C, G, eps, relu, y, mean_rstd);
  else gn_apply_kernel<false><<<dim3(g2, B), 256, 0, st>>>(x, gsum, gamma, beta, N, C, G, eps, relu, y, mean_rstd);
  return check_launch("gn_apply_kernel");
}

GCN_EXPORT long gcn_gn_bwd_ws_bytes(int B, int N, int C, int G) {
  if (B < 0 || N < 1 || C < 1 || G < 1) return -1;
  const long nblk = cdiv(N, slab_rows(N, B > 0 ? B : 1));
  return 8L * (2L * B * G) * (1 + nblk) + 4L * 2 * C * B * nblk;
}

GCN_EXPORT int gcn_gn_bwd(const void *dy, const void *x, int dtype, const float *gamma, const float *beta,
                          const float *mean_rstd, int B, int N, int C, int G, int relu, void *dx, float *dgamma,
                          float *dbeta, double *s_ws, void *stream) {
  int rc = gn_check("gcn_gn_bwd", B, N, C, G, dtype);
  if (rc) return rc;
  GCN_REQUIRE(dy && x && gamma && beta && mean_rstd && dx && dgamma && dbeta && s_ws, "gcn_gn_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) { GCN_HIP(zero_spans(st, {dgamma, sizeof(float) * C}, {dbeta, sizeof(float) * C})); return GCN_OK; }
  GCN_REQUIRE(((uintptr_t)s_ws & 7) == 0, "gcn_gn_bwd: s_ws must be 8-byte aligned");
  const int rows = slab_rows(N, B);
  const int nblk = cdiv(N, rows);
  double *S = s_ws;                                             // (B,G,2), then the per-workgroup partials
  double *part_s = S + 2L * B * G;
  float *part_c = reinterpret_cast<float *>(part_s + 2L * G * B * nblk);
  const dim3 g1(nblk, B);
  const int g2 = (int)(((long)N * C / 4 + 255) / 256 > 2048 ? 2048 : ((long)N * C / 4 + 255) / 256);
  if (dtype == 1) {
    gn_bwd_reduce_kernel<true><<<g1, GN_RT, gn_reduce_lds(C, G), st>>>(dy, x, gamma, beta, mean_rstd, N, C, G, relu, rows, part_s, part_c);
    fold_partials_kernel<<<cdiv(2 * C, 64) + cdiv(B * 2 * G, 64), 256, 0, st>>>(part_s, part_c, nblk, B, C, G, S, dgamma, dbeta);
    gn_bwd_apply_kernel<true><<<dim3(g2, B), 256, 0, st>>>(dy, x, gamma, beta, mean_rstd, S, N, C, G, relu, dx);
  } else {
    gn_bwd_reduce_kernel<false><<<g1, GN_RT, gn_reduce_lds(C, G), st>>>(dy, x, gamma, beta, mean_rstd, N, C, G, relu, rows, part_s, part_c);
    fold_partials_kernel<<<cdiv(2 * C, 64) + cdiv(B * 2 * G, 64), 256, 0, st>>>(part_s, part_c, nblk, B, C, G, S, dgamma, dbeta);
    gn_bwd_apply_kernel<false><<<dim3(g2, B), 256, 0, st>>>(dy, x, gamma, beta, mean_rstd, S, N, C, G, relu, dx);
  }
  return check_launch("gn_bwd");
}

GCN_EXPORT int gcn_gn_max_fwd(const void *x, int dtype, const float *gamma, const float *beta, int B, int N, int C, int G,
                              float eps, int relu, float *out_max, int64_t *out_arg, float *mean_rstd, double *gsum_ws,
                              void *best_ws, void *stream) {
  int rc = gn_check("gcn_gn_max_fwd", B, N, C, G, dtype);
  if (rc) return rc;
  GCN_REQUIRE(x && gamma && beta && out_max && out_arg && gsum_ws && best_ws, "gcn_gn_max_fwd: null pointer");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(gsum_ws, sizeof(double) * 2 * B * G, st));
  GCN_HIP(zero_dev(best_ws, sizeof(unsigned long long) * (size_t)B * C, st));
  const int rows = slab_rows(N, B);
  const dim3 g1(cdiv(N, rows), B);
  unsigned long long *best = (unsigned long long *)best_ws;
  if (dtype == 1) {
    gn_stats_kernel<true><<<g1, GN_ST, sizeof(double) * (2 * G + 2 * GN_ST), st>>>(x, N, C, G, rows, gsum_ws);
    gn_apply_max_kernel<true><<<g1, 256, 0, st>>>(x, gsum_ws, gamma, beta, N, C, G, eps, relu, rows, best, mean_rstd);
  } else {
    gn_stats_kernel<false><<<g1, GN_ST, sizeof(double) * (2 * G + 2 * GN_ST), st>>>(x, N, C, G, rows, gsum_ws);
    gn_apply_max_kernel<false><<<g1, 256, 0, st>>>(x, gsum_ws, gamma, beta, N, C, G, eps, relu, rows, best, mean_rstd);
  }
  gn_max_unpack_kernel<<<cdiv((long)B * C, 256), 256, 0, st>>>(best, (long)B * C, out_max, out_arg);
  return check_launch("gn_max_fwd");
}

GCN_EXPORT long gcn_gn_max_bwd_ws_floats(int B, int C, int G) {
  if (B < 0 || C < 1 || G < 1) return -1;
  return 6L * B * G + (long)B * C;             // (B,G,2) f64 sums, (B,G,2) f32 constants, (B,C) f32 sparse terms
}

GCN_EXPORT int gcn_gn_max_bwd(const void *x, int dtype, const float *gamma, const float *beta, const float *mean_rstd,
                              const float *dout, const int64_t *arg, int B, int N, int C, int G, int relu, void *dx,
                              float *dgamma, float *dbeta, float *ws, void *stream) {
  int rc = gn_check("gcn_gn_max_bwd", B, N, C, G, dtype);
  if (rc) return rc;
  GCN_REQUIRE(x && gamma && beta && mean_rstd && dout && arg && dx && dgamma && dbeta && ws, "gcn_gn_max_bwd: null pointer");
  GCN_REQUIRE(((uintptr_t)ws & 7) == 0, "gcn_gn_max_bwd: ws must be 8-byte aligned");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  float *ab = ws + 4L * B * G;
  float *sp = ws + 6L * B * G;
  const int g2 = (int)(((long)N * C / 4 + 255) / 256 > 2048 ? 2048 : ((long)N * C / 4 + 255) / 256);
  const int g3 = (int)cdiv((long)B * C, 256);
  if (dtype == 1) {
    gn_max_bwd_sums_kernel<true><<<1, 1024, 0, st>>>(x, gamma, beta, mean_rstd, dout, arg, B, N, C, G, relu, ws, sp, dgamma, dbeta);
    gn_max_bwd_dense_kernel<true><<<dim3(g2, B), 256, 0, st>>>(x, ab, N, C, G, dx);
    gn_max_bwd_sparse_kernel<true><<<g3, 256, 0, st>>>(x, ab, sp, arg, B, N, C, G, dx);
  } else {
    gn_max_bwd_sums_kernel<false><<<1, 1024, 0, st>>>(x, gamma, beta, mean_rstd, dout, arg, B, N, C, G, relu, ws, sp, dgamma, dbeta);
    gn_max_bwd_dense_kernel<false><<<dim3(g2, B), 256, 0, st>>>(x, ab, N, C, G, dx);
    gn_max_bwd_sparse_kernel<false><<<g3, 256, 0, st>>>(x, ab, sp, arg, B, N, C, G, dx);
  }
  return check_launch("gn_max_bwd");
}
