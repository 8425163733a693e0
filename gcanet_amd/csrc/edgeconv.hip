// edgeconv.hip -- fused DGCNN EdgeConv block for gfx950: neighbour gather -> grouped
// (N*k, 2C) x (2C, Cout) contraction on MFMA -> per-point max/min over k + GroupNorm
// statistics, without ever materialising the (B,2C,N,k) edge tensor or the (B,Cout,N,k)
// activation (reference: get_graph_feature M4:93-124 + Conv2d 1x1 + GroupNorm + LeakyReLU +
// max, models/dgcnn-hais-concat-direct-4.py:463-505; its edge tensor alone is 4.3 GB at
// B=8, N=8192, k=64, C=128).
//
// Formulation.  The reference's row is e = [x_j - x_i ; x_i] with weight W = [W1 | W2].  The
// kernel contracts the SAME (N*k, 2C) row count against the re-parameterised weight
// W' = [W1 | W2 - W1] with rows a = [x_j ; x_i]:  W.e == W'.a in exact arithmetic.  That
// keeps the full grouped contraction on the matrix cores while making the A operand a pure
// row gather (no VALU work per element), so rows go HBM/L2 -> LDS by LDS-DMA.
//
// Monotonicity.  y -> LeakyReLU(gamma*(y-mu)*rstd+beta) is monotone in y, so
// max_k f(y_k) = f(max_k y_k) for gamma >= 0 and f(min_k y_k) for gamma < 0: the kernel keeps
// per-(point, channel) max and min of the RAW conv output plus per-(cloud, group) sum and
// sum of squares; gcn_edgeconv_finish applies GroupNorm + LeakyReLU to the routed extreme.
//
// bf16 kernel geometry (wave64, MFMA 32x32x16 bf16):
//   tile      = TP points x kp rows (kp = k rounded up to 32; padded slots repeat neighbour 0,
//               which cannot change max/min and is masked out of the sums)
//   wave tile = 64 rows x 64 cols (2x2 MFMA blocks); B fragments (W') live in REGISTERS for
//               the whole kernel, so the LDS only streams A
//   A in LDS  = [rows][Cp] bf16, 16-B chunks XOR-swizzled (on the DMA *source* side) so the
//               ds_read_b128 fragment reads are bank-conflict-free; double buffered: tile
//               t+1 is gathered by global_load_lds_dwordx4 while tile t is on the MFMAs
//   centre rows x_i are staged once per point and read as LDS broadcasts.
#include <algorithm>

#include "common.h"
#include "edgeconv_fwd.h"

namespace gcn {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  // round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32)
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// bf16 (half == 0) or IEEE half (half == 1) image of an f32 value, as the 16-bit pattern the MFMA operand holds
__device__ __forceinline__ unsigned short f32_to_16(float v, int half) {
  if (half) {
    const _Float16 h = (_Float16)v;                        // round to nearest even; |v| > 65504 becomes inf, as torch's .half()
    return __builtin_bit_cast(unsigned short, h);
  }
  return f32_to_bf16(v);
}

// ------------------------------------------------------------------ operand packing
// x (B,C,N) f32 channel-major -> x_bf (B,N,Cp) bf16 zero padded and/or x_f32 (B,N,C)
__global__ __launch_bounds__(256) void pack_x_kernel(const float *__restrict__ x, int C, int N, int Cp,
                                                     unsigned short *__restrict__ x_bf, float *__restrict__ x_f32, int half) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, n = n0 + tx;
    tile[i][tx] = (c < C && n < N) ? x[((long)b * C + c) * N + n] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    if (n < N) {
      const float v = tile[tx][i];
      if (x_bf && c < Cp) x_bf[((long)b * N + n) * Cp + c] = f32_to_16(v, half);
      if (x_f32 && c < C) x_f32[((long)b * N + n) * C + c] = v;
    }
  }
}


// x (R, C) f32 row-major -> (R, Cp) bf16 zero padded (point-major operand of the fused kernel)
__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float *__restrict__ x, long R, int C, int Cp,
                                                            unsigned short *__restrict__ y, int half) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= R * Cp) return;
  const long r = e / Cp;
  const int c = (int)(e % Cp);
  y[e] = c < C ? f32_to_16(x[r * C + c], half) : (unsigned short)0;
}

// w (Cout, 2C) f32 -> wp (Cout, 2Cp) bf16 = [W1 | 0 | W2 - W1 | 0]
__global__ void pack_w_kernel(const float *__restrict__ w, int Cout, int C, int Cp, unsigned short *__restrict__ wp, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cout * 2 * Cp) return;
  const int co = i / (2 * Cp), kk = i % (2 * Cp);
  float v = 0.f;
  if (kk < C) v = w[(long)co * 2 * C + kk];
  else if (kk >= Cp && kk < Cp + C) v = w[(long)co * 2 * C + C + (kk - Cp)] - w[(long)co * 2 * C + (kk - Cp)];
  wp[i] = f32_to_16(v, half);
}

// ------------------------------------------------------------------ fused forward (bf16 MFMA)
struct EcArgs {
  const unsigned short *x;  // (B,N,Cp) bf16
  const unsigned short *wp; // (Cout, 2Cp) bf16
  const int64_t *idx;       // (B,N,k)
  int B, N, NX, k, kp, Cout, G, TP;  // NX = rows of x per cloud (== N for EdgeConv)
  int tiles_per_cloud, total_tiles;
  float *ymax, *ymin;       // (B,N,Cout)
  unsigned char *amax, *amin;
  double *gsum;             // (B,G,2)
  const float *gamma_route; // non-null: keep only the extreme GroupNorm+LeakyReLU will route (max if gamma>=0 else min)
};

// KSTEPS = 2*Cp/16 (= Cp/8 = 16-B chunks per row), CW = Cout/32 column groups,
// RWT = 128-row groups per tile; block = 64*RWT*CW threads.  A wave owns 128 rows x 32 cols
// (4 MFMA blocks): 64 VGPRs of B fragments + 64 accumulators, so two waves fit per SIMD.
//
// RWT == 1 (k <= 128, the normal case): the tile is ONE wave-row-group, every point's rows live in a
// single wave, so max/min/arg are combined in registers and stored straight from the epilogue -- one
// barrier per tile, no LDS round trip.  Workgroups are small (CW waves, <= 66 KB LDS) so that TWO of
// them share a CU and drift out of phase: one workgroup's VALU epilogue overlaps the other's MFMAs.
// The neighbour ids of tile t+2 are fetched while tile t computes, so the DMA issue of tile t+1 never
// waits on a dependent global load.  RWT == 2 (128 < k <= 255) keeps the cross-wave LDS combine.
template <int KSTEPS, int CW, int RWT, bool WITH_ARG, bool ROUTED, int KP>   // KP: compile-time padded k (0 = runtime)
__global__ __launch_bounds__(64 * RWT * CW, 2) void edgeconv_fwd_bf16_kernel(EcArgs a) {
  constexpr int NC = KSTEPS;               // chunks per x row
  constexpr int CP = KSTEPS * 8;           // padded channels
  constexpr int K = 2 * CP;
  constexpr int ROW_BYTES = CP * 2;
  constexpr int TILE_ROWS = RWT * 128;
  constexpr int NW = RWT * CW;
  constexpr int COUT = CW * 32;
  constexpr int RPP = 64 / NC;             // rows per 1-KiB DMA piece
  constexpr int PIECES = TILE_ROWS / RPP;
  constexpr int PPW = (PIECES + NW - 1) / NW;
  constexpr int RPB = NC >= 16 ? 1 : 16 / NC;  // rows per 256-B bank row
  constexpr int A_BYTES = TILE_ROWS * ROW_BYTES;
  constexpr int MAXTP = TILE_ROWS / 32;
  constexpr int C_BYTES = ((MAXTP * ROW_BYTES + 1023) / 1024) * 1024;
  constexpr int CPIECES = C_BYTES / 1024;
  constexpr int BUF_BYTES = A_BYTES + C_BYTES;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int lane = lane_id();
  const int wave = wave_id();
  const int rg = wave / CW, cg = wave % CW;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- B fragments (W') in registers for the whole kernel
  bf16x8 breg[KSTEPS];
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s)
    breg[s] = *reinterpret_cast<const bf16x8 *>(a.wp + (long)(cg * 32 + lr) * K + s * 16 + lh * 8);

  // routed mode: sgn = +1 keeps the max, -1 turns the min into a max of the NEGATED column: the sign is folded
  // into this lane's W' column once (bf16 sign bits; products and sums negate exactly), not into every output
  const float sgn = ROUTED ? (a.gamma_route[cg * 32 + lr] >= 0.f ? 1.f : -1.f) : 1.f;
  if (ROUTED && sgn < 0.f) {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
      unsigned int *w4 = reinterpret_cast<unsigned int *>(&breg[s]);
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] ^= 0x80008000u;
    }
  }

  // Workgroup g runs on XCD g % 8.  Give the workgroups of one XCD CONSECUTIVE tile ranges, i.e. (with 8 clouds) one
  // cloud per XCD: its point rows (1-2 MB) then stay in that XCD's 4 MB L2 across the k-fold gathers instead of every
  // L2 seeing every cloud.
  const int G = gridDim.x;
  const int vb = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int t_begin = (int)((long)vb * a.total_tiles / G);
  const int t_end = (int)((long)(vb + 1) * a.total_tiles / G);
  if (t_begin >= t_end) return;

  const int kp = KP ? KP : a.kp, k = a.k, TP = a.TP;      // a constant kp turns the row -> (point, slot) divisions into shifts
  const int rows_used = TP * kp;

  // neighbour row ids of this lane's DMA pieces for one tile (always PPW loads, clamped, so that the
  // number of VMEM operations per wave is a compile-time constant for the counted waits below)
  int grow[PPW];
  auto load_ids = [&](int t) {
    const int tt = t < t_end ? t : t_end - 1;
    const int b = tt / a.tiles_per_cloud;
    const int n0 = (tt % a.tiles_per_cloud) * TP;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      int p = wave + i * NW;
      if (p >= PIECES) p = PIECES - 1;
      int row = p * RPP + lane / NC;
      if (row >= rows_used) row = rows_used - 1;
      int pt = row / kp, j = row % kp;
      if (j >= k) j = 0;
      int n = n0 + pt;
      if (n >= a.N) n = a.N - 1;
      grow[i] = reinterpret_cast<const int *>(a.idx)[2 * (((long)b * a.N + n) * k + j)];   // low dword of the int64 id
    }
  };
  // issue the LDS-DMA gather of tile t into buffer `buf` from the ids in grow[]
  auto issue_gather = [&](int t, int buf) {
    const int b = t / a.tiles_per_cloud;
    const int n0 = (t % a.tiles_per_cloud) * TP;
    unsigned char *abuf = lds + buf * BUF_BYTES;
    const unsigned short *xb = a.x + (long)b * a.NX * CP;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = wave + i * NW;
      if (p < PIECES && p * RPP < rows_used) {
        const int row = p * RPP + lane / NC;
        const int cs = lane % NC;                       // physical chunk slot
        const int c = cs ^ ((row / RPB) & (NC - 1));    // logical chunk (swizzle on the source side)
        // 32-bit byte offset from the cloud's (wave-uniform) base: scalar base + VGPR offset addressing
        const unsigned char *src = reinterpret_cast<const unsigned char *>(xb) +
                                   ((unsigned int)grow[i] * (unsigned int)ROW_BYTES + (unsigned int)(c * 16));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(abuf + p * 1024), 16, 0, 0);
      }
    }
    // centre rows of the tile's points (linear, no swizzle)
    for (int p = wave; p < CPIECES; p += NW) {
      const int row = p * RPP + lane / NC;
      int n = n0 + (row < TP ? row : 0);
      if (n >= a.N) n = a.N - 1;
      const unsigned short *src = xb + (long)n * CP + (lane % NC) * 8;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(abuf + A_BYTES + p * 1024), 16, 0, 0);
    }
  };

  float s1 = 0.f, s2 = 0.f;
  int cur_b = t_begin / a.tiles_per_cloud;
  const int cpg = a.Cout / a.G;  // channels per group (multiple of 32 -> a column block is in one group)

  auto flush_stats = [&](int b) {
    double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      d1 += __shfl_xor(d1, o);
      d2 += __shfl_xor(d2, o);
    }
    if (lane == 0) {
      const int g = (cg * 32) / cpg;
      atomicAdd(a.gsum + ((long)b * a.G + g) * 2, d1);
      atomicAdd(a.gsum + ((long)b * a.G + g) * 2 + 1, d2);
    }
    s1 = 0.f;
    s2 = 0.f;
  };

  load_ids(t_begin);
  issue_gather(t_begin, 0);
  load_ids(t_begin + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    const int b = t / a.tiles_per_cloud;
    const int n0 = (t % a.tiles_per_cloud) * TP;
    if (b != cur_b) {
      flush_stats(cur_b);
      cur_b = b;
    }
    if (t + 1 < t_end) issue_gather(t + 1, buf ^ 1);   // ids were fetched one tile ago
    load_ids(t + 2);                                   // PPW plain loads, consumed next iteration

    unsigned char *abuf = lds + buf * BUF_BYTES;
    const unsigned char *cbuf = abuf + A_BYTES;
    const bool active = rg * 128 < rows_used;  // wave-uniform

    f32x16 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][i] = 0.f;

    if (active) {
      // neighbour half (k-steps 0..NC/2-1): chunk (2s + lh) ^ swz = (2s) ^ (lh ^ swz) since 2s and lh share no
      // bit, and rows are ROW_BYTES-aligned, so the address is base ^ (2s << 4) with base = row start ^ ((lh ^ swz)
      // << 4): ONE xor with a literal per fragment.  Centre half: purely additive (immediate offsets).
      int abase[4], cbase[4];
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int row = rg * 128 + rb * 32 + lr;
        abase[rb] = (row * ROW_BYTES) ^ ((lh ^ ((row / RPB) & (NC - 1))) << 4);
        cbase[rb] = (row / kp) * ROW_BYTES + (lh << 4);
      }
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) {
        bf16x8 af[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
          const unsigned char *p;
          if (NC == 1)        // one 16-B chunk per row: the half-wave decides (lh = 0 neighbour row, 1 centre row)
            p = lh == 0 ? abuf + (abase[rb] ^ (lh << 4)) : cbuf + cbase[rb] - (lh << 4);
          else
            p = s < NC / 2 ? abuf + (abase[rb] ^ ((2 * s) << 4)) : cbuf + cbase[rb] + ((2 * s - NC) << 4);
          af[rb] = *reinterpret_cast<const bf16x8 *>(p);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[rb], breg[s], acc[rb], 0, 0, 0);
      }
    }
    // ---- per 32-row block: max / min (/ arg) over rows per column, sums for GroupNorm
    float *pmax = reinterpret_cast<float *>(lds + 2 * BUF_BYTES);  // RWT > 1 only: [blocks][Cout]
    float *pmin = pmax + (TILE_ROWS / 32) * COUT;
    int *pamax = reinterpret_cast<int *>(pmin + (TILE_ROWS / 32) * COUT);
    int *pamin = pamax + (TILE_ROWS / 32) * COUT;
    float bmx[4], bmn[4];
    int bax[4], ban[4];
    if (active) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        const int blk = rg * 4 + rb;
        bmx[rb] = -__builtin_inff(); bmn[rb] = __builtin_inff(); bax[rb] = 0; ban[rb] = 0;
        if (blk * 32 >= rows_used) continue;          // wave-uniform: block beyond the tile's points
        if (n0 + (blk * 32) / kp >= a.N) continue;    // tail tile: point past the end of the cloud
        const int rbase = (blk * 32) % kp + 4 * lh;   // point-row of register 0 (kp % 32 == 0)
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rr = rbase + (i & 3) + 8 * (i >> 2);
          const float v = (kp == k || rr < k) ? acc[rb][i] : 0.f;
          ps += v;
          pq = fmaf(v, v, pq);
        }
        s1 += ROUTED ? ps * sgn : ps;                   // acc holds sgn * y in routed mode
        s2 += pq;
        float mx = acc[rb][0], mn = acc[rb][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) {
          mx = fmaxf(mx, acc[rb][i]);
          if (!ROUTED) mn = fminf(mn, acc[rb][i]);
        }
        int ax = 0x7fffffff, an = 0x7fffffff;
        if (WITH_ARG) {
          // the point-row of register i grows with i, so the lowest row is the lowest i
          int ix = 0, in_ = 0;
#pragma unroll
          for (int i = 15; i >= 0; --i) {
            ix = acc[rb][i] == mx ? i : ix;
            if (!ROUTED) in_ = acc[rb][i] == mn ? i : in_;
          }
          ax = rbase + (ix & 3) + 8 * (ix >> 2);
          an = rbase + (in_ & 3) + 8 * (in_ >> 2);
        }
        // combine the two half-waves (rows 4h..)
        const float mx2 = __shfl_xor(mx, 32);
        const float mn2 = ROUTED ? mn : __shfl_xor(mn, 32);
        if (WITH_ARG) {
          const int ax2 = __shfl_xor(ax, 32);
          ax = mx2 > mx ? ax2 : (mx2 == mx ? min(ax, ax2) : ax);
          if (!ROUTED) {
            const int an2 = __shfl_xor(an, 32);
            an = mn2 < mn ? an2 : (mn2 == mn ? min(an, an2) : an);
          }
        }
        mx = fmaxf(mx, mx2);
        mn = fminf(mn, mn2);
        bmx[rb] = mx; bmn[rb] = mn; bax[rb] = ax; ban[rb] = an;
        if (RWT > 1 && lh == 0) {
          const int o = blk * COUT + cg * 32 + lr;
          pmax[o] = mx;
          pmin[o] = mn;
          if (WITH_ARG) {
            pamax[o] = ax;
            pamin[o] = an;
          }
        }
      }
    }
    if (RWT == 1) {
      // every point of the tile lives in this wave: combine its kp/32 blocks in registers and store
      if (lh == 0) {
        const int nb = kp / 32;   // 1, 2 or 4 (3 when k in (64, 96]: TP == 1, blocks 0..2)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          if (pt < TP && n0 + pt < a.N) {
            float mx = -__builtin_inff(), mn = __builtin_inff();
            int ax = 0, an = 0;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
              if (rb >= pt * nb && rb < (pt + 1) * nb) {
                if (WITH_ARG) {
                  if (bmx[rb] > mx) ax = bax[rb];   // blocks ascend in point-row: strict > keeps the lowest row
                  if (bmn[rb] < mn) an = ban[rb];
                }
                mx = fmaxf(mx, bmx[rb]);
                mn = fminf(mn, bmn[rb]);
              }
            }
            const long o = ((long)b * a.N + n0 + pt) * COUT + cg * 32 + lr;
            a.ymax[o] = ROUTED ? mx * sgn : mx;
            if (!ROUTED) a.ymin[o] = mn;
            if (WITH_ARG) {
              a.amax[o] = (unsigned char)ax;
              if (!ROUTED) a.amin[o] = (unsigned char)an;
            }
          }
        }
      }
      // DMA of tile t+1 is older than the PPW id loads issued after it: wait for everything but those
      if (PPW == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else if (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (PPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // NOTE: the stores above are younger than the id loads, so the counted wait is conservative
      // (it also covers them only when they retire in order); correctness needs only the DMA.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      continue;
    }
    __syncthreads();

    // ---- RWT > 1: combine the kp/32 blocks of each point through LDS, write (B,N,Cout)
    {
      const int nb = kp / 32;
      const int Cout = COUT;
      for (int e = threadIdx.x; e < TP * Cout; e += 64 * NW) {
        const int pt = e / Cout, col = e % Cout;
        const int n = n0 + pt;
        if (n >= a.N) continue;
        float mx = -__builtin_inff(), mn = __builtin_inff();
        int ax = 0, an = 0;
        for (int q = 0; q < nb; ++q) {
          const int o = (pt * nb + q) * Cout + col;
          const float v1 = pmax[o], v2 = pmin[o];
          if (WITH_ARG) {
            if (v1 > mx) ax = pamax[o];   // blocks ascend in point-row, strict > keeps the lowest row
            if (v2 < mn) an = pamin[o];
          }
          mx = fmaxf(mx, v1);
          mn = fminf(mn, v2);
        }
        const long o = ((long)b * a.N + n) * Cout + col;
        a.ymax[o] = ROUTED ? mx * (a.gamma_route[col] >= 0.f ? 1.f : -1.f) : mx;
        if (!ROUTED) a.ymin[o] = mn;
        if (WITH_ARG) {
          a.amax[o] = (unsigned char)ax;
          if (!ROUTED) a.amin[o] = (unsigned char)an;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile's DMA has landed
    __syncthreads();
  }
  flush_stats(cur_b);
}

// ------------------------------------------------------------------ fp32 exact path (VALU)
// Literal reference arithmetic: y = sum_c W[co][c]*(x_j[c]-x_i[c]) + sum_c W[co][C+c]*x_i[c],
// one k-ordered fmaf chain (bit-exact vs oracle).  One workgroup per point, one thread per
// output channel; the k edge rows are staged in LDS.  Parity path, not the fast path.
__global__ __launch_bounds__(256) void edgeconv_fwd_f32_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const int64_t *__restrict__ idx, int N, int NX, int C, int k,
                                                               int Cout, int G, float *__restrict__ ymax,
                                                               float *__restrict__ ymin, unsigned char *__restrict__ amax,
                                                               unsigned char *__restrict__ amin, double *__restrict__ gsum,
                                                               const float *__restrict__ gamma_route) {
  extern __shared__ float e[];  // [k][2C] edge rows, then [Cout] scratch for sums
  const int n = blockIdx.x, b = blockIdx.y;
  const float *xb = x + (long)b * NX * C;
  const float *xi = xb + (long)n * C;
  for (int t = threadIdx.x; t < k * C; t += blockDim.x) {
    const int j = t / C, c = t % C;
    const long g = idx[((long)b * N + n) * k + j];
    const float ctr = xi[c];
    e[j * 2 * C + c] = xb[g * C + c] - ctr;
    e[j * 2 * C + C + c] = ctr;
  }
  __syncthreads();
  double *gs = reinterpret_cast<double *>(e + (size_t)k * 2 * C + ((k * 2 * C) & 1));
  for (int co = threadIdx.x; co < Cout; co += blockDim.x) {
    const float *wr = w + (long)co * 2 * C;
    float mx = -__builtin_inff(), mn = __builtin_inff();
    int ax = 0, an = 0;
    double s1 = 0.0, s2 = 0.0;
    for (int j = 0; j < k; ++j) {
      float y = 0.f;
      for (int c = 0; c < 2 * C; ++c) y = fmaf(wr[c], e[j * 2 * C + c], y);
      if (y > mx) { mx = y; ax = j; }
      if (y < mn) { mn = y; an = j; }
      s1 += (double)y;
      s2 += (double)y * (double)y;
    }
    const long o = ((long)b * N + n) * Cout + co;
    if (gamma_route) {  // routed mode: only the extreme that GroupNorm+LeakyReLU will select
      const bool pos = gamma_route[co] >= 0.f;
      ymax[o] = pos ? mx : mn;
      if (amax) amax[o] = (unsigned char)(pos ? ax : an);
    } else {
      ymax[o] = mx; ymin[o] = mn;
      if (amax) { amax[o] = (unsigned char)ax; amin[o] = (unsigned char)an; }
    }
    gs[co * 2] = s1; gs[co * 2 + 1] = s2;
  }
  __syncthreads();
  if ((int)threadIdx.x < G * 2) {
    const int g = threadIdx.x >> 1, which = threadIdx.x & 1;
    const int cpg = Cout / G;
    double t = 0.0;
    for (int co = g * cpg; co < (g + 1) * cpg; ++co) t += gs[co * 2 + which];
    atomicAdd(gsum + ((long)b * G + g) * 2 + which, t);
  }
}

// ------------------------------------------------------------------ GroupNorm + LeakyReLU on the routed extreme
__global__ __launch_bounds__(256) void edgeconv_finish_kernel(const float *__restrict__ ymax, const float *__restrict__ ymin,
                                                              const double *__restrict__ gsum, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, int N, int k, int Cout, int G,
                                                              float eps, float slope, float *__restrict__ out_cm,
                                                              float *__restrict__ out_pm, float *__restrict__ mean_rstd,
                                                              unsigned short *__restrict__ out_bf, int bf_pitch) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int cpg = Cout / G;
  const double cnt = (double)cpg * N * k;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, c = c0 + tx;
    float v = 0.f;
    if (n < N && c < Cout) {
      const int g = c / cpg;
      const double m = gsum[((long)b * G + g) * 2] / cnt;
      double var = gsum[((long)b * G + g) * 2 + 1] / cnt - m * m;
      if (var < 0.0) var = 0.0;
      const float mean = (float)m, rstd = (float)(1.0 / sqrt(var + (double)eps));
      const float ga = gamma[c];
      const long o = ((long)b * N + n) * Cout + c;
      const float y = (ga >= 0.f || ymin == nullptr) ? ymax[o] : ymin[o];   // ymin == NULL: ymax already holds the routed extreme
      const float z = (y - mean) * rstd * ga + beta[c];
      v = z > 0.f ? z : z * slope;
      if (out_pm) out_pm[o] = v;
      if (out_bf) out_bf[((long)b * N + n) * bf_pitch + c] = f32_to_bf16(v);
      if (mean_rstd && n == 0 && (c % cpg) == 0) {
        mean_rstd[((long)b * G + g) * 2] = mean;
        mean_rstd[((long)b * G + g) * 2 + 1] = rstd;
      }
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (out_cm) {
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, n = n0 + tx;
      if (c < Cout && n < N) out_cm[((long)b * Cout + c) * N + n] = tile[tx][i];
    }
  }
}


// ------------------------------------------------------------------ graph aggregations for backward
// s[n] = sum_j x[idx[n,j]]  (neighbour sum, gather).  G lanes per point (channels across lanes; 64/G points per wave, so
// narrow rows -- the 6-channel input cloud -- still fill the wave); sixteen row loads in flight per lane, the ids of a
// point read once (G = 64: one coalesced load of 64 ids, then v_readlane); adds stay in neighbour order.
template <int G>
__global__ __launch_bounds__(256) void neighbor_sum_kernel(const float *__restrict__ x, const int64_t *__restrict__ idx,
                                                           int N, int C, int k, float *__restrict__ s) {
  constexpr int PW = 64 / G;
  constexpr int U = 16;
  const int lane = lane_id();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int p = lane / G, cl = lane % G;
  const int n = (tile * 4 + wave_id()) * PW + p;
  if ((tile * 4 + wave_id()) * PW >= N) return;                 // wave-uniform
  const int nn = min(n, N - 1);
  const float *xb = x + (long)b * N * C;
  const int64_t *ip = idx + ((long)b * N + nn) * k;
  for (int c0 = 0; c0 < C; c0 += G) {
    const int c = c0 + cl;
    const bool cok = c < C;
    const int cc = cok ? c : C - 1;
    float acc = 0.f;
    int ids64 = 0;
    for (int j0 = 0; j0 < k; j0 += U) {
      int ids[U];
      if (G == 64) {
        if ((j0 & 63) == 0) ids64 = (int)ip[min(j0 + lane, k - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) ids[u] = __builtin_amdgcn_readlane(ids64, ((j0 & 63) + u) & 63);
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) ids[u] = (int)ip[min(j0 + u, k - 1)];
      }
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = xb[(unsigned int)ids[u] * (unsigned int)C + (unsigned int)cc];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (j0 + u < k) acc += v[u];
    }
    if (cok && n < N) s[((long)b * N + n) * C + c] = acc;
  }
}

// r[m] = sum_{(n,j): idx[n,j]=m} x[n]  and indeg[m]  (scatter; 256-B contiguous f32 atomics per row)
// r[m] = sum over incoming edges (n -> m) of x[n], indeg[m] = their number -- the transposed aggregation of
// the closed-form backward.  Scattering 4 M rows with global f32 atomics is the slow way (they execute at the
// memory side); building the inverted lists first costs two more passes of integer atomics.  Here the
// destinations are PARTITIONED: a workgroup owns R consecutive destination rows of one cloud as an LDS
// accumulator, scans the cloud's whole edge list (coalesced, L2-resident: every partition of the cloud reads
// the same 4 MB) and adds x[n] for the edges that land in its range -- on average k*R/N per 64 edges.
// The accumulator is 64-bit FIXED POINT: ds_add_f32 retires 0.33 lane-ops/clk/CU on gfx950 against 3 for
// ds_add_u64 (tools/micro/lds_atomic_bench.hip), and integer sums are order-independent, so r is bitwise
// reproducible.  Scale 2^S with S chosen from max|x| (absmax_kernel) and N*k so that no sum can overflow:
// the quantisation step is <= 2^-43 * max|x| * N*k / 2^19, far below f32 rounding of the result.
// neighbour ids as u16 (N <= 65536): the scan below re-reads the list once per partition, and at 2 bytes per
// edge a cloud's list (1 MB) plus its rows stays inside the XCD's 4 MB L2
__global__ __launch_bounds__(256) void idx_to_u16_kernel(const int64_t *__restrict__ idx, long n, unsigned short *__restrict__ out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 < n) {
    const longlong2 v = *reinterpret_cast<const longlong2 *>(idx + i);
    *reinterpret_cast<unsigned int *>(out + i) = (unsigned int)(v.x & 0xffff) | ((unsigned int)(v.y & 0xffff) << 16);
  } else if (i < n) {
    out[i] = (unsigned short)idx[i];
  }
}

__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, long n, unsigned int *__restrict__ out) {
  // 16 bytes per lane, four loads in flight; ONE atomic per workgroup (2048 same-address atomics from the waves of a
  // 512-workgroup grid cost more than the 16 MB read)
  __shared__ float red[4];
  float m = 0.f;
  const long n4 = n >> 2;
  const float4 *x4 = reinterpret_cast<const float4 *>(x);
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = x4[i + u * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w))));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  if (blockIdx.x == 0 && (long)threadIdx.x < n - (n4 << 2)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane_id() == 0) red[wave_id()] = m;
  __syncthreads();
  if (threadIdx.x == 0)                                  // non-negative floats order like their bit patterns
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

template <typename IdxT, bool C64>
__global__ __launch_bounds__(1024) void reverse_sum_lds_kernel(const float *__restrict__ x, const IdxT *__restrict__ idx,
                                                               const unsigned int *__restrict__ absmax_bits, int B, int N,
                                                               int C, int k, int R, float *__restrict__ r,
                                                               float *__restrict__ indeg) {
  if (C64) C = 64;                                       // compile-time channel count: the loads below lose their C > 0 guards
  extern __shared__ unsigned long long qacc[];          // [R][C] fixed-point sums, then [R] u32 counts
  unsigned int *cnt = reinterpret_cast<unsigned int *>(qacc + (long)R * C);
  const int lane = lane_id(), wave = wave_id();
  // workgroup id -> (cloud, partition) with the CLOUD fastest: ids are dealt round-robin to the 8 XCDs, so all
  // partitions of a cloud run on one XCD and share its L2 copy of the cloud's edge list and rows (with the
  // partition fastest every XCD streamed all clouds through a 4 MB L2: 880 us instead of ~100)
  const int b = blockIdx.x % B, m0 = (blockIdx.x / B) * R;
  for (int i = threadIdx.x; i < R * C; i += 1024) qacc[i] = 0ull;
  for (int i = threadIdx.x; i < R; i += 1024) cnt[i] = 0u;
  __syncthreads();
  // S = 62 - ceil(log2(max|x|)) - ceil(log2(N*k)), clamped to [0, 40]
  const float mx = __uint_as_float(*absmax_bits);
  int ex = 0;
  if (mx > 0.f) (void)frexpf(mx, &ex);                   // mx = f * 2^ex, f in [0.5,1)  => mx < 2^ex
  int S = 62 - ex - (64 - __clzll((long long)N * k));
  S = S < 0 ? 0 : (S > 40 ? 40 : S);
  const float scale = ldexpf(1.f, S);
  const double inv = ldexp(1.0, -S);
  const IdxT *ib = idx + (long)b * N * k;
  const float *xb = x + (long)b * N * C;
  // A wave walks source points four at a time: their neighbour lists (lane = slot) and, unconditionally, their
  // rows (lane = channel; P(some edge of a point lands in the range) = 1-(1-R/N)^k ~ 0.9) are fetched together,
  // so the per-match work is a scalar read and ONE conflict-free ds_add_u64 per 64 channels.
  // (loads are unconditional on clamped addresses and masked afterwards: a predicated load makes hipcc emit
  //  branch + s_waitcnt vmcnt(0) per load, i.e. eight serialised round trips per iteration -- 880 us vs 90;
  //  and the next four points are fetched before the current four are processed)
  float xf_n[4];
  int mraw_n[4];
  auto fetch = [&](int n0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) xf_n[i] = C > 0 ? xb[(long)min(n0 + i, N - 1) * C + min(lane, C - 1)] : 0.f;   // C == 0: count only
#pragma unroll
    for (int i = 0; i < 4; ++i) mraw_n[i] = (int)ib[(long)min(n0 + i, N - 1) * k + min(lane, k - 1)];
  };
  fetch(wave * 4);
  for (int n0 = wave * 4; n0 < N; n0 += 16 * 4) {
    float xf[4];
    int mraw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { xf[i] = xf_n[i]; mraw[i] = mraw_n[i]; }
    if (n0 + 64 < N) fetch(n0 + 64);                   // wave-uniform
    unsigned long long xr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = (unsigned long long)__float2ll_rn(xf[i] * scale);
    for (int j0 = 0; j0 < k; j0 += 64) {
      int mi[4];
      if (j0 > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) mraw[i] = (int)ib[(long)min(n0 + i, N - 1) * k + min(j0 + lane, k - 1)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mi[i] = (n0 + i < N && j0 + lane < k) ? mraw[i] - m0 : -1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned long long hit = __ballot((unsigned)mi[i] < (unsigned)R);
        while (hit) {
          const int l = __ffsll((long long)hit) - 1;
          hit &= hit - 1;
          const int mm = readlane_i(mi[i], l);
          if (C64) {                               // one full-wave add per match, no channel masks
            atomicAdd(&qacc[mm * 64 + lane], xr[i]);
          } else {
            if (lane < C) atomicAdd(&qacc[mm * C + lane], xr[i]);
            for (int c = lane + 64; c < C; c += 64)
              atomicAdd(&qacc[mm * C + c], (unsigned long long)__float2ll_rn(xb[(long)(n0 + i) * C + c] * scale));
          }
          if (lane == 0) atomicAdd(&cnt[mm], 1u);
        }
      }
    }
  }
  __syncthreads();
  const int rows = min(R, N - m0);
  if (C > 0) {
    float *rb = r + ((long)b * N + m0) * C;
    for (int i = threadIdx.x; i < rows * C; i += 1024) rb[i] = (float)((double)(long long)qacc[i] * inv);
  }
  if (indeg)
    for (int i = threadIdx.x; i < rows; i += 1024) indeg[(long)b * N + m0 + i] = (float)cnt[i];
}


// in-degrees only (the first EdgeConv layer needs no transposed sum: the cloud itself carries no gradient): every
// workgroup owns R destination counters in LDS and scans the cloud's u16 edge list, 8 ids per 16-byte load
__global__ __launch_bounds__(1024) void indeg_lds_kernel(const unsigned short *__restrict__ idx, int B, int N, int k, int R,
                                                         float *__restrict__ indeg) {
  extern __shared__ unsigned int cnt_s[];
  const int b = blockIdx.x % B, m0 = (blockIdx.x / B) * R;     // cloud fastest: one cloud per XCD (see reverse_sum_lds_kernel)
  for (int i = threadIdx.x; i < R; i += 1024) cnt_s[i] = 0u;
  __syncthreads();
  const long E = (long)N * k;
  const unsigned short *ib = idx + (long)b * E;
  const long E8 = E & ~7L;                                     // b*E*2 bytes is 16-byte aligned when E % 8 == 0 (checked by the host)
  for (long e = (long)threadIdx.x * 8; e < E8; e += 1024 * 8) {
    const uint4 v = *reinterpret_cast<const uint4 *>(ib + e);
    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned int lo = (w[i] & 0xffffu) - (unsigned int)m0, hi = (w[i] >> 16) - (unsigned int)m0;
      if (lo < (unsigned int)R) atomicAdd(&cnt_s[lo], 1u);
      if (hi < (unsigned int)R) atomicAdd(&cnt_s[hi], 1u);
    }
  }
  for (long e = E8 + threadIdx.x; e < E; e += 1024) {
    const unsigned int d = (unsigned int)ib[e] - (unsigned int)m0;
    if (d < (unsigned int)R) atomicAdd(&cnt_s[d], 1u);
  }
  __syncthreads();
  const int rows = min(R, N - m0);
  for (int i = threadIdx.x; i < rows; i += 1024) indeg[(long)b * N + m0 + i] = (float)cnt_s[i];
}

// ------------------------------------------------------------------ key-point edge block (offset module)
// OFFSET_PRED_MODULE (M4:398-452) builds, for every point, k edges to a fixed set of NK key points,
// scales the 131-channel edge feature by the KPAM weight att[n,j] and runs Conv2d(131->128)+GN+
// LeakyReLU+max.  The conv is linear, so y[n,j,:] = att[n,j] * (U[m_j,:] - V[n,:]) with
// U = Wf.f_key + Wp.p_key (NK rows per cloud) and V = Wp.p_n: the whole key table sits in LDS and the
// (B,N,k,128) tensor (1 GB at B=8,N=8192,k=30) is never formed.  Same outputs as edgeconv_fwd.
// NCH = 64-channel blocks a lane carries per pass (2 when Cout % 128 == 0: the per-neighbour (weight, key id) reads
// and the loop control are shared by two channels); the neighbour loop is unrolled by two so that four key-table reads
// are in flight per lane (one dependent LDS round trip per neighbour made the loop latency-bound: 185 us at B=8,
// N=8192, k=30, Cout=128 against ~65 us of VALU work).
// ROUTED (gamma_route given): GroupNorm's scale has the sign of gamma and LeakyReLU is increasing, so the max over k
// the block returns comes from max_j y where gamma >= 0 and from min_j y elsewhere -- known before the kernel runs.
// The key table and V are loaded with the channel's sign folded in (exact), one extreme of s*y is tracked (7 instead of
// 10 VALU per edge and channel) and s*max is stored to ymax / its position to amax; ymin / amin are not written.
template <int NCH, bool ROUTED>
__global__ __launch_bounds__(1024) void keyedge_fwd_kernel(const float *__restrict__ att, const int64_t *__restrict__ kidx,
                                                          const float *__restrict__ U, const float *__restrict__ V,
                                                          int N, int k, int NK, int Cout, int G, int pts_per_block,
                                                          float *__restrict__ ymax, float *__restrict__ ymin,
                                                          unsigned char *__restrict__ amax, unsigned char *__restrict__ amin,
                                                          double *__restrict__ gsum, const float *__restrict__ gamma_route) {
  extern __shared__ __attribute__((aligned(16))) float u_lds[];  // NK * Cout, then 16 waves x 64 (weight, key id) slots
  __shared__ double red[128];       // (group, statistic) sums of this workgroup, G <= 64
  const int lane = lane_id(), wave = wave_id();
  const int b = blockIdx.y;
  if (threadIdx.x < 128) red[threadIdx.x] = 0.0;
  const float *Ub = U + (long)b * NK * Cout;
  for (int i = threadIdx.x; i < NK * Cout; i += blockDim.x) {
    float u = Ub[i];
    if (ROUTED && gamma_route[i % Cout] < 0.f) u = -u;
    u_lds[i] = u;
  }
  __syncthreads();
  float2 *slot = reinterpret_cast<float2 *>(u_lds + (((long)NK * Cout + 3) & ~3L)) + wave * 64;    // 16-byte aligned
  const int n_lo = blockIdx.x * pts_per_block;
  const int n_hi = min(n_lo + pts_per_block, N);
  const int cpg = Cout / G;
  for (int c0 = 0; c0 < Cout; c0 += 64 * NCH) {
    int c[NCH];
    bool cv[NCH], neg[NCH];
    float s1[NCH], s2[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      c[h] = c0 + 64 * h + lane;
      cv[h] = c[h] < Cout;
      c[h] = min(c[h], Cout - 1);
      s1[h] = 0.f; s2[h] = 0.f;
      neg[h] = ROUTED && gamma_route[c[h]] < 0.f;
    }
    // the point's k weights / key ids: one lane-parallel load each (lane = slot), read back in the loop -- loading
    // att[pn*k+j] inside it costs a dependent global round trip per iteration (347 -> ~100 us); the NEXT point's
    // operands are fetched while the current point is processed
    const int nstep = (int)(blockDim.x >> 6);
    float v_n[NCH], a_n[4] = {0.f, 0.f, 0.f, 0.f};
    int m_n[4] = {0, 0, 0, 0};
#pragma unroll
    for (int h = 0; h < NCH; ++h) v_n[h] = 0.f;
    auto fetch = [&](int n) {
      const long pn = (long)b * N + min(n, n_hi - 1);
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        v_n[h] = V[pn * Cout + c[h]];
        if (neg[h]) v_n[h] = -v_n[h];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = min(q * 64 + lane, k - 1);
        a_n[q] = att[pn * k + jj];
        m_n[q] = (int)kidx[pn * k + jj];
        if (q * 64 + 64 >= k) break;                   // wave-uniform
      }
    };
    if (n_lo + wave < n_hi) fetch(n_lo + wave);
    for (int n = n_lo + wave; n < n_hi; n += nstep) {
      const long pn = (long)b * N + n;
      float v[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) v[h] = v_n[h];
      float a_l[4];
      int m_l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { a_l[q] = a_n[q]; m_l[q] = m_n[q]; }
      if (n + nstep < n_hi) fetch(n + nstep);
      float mx[NCH], mn[NCH];
      int ax[NCH], an[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) { mx[h] = -__builtin_inff(); mn[h] = __builtin_inff(); ax[h] = 0; an[h] = 0; }
      auto edge = [&](float a, int m, int j, const float *uv) {
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
          const float y = a * (uv[h] - v[h]);
          if (y > mx[h]) { mx[h] = y; ax[h] = j; }
          if (!ROUTED && y < mn[h]) { mn[h] = y; an[h] = j; }
          s1[h] += y;
          s2[h] = fmaf(y, y, s2[h]);
        }
      };
      if (k <= 64) {                                   // one broadcast LDS read per neighbour pair instead of v_readlane pairs
        __builtin_amdgcn_wave_barrier();
        if (lane < k) slot[lane] = float2{a_l[0], __int_as_float(m_l[0])};
        __builtin_amdgcn_wave_barrier();
        int j = 0;
        for (; j + 1 < k; j += 2) {
          const float4 am = *reinterpret_cast<const float4 *>(slot + j);
          const int m0 = __float_as_int(am.y), m1 = __float_as_int(am.w);
          float u0[NCH], u1[NCH];
#pragma unroll
          for (int h = 0; h < NCH; ++h) { u0[h] = u_lds[m0 * Cout + c[h]]; u1[h] = u_lds[m1 * Cout + c[h]]; }
          edge(am.x, m0, j, u0);
          edge(am.z, m1, j + 1, u1);
        }
        if (j < k) {
          const float2 am = slot[j];
          const int m0 = __float_as_int(am.y);
          float u0[NCH];
#pragma unroll
          for (int h = 0; h < NCH; ++h) u0[h] = u_lds[m0 * Cout + c[h]];
          edge(am.x, m0, j, u0);
        }
      } else {
        for (int j = 0; j < k; ++j) {
          float a = 0.f;
          int m = 0;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if ((j >> 6) == q) { a = readlane_f(a_l[q], j & 63); m = readlane_i(m_l[q], j & 63); }
          float u0[NCH];
#pragma unroll
          for (int h = 0; h < NCH; ++h) u0[h] = u_lds[m * Cout + c[h]];
          edge(a, m, j, u0);
        }
      }
#pragma unroll
      for (int h = 0; h < NCH; ++h)
        if (cv[h]) {
          if (ROUTED) {
            ymax[pn * Cout + c[h]] = neg[h] ? -mx[h] : mx[h];
            if (amax) amax[pn * Cout + c[h]] = (unsigned char)ax[h];
          } else {
            ymax[pn * Cout + c[h]] = mx[h]; ymin[pn * Cout + c[h]] = mn[h];
            if (amax) { amax[pn * Cout + c[h]] = (unsigned char)ax[h]; amin[pn * Cout + c[h]] = (unsigned char)an[h]; }
          }
        }
    }
    // one LDS f64 atomic pair per GroupNorm group and wave, one global pair per group and WORKGROUP at the end: device
    // atomics on one address retire ~0.4 us apart, and every wave of the grid gets here at about the same time
    const int seg = (cpg % 64) == 0 ? 64 : cpg;
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      if (neg[h]) s1[h] = -s1[h];                      // the sums are those of y itself
      if ((seg & (seg - 1)) == 0 && seg <= 64) {
        double d1 = cv[h] ? (double)s1[h] : 0.0, d2 = cv[h] ? (double)s2[h] : 0.0;
        for (int o = seg >> 1; o >= 1; o >>= 1) { d1 += __shfl_xor(d1, o); d2 += __shfl_xor(d2, o); }
        if ((lane & (seg - 1)) == 0 && cv[h]) {
          atomicAdd(&red[(c[h] / cpg) * 2], d1);
          atomicAdd(&red[(c[h] / cpg) * 2 + 1], d2);
        }
      } else if (cv[h]) {
        atomicAdd(&red[(c[h] / cpg) * 2], (double)s1[h]);
        atomicAdd(&red[(c[h] / cpg) * 2 + 1], (double)s2[h]);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * G) atomicAdd(gsum + (long)b * G * 2 + threadIdx.x, red[threadIdx.x]);
}


// Backward of the key-point edge block.  dy[n,j,c] = coef[n,c]*[j == jsel[n,c]] + A[b,c] + B[b,c]*y[n,j,c]
// (sparse routed gradient + GroupNorm coupling, see gcanet_amd/dgcnn.py) with y = att*(U[m]-V).  Everything that
// is affine in y collapses (y is linear in U, V, att), so the (n,j,c) triple loop carries no atomics:
//   dV[n,c]   = -(cf*att_js + A_c*a1[n] + B_c*(sum_j att_j^2 U[m_j,c] - a2[n]*V[n,c]))           (complete here)
//   datt[n,j] = sum_{c: js=j} cf*d_js + (UA[m_j] - VA[n]) + att_j*(UB2[m_j] - 2 X[n,m_j] + VB2[n])  (complete here;
//               UA = U.A, UB2 = U^2.B per key point, VA/VB2 per point, X = (V o B).U^T from one GEMM outside)
//   dU[m,c]   = dUsp[m,c] + A_c*T1[m] + B_c*(U[m,c]*T2[m] - (A2^T V)[m,c])   with A2[n,m] = sum_j att_j^2 [m_j = m]
//               written here as a dense (N x NK) matrix for one tall-skinny GEMM; dUsp, T1, T2 accumulate in LDS
//               (B*N*(Cout+2k) float atomics instead of B*N*k*Cout: ds_add_f32 retires 0.33 lane-ops/clk/CU).
// One wave per point; U, the sparse dU accumulator and the per-key tables live in LDS.
template <int NCH>
__global__ __launch_bounds__(1024) void keyedge_bwd_kernel(const float *__restrict__ att, const int64_t *__restrict__ kidx,
                                                          const float *__restrict__ U, const float *__restrict__ V,
                                                          const float *__restrict__ coef, const int64_t *__restrict__ jsel,
                                                          const float *__restrict__ Ac, const float *__restrict__ Bc,
                                                          const float *__restrict__ X, int N, int k, int NK, int Cout,
                                                          int pts_per_block, float *__restrict__ datt, float *__restrict__ dV,
                                                          float *__restrict__ A2, float *__restrict__ dUsp,
                                                          float *__restrict__ T12) {
  // U | dUsp | UA | UB2 | T1, T2 (64-bit fixed point) | per-wave: (att, kidx)[64], datt_sp[64] (64-bit fixed point), row[NKp]
  // The per-key sums T1, T2 and the per-neighbour sums of datt are scatter-adds with same-address collisions inside a
  // wave: ds_add_f32 retires 0.33 lane-ops/clk/CU, ds_add_u64 ~3 (tools/micro/lds_atomic_bench.hip) -- as f32 they were
  // ~60 of this kernel's 187 us.  Scales: 2^S with S from the largest magnitude that can arrive (the workgroup's max
  // |att| for T1/T2; the wave's max |value| of the current point and pass for datt), so that a sum of 256 terms stays
  // below 2^58 and the rounding is 2^-50 of that magnitude -- finer than the f32 sums it replaces, and order-free.
  extern __shared__ __attribute__((aligned(16))) float lds_f[];     // (behind the 4-byte static below: keep it 16-byte aligned)
  __shared__ unsigned int attmax_s;
  const int NKp = (NK + 63) & ~63;
  const int UC = (NK * Cout + 3) & ~3;      // table pitch: keeps everything behind it 16-byte aligned
  float *u_lds = lds_f, *du_lds = u_lds + UC;
  float *ua = du_lds + UC, *ub2 = ua + NKp;
  unsigned long long *t1q = reinterpret_cast<unsigned long long *>(ub2 + NKp), *t2q = t1q + NKp;
  const int lane = lane_id(), wave = wave_id();
  float2 *wam = reinterpret_cast<float2 *>(reinterpret_cast<float *>(t2q + NKp) + wave * (256 + NKp));     // slot j: (att_j, key id as bits)
  unsigned long long *wdq = reinterpret_cast<unsigned long long *>(wam + 64);
  float *wrow = reinterpret_cast<float *>(wdq + 64);
  const int b = blockIdx.y;
  const int n_lo = blockIdx.x * pts_per_block;
  const int n_hi = min(n_lo + pts_per_block, N);
  for (int i = threadIdx.x; i < NK * Cout; i += blockDim.x) {
    u_lds[i] = U[(long)b * NK * Cout + i];
    du_lds[i] = 0.f;
  }
  for (int i = threadIdx.x; i < NKp; i += blockDim.x) { t1q[i] = 0ull; t2q[i] = 0ull; }
  wdq[lane] = 0ull;
  if (threadIdx.x == 0) attmax_s = 0u;
  __syncthreads();
  {
    float am = 0.f;
    const float *ab = att + ((long)b * N + n_lo) * k;
    for (int i = threadIdx.x; i < (n_hi - n_lo) * k; i += blockDim.x) am = fmaxf(am, fabsf(ab[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0) atomicMax(&attmax_s, __float_as_uint(am));     // non-negative floats order as their bit patterns
  }
  for (int m = threadIdx.x; m < NK; m += blockDim.x) {
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < Cout; ++c) {
      const float u = u_lds[m * Cout + c];
      sa = fmaf(Ac[(long)b * Cout + c], u, sa);
      sb = fmaf(Bc[(long)b * Cout + c] * u, u, sb);
    }
    ua[m] = sa;
    ub2[m] = sb;
  }
  __syncthreads();
  auto pow2 = [](int e) -> double { return __longlong_as_double((long long)(1023 + e) << 52); };
  auto fxq = [](float x, double scale) -> unsigned long long {       // round(x * scale), |x * scale| < 2^51
    return (unsigned long long)(__double_as_longlong(fma((double)x, scale, 6755399441055744.0)) - 0x4338000000000000LL);
  };
  auto fexp = [](float bound) -> int { return (int)((__float_as_uint(bound) >> 23) & 0xffu) - 127; };   // bound < 2^(e+1)
  const int e_att = fexp(__uint_as_float(attmax_s));
  const double sc1 = pow2(49 - e_att), sc2 = pow2(48 - 2 * e_att);      // att < 2^(e+1), att^2 < 2^(2e+2)
  // The NEXT point's operands are requested while the current point is processed (each point otherwise waits out two
  // dependent global round trips); with one channel pass (Cout <= 64 NCH) that includes its V / coef / jsel rows.
  const int nstep = (int)(blockDim.x >> 6);
  const bool jv = lane < k;
  const bool single = Cout <= 64 * NCH;
  float a_nx = 0.f, v_nx[NCH], cf_nx[NCH];
  int m_nx = 0, js_nx[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { v_nx[h] = 0.f; cf_nx[h] = 0.f; js_nx[h] = 0; }
  auto fetch = [&](int n) {
    const long pq = (long)b * N + n;
    a_nx = jv ? att[pq * k + lane] : 0.f;
    m_nx = jv ? (int)kidx[pq * k + lane] : 0;
    if (single) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const int cc = min(64 * h + lane, Cout - 1);
        v_nx[h] = V[pq * Cout + cc];
        cf_nx[h] = coef[pq * Cout + cc];
        js_nx[h] = (int)jsel[pq * Cout + cc];
      }
    }
  };
  if (n_lo + wave < n_hi) fetch(n_lo + wave);
  for (int n = n_lo + wave; n < n_hi; n += nstep) {
    const long pn = (long)b * N + n;
    const float a = a_nx;
    const int m = m_nx;
    float v_cur[NCH], cf_cur[NCH];
    int js_cur[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) { v_cur[h] = v_nx[h]; cf_cur[h] = cf_nx[h]; js_cur[h] = js_nx[h]; }
    if (n + nstep < n_hi) fetch(n + nstep);
    wam[lane] = float2{a, __int_as_float(m)};
    for (int i = lane; i < NKp; i += 64) wrow[i] = 0.f;
    __builtin_amdgcn_wave_barrier();
    float a1 = a, a2 = a * a;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
    if (jv) {
      wrow[m] = a * a;                       // top-k key ids of one point are distinct
      atomicAdd(&t1q[m], fxq(a, sc1));
      atomicAdd(&t2q[m], fxq(a * a, sc2));
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < NK; i += 64) A2[pn * NK + i] = wrow[i];
    float va = 0.f, vb2 = 0.f, dsp = 0.f;
    // NCH 64-channel blocks per lane and pass (2 when Cout % 128 == 0): the per-neighbour (att, key id) broadcast read and
    // the loop control serve two channels, and two neighbours are in flight per iteration (same fma order as one by one)
    for (int c0 = 0; c0 < Cout; c0 += 64 * NCH) {
      int c[NCH], js[NCH];
      bool cv[NCH];
      float v[NCH], a_c[NCH], b_c[NCH], cf[NCH], wu2[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        cv[h] = c0 + 64 * h + lane < Cout;
        c[h] = min(c0 + 64 * h + lane, Cout - 1);
        a_c[h] = Ac[(long)b * Cout + c[h]]; b_c[h] = Bc[(long)b * Cout + c[h]];
        if (single) {
          v[h] = v_cur[h]; cf[h] = cf_cur[h]; js[h] = js_cur[h];
        } else {
          v[h] = V[pn * Cout + c[h]];
          cf[h] = coef[pn * Cout + c[h]];
          js[h] = (int)jsel[pn * Cout + c[h]];
        }
        wu2[h] = 0.f;
      }
      int j = 0;
      for (; j + 1 < k; j += 2) {
        const float4 am = *reinterpret_cast<const float4 *>(wam + j);
        const int m0 = __float_as_int(am.y), m1 = __float_as_int(am.w);
        const float q0 = am.x * am.x, q1 = am.z * am.z;
        float u0[NCH], u1[NCH];
#pragma unroll
        for (int h = 0; h < NCH; ++h) { u0[h] = u_lds[m0 * Cout + c[h]]; u1[h] = u_lds[m1 * Cout + c[h]]; }
#pragma unroll
        for (int h = 0; h < NCH; ++h) { wu2[h] = fmaf(q0, u0[h], wu2[h]); wu2[h] = fmaf(q1, u1[h], wu2[h]); }
      }
      if (j < k) {
        const float2 am = wam[j];
        const int m0 = __float_as_int(am.y);
        const float q0 = am.x * am.x;
#pragma unroll
        for (int h = 0; h < NCH; ++h) wu2[h] = fmaf(q0, u_lds[m0 * Cout + c[h]], wu2[h]);
      }
      float val[NCH], vmax = 0.f;
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        const float2 sel = wam[js[h]];
        const float att_sel = sel.x;
        const int m_sel = __float_as_int(sel.y);
        const float d_sel = u_lds[m_sel * Cout + c[h]] - v[h];
        val[h] = cv[h] ? cf[h] * d_sel : 0.f;
        vmax = fmaxf(vmax, fabsf(val[h]));
        if (cv[h]) {
          dV[pn * Cout + c[h]] = -(cf[h] * att_sel + a_c[h] * a1 + b_c[h] * (wu2[h] - a2 * v[h]));
          atomicAdd(&du_lds[m_sel * Cout + c[h]], cf[h] * att_sel);
          va = fmaf(a_c[h], v[h], va);
          vb2 = fmaf(b_c[h] * v[h], v[h], vb2);
        }
      }
      // datt's routed part of this pass: sum_{c: js = j} cf * d_js, scattered by js in fixed point
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
      const int e_w = fexp(vmax);
      const double scw = pow2(49 - e_w);
#pragma unroll
      for (int h = 0; h < NCH; ++h)
        if (cv[h]) atomicAdd(&wdq[js[h]], fxq(val[h], scw));
      __builtin_amdgcn_wave_barrier();
      dsp += (float)((double)(long long)wdq[lane] * pow2(e_w - 49));
      __builtin_amdgcn_wave_barrier();
      wdq[lane] = 0ull;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { va += __shfl_xor(va, o); vb2 += __shfl_xor(vb2, o); }
    __builtin_amdgcn_wave_barrier();
    if (jv) datt[pn * k + lane] = dsp + (ua[m] - va) + a * (ub2[m] - 2.f * X[pn * NK + m] + vb2);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NK * Cout; i += blockDim.x) atomicAdd(dUsp + (long)b * NK * Cout + i, du_lds[i]);
  for (int i = threadIdx.x; i < NK; i += blockDim.x) {
    atomicAdd(T12 + ((long)b * 2) * NK + i, (float)((double)(long long)t1q[i] * pow2(e_att - 49)));
    atomicAdd(T12 + ((long)b * 2 + 1) * NK + i, (float)((double)(long long)t2q[i] * pow2(2 * e_att - 48)));
  }
}

template <int KSTEPS, int CW, int RWT>
static int launch_fwd_bf16(EcArgs &a, bool with_arg, hipStream_t st) {
  constexpr int CP = KSTEPS * 8;
  constexpr int TILE_ROWS = RWT * 128;
  constexpr int A_BYTES = TILE_ROWS * CP * 2;
  constexpr int C_BYTES = (((TILE_ROWS / 32) * CP * 2 + 1023) / 1024) * 1024;
  constexpr int PART_BYTES = RWT == 1 ? 0 : (TILE_ROWS / 32) * CW * 32 * 16;
  constexpr int BUF_BYTES = A_BYTES + C_BYTES;
  const int lds_bytes = 2 * BUF_BYTES + PART_BYTES;
  static_assert(2 * BUF_BYTES + PART_BYTES <= 160 * 1024, "LDS budget");
  a.TP = TILE_ROWS / a.kp;
  if (a.TP < 1) {
    set_error("gcn_edgeconv_fwd: k=%d needs %d rows per point > tile of %d rows", a.k, a.kp, TILE_ROWS);
    return GCN_EINVAL;
  }
  if (RWT == 1 && a.TP > 4) a.TP = 4;
  a.tiles_per_cloud = (a.N + a.TP - 1) / a.TP;
  a.total_tiles = a.B * a.tiles_per_cloud;
  // persistent grid: as many workgroups as stay resident (2 waves per SIMD = 8 waves per CU)
  const int wg_per_cu = 8 / (RWT * CW) > 0 ? 8 / (RWT * CW) : 1;
  const int by_lds = (160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1);
  const int resident = 256 * (wg_per_cu < by_lds ? wg_per_cu : by_lds);
  int grid = a.total_tiles < resident ? a.total_tiles : resident;
  using kern_t = void (*)(EcArgs);
  // k = 64 (the configuration everything is tuned on) gets the padded k as a compile-time constant
  const bool kp64 = RWT == 1 && a.kp == 64;
  kern_t kern;
  if (kp64)
    kern = a.gamma_route ? (with_arg ? (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, true, true, 64> : (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, false, true, 64>)
                         : (with_arg ? (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, true, false, 64> : (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, false, false, 64>);
  else
    kern = a.gamma_route ? (with_arg ? (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, true, true, 0> : (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, false, true, 0>)
                         : (with_arg ? (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, true, false, 0> : (kern_t)edgeconv_fwd_bf16_kernel<KSTEPS, CW, RWT, false, false, 0>);
  GCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  kern<<<grid, 64 * RWT * CW, lds_bytes, st>>>(a);
  return check_launch("edgeconv_fwd_bf16_kernel");
}

}  // namespace gcn

using namespace gcn;

static int padded_channels(int C) {
  int cp = 16;                                     // one MFMA k-step (16 bf16) per row at least
  while (cp < C) cp <<= 1;
  return cp;
}

GCN_EXPORT int gcn_edgeconv_padded_channels(int C) { return padded_channels(C); }

GCN_EXPORT int gcn_edgeconv_pack_x16(const float *x_cm, int B, int C, int N, void *x_pm_16, float *x_pm_f32, int half,
                                     void *stream) {
  GCN_REQUIRE(x_cm && (x_pm_16 || x_pm_f32), "gcn_edgeconv_pack_x: null pointer");
  GCN_REQUIRE(B >= 0 && C >= 1 && N >= 1 && (half == 0 || half == 1), "gcn_edgeconv_pack_x: bad shape");
  if (B == 0) return GCN_OK;
  const int Cp = padded_channels(C);
  pack_x_kernel<<<dim3(cdiv(N, 32), cdiv(Cp, 32), B), 256, 0, (hipStream_t)stream>>>(x_cm, C, N, Cp, (unsigned short *)x_pm_16, x_pm_f32, half);
  return check_launch("pack_x_kernel");
}
GCN_EXPORT int gcn_edgeconv_pack_x(const float *x_cm, int B, int C, int N, void *x_pm_bf16, float *x_pm_f32, void *stream) {
  return gcn_edgeconv_pack_x16(x_cm, B, C, N, x_pm_bf16, x_pm_f32, 0, stream);
}

GCN_EXPORT int gcn_edgeconv_pack_w16(const float *w, int Cout, int C, void *wp_16, int half, void *stream) {
  GCN_REQUIRE(w && wp_16, "gcn_edgeconv_pack_w: null pointer");
  GCN_REQUIRE(Cout >= 1 && C >= 1 && (half == 0 || half == 1), "gcn_edgeconv_pack_w: bad shape");
  const int Cp = padded_channels(C);
  pack_w_kernel<<<cdiv((long)Cout * 2 * Cp, 256), 256, 0, (hipStream_t)stream>>>(w, Cout, C, Cp, (unsigned short *)wp_16, half);
  return check_launch("pack_w_kernel");
}
GCN_EXPORT int gcn_edgeconv_pack_w(const float *w, int Cout, int C, void *wp_bf16, void *stream) {
  return gcn_edgeconv_pack_w16(w, Cout, C, wp_bf16, 0, stream);
}

GCN_EXPORT int gcn_edgeconv_fwd(const void *x_pm, const void *w, const int64_t *idx, int dtype, int B, int N, int NX,
                                int C, int k, int Cout, int G, const float *q, float *ymax, float *ymin, uint8_t *amax,
                                uint8_t *amin, double *gsum, const float *gamma_route, void *stream) {
  GCN_REQUIRE(x_pm && w && idx && ymax && gsum, "gcn_edgeconv_fwd: null pointer");
  GCN_REQUIRE(dtype >= 1 || q == nullptr, "gcn_edgeconv_fwd: q belongs to the matrix-core paths (dtype 1, 2) only");
  GCN_REQUIRE(gamma_route || ymin, "gcn_edgeconv_fwd: ymin may be NULL only in routed mode (gamma_route given)");
  GCN_REQUIRE(gamma_route || (amax == nullptr) == (amin == nullptr), "gcn_edgeconv_fwd: pass both amax and amin or neither");
  GCN_REQUIRE(dtype == 0 || dtype == 1 || dtype == 2, "gcn_edgeconv_fwd: dtype must be 0 (f32), 1 (bf16) or 2 (IEEE half)");
  GCN_REQUIRE(B >= 0 && N >= 1 && NX >= N && C >= 1 && k >= 1 && k <= 255, "gcn_edgeconv_fwd: bad shape (need NX >= N, 1 <= k <= 255)");
  GCN_REQUIRE(G >= 1 && Cout % G == 0, "gcn_edgeconv_fwd: Cout=%d not divisible by G=%d", Cout, G);
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(gsum, sizeof(double) * 2 * B * G, st));
  if (dtype == 0) {
    const size_t lds = sizeof(float) * ((size_t)k * 2 * C + 2) + sizeof(double) * 2 * Cout;
    GCN_REQUIRE(lds <= 150 * 1024, "gcn_edgeconv_fwd(f32): k*2C too large for the exact path (%zu B LDS)", lds);
    GCN_HIP(hipFuncSetAttribute((const void *)edgeconv_fwd_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    edgeconv_fwd_f32_kernel<<<dim3(N, B), 256, lds, st>>>((const float *)x_pm, (const float *)w, idx, N, NX, C, k, Cout, G,
                                                          ymax, ymin, amax, amin, gsum, gamma_route);
    return check_launch("edgeconv_fwd_f32_kernel");
  }
  GCN_REQUIRE(Cout == 64 || Cout == 128, "gcn_edgeconv_fwd(bf16): Cout must be 64 or 128, got %d", Cout);
  GCN_REQUIRE((Cout / G) % 32 == 0, "gcn_edgeconv_fwd(bf16): Cout/G must be a multiple of 32");
  const int Cp = padded_channels(C);
  GCN_REQUIRE(Cp <= 128 || (Cp == 256 && Cout == 128 && (k + 31) / 32 * 32 <= 128),
              "gcn_edgeconv_fwd(bf16): C=%d unsupported (C <= 128, or C <= 256 with Cout == 128 and k <= 128)", C);
  EcArgs a{};
  a.x = (const unsigned short *)x_pm; a.wp = (const unsigned short *)w; a.idx = idx;
  a.B = B; a.N = N; a.NX = NX; a.k = k; a.kp = (k + 31) / 32 * 32; a.Cout = Cout; a.G = G;
  a.ymax = ymax; a.ymin = ymin; a.amax = amax; a.amin = amin; a.gsum = gsum; a.gamma_route = gamma_route;
  const bool wa = amax != nullptr;
  if (a.kp <= 128) {      // the normal case: x_j half on the matrix cores, centre term q precomputed per point
    EcqArgs e{};
    e.x = a.x; e.wp = a.wp; e.idx = idx; e.q = q;
    e.B = B; e.N = N; e.NX = NX; e.k = k; e.Cout = Cout; e.G = G;
    e.ymax = ymax; e.ymin = ymin; e.amax = amax; e.amin = amin; e.gsum = gsum; e.gamma_route = gamma_route;
    return dtype == 2 ? launch_edgeconv_fwd_q_f16(e, Cp, wa, st) : launch_edgeconv_fwd_q(e, Cp, wa, st);
  }
  GCN_REQUIRE(dtype == 1, "gcn_edgeconv_fwd(f16): k <= 128 only");
  // 128 < k <= 255: two 128-row groups per point, full [x_j ; x_i] rows (q is not needed)
  const int ks = Cp / 8;
#define EC_CASE(KS, CWV) \
  if (ks == KS && Cout == CWV * 32) return launch_fwd_bf16<KS, CWV, 2>(a, wa, st);
  EC_CASE(2, 2) EC_CASE(4, 2) EC_CASE(8, 2) EC_CASE(16, 2)
  EC_CASE(2, 4) EC_CASE(4, 4) EC_CASE(8, 4) EC_CASE(16, 4)
#undef EC_CASE
  set_error("gcn_edgeconv_fwd(bf16): unsupported configuration");
  return GCN_EINVAL;
}

GCN_EXPORT int gcn_edgeconv_finish(const float *ymax, const float *ymin, const double *gsum, const float *gamma,
                                   const float *beta, int B, int N, int k, int Cout, int G, float eps, float slope,
                                   float *out_cm, float *out_pm, float *mean_rstd, void *out_pm_bf16, int bf16_pitch,
                                   void *stream) {
  GCN_REQUIRE(ymax && gsum && gamma && beta && (out_cm || out_pm || out_pm_bf16), "gcn_edgeconv_finish: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && k >= 1 && G >= 1 && Cout % G == 0, "gcn_edgeconv_finish: bad shape");
  GCN_REQUIRE(!out_pm_bf16 || bf16_pitch >= Cout, "gcn_edgeconv_finish: bf16 row pitch %d < Cout %d", bf16_pitch, Cout);
  if (B == 0) return GCN_OK;
  edgeconv_finish_kernel<<<dim3(cdiv(N, 32), cdiv(Cout, 32), B), 256, 0, (hipStream_t)stream>>>(
      ymax, ymin, gsum, gamma, beta, N, k, Cout, G, eps, slope, out_cm, out_pm, mean_rstd, (unsigned short *)out_pm_bf16,
      bf16_pitch);
  return check_launch("edgeconv_finish_kernel");
}

GCN_EXPORT int gcn_neighbor_sum(const float *x_pm, const int64_t *idx, int B, int N, int C, int k, float *s, void *stream) {
  GCN_REQUIRE(x_pm && idx && s, "gcn_neighbor_sum: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && C >= 1 && k >= 1, "gcn_neighbor_sum: bad shape");
  if (B == 0) return GCN_OK;
  GCN_REQUIRE((long)N * C < (1L << 32), "gcn_neighbor_sum: N*C must fit 32 bits");
  hipStream_t st = (hipStream_t)stream;
  if (C <= 8) neighbor_sum_kernel<8><<<dim3(cdiv(N, 4 * 8), B), 256, 0, st>>>(x_pm, idx, N, C, k, s);
  else if (C <= 16) neighbor_sum_kernel<16><<<dim3(cdiv(N, 4 * 4), B), 256, 0, st>>>(x_pm, idx, N, C, k, s);
  else if (C <= 32) neighbor_sum_kernel<32><<<dim3(cdiv(N, 4 * 2), B), 256, 0, st>>>(x_pm, idx, N, C, k, s);
  else neighbor_sum_kernel<64><<<dim3(cdiv(N, 4), B), 256, 0, st>>>(x_pm, idx, N, C, k, s);
  return check_launch("neighbor_sum_kernel");
}

namespace gcn {   // rsum.hip: stage + sort + gather form of the transposed aggregation
bool rsum_staged_supported(int B, int N, int C, int k);
size_t rsum_staged_ws_bytes(int B, int N, int C, int k);
int run_reverse_sum_staged(const float *x, const int64_t *idx, int B, int N, int C, int k, float *r, float *indeg, void *ws,
                           hipStream_t st);
}

GCN_EXPORT long gcn_reverse_sum_ws_bytes(int B, int N, int C, int k) {
  if (B < 0 || N < 1 || k < 1 || C < 0) return -1;
  long need = 256 + (N <= 65536 ? 2L * B * N * k : 0);
  if (rsum_staged_supported(B, N, C, k)) need = std::max(need, (long)rsum_staged_ws_bytes(B, N, C, k));
  return need;
}

GCN_EXPORT int gcn_reverse_sum(const float *x_pm, const int64_t *idx, int B, int N, int C, int k, float *r, float *indeg,
                               void *ws, void *stream) {
  GCN_REQUIRE(x_pm && idx && ws && (r || indeg), "gcn_reverse_sum: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && C >= 1 && k >= 1, "gcn_reverse_sum: bad shape");
  if (!r) C = 0;                                   // in-degrees only: no rows are read or accumulated
  GCN_REQUIRE(C <= 2048, "gcn_reverse_sum: C=%d too wide for one LDS row block", C);
  GCN_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)idx & 15) == 0, "gcn_reverse_sum: ws/idx must be 16-B aligned");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(ws, 256, st));          // max |x| bits + the staged path's overflow counters
  if (C > 0) absmax_kernel<<<256, 256, 0, st>>>(x_pm, (long)B * N * C, (unsigned int *)ws);
  if (C > 0 && rsum_staged_supported(B, N, C, k)) return run_reverse_sum_staged(x_pm, idx, B, N, C, k, r, indeg, ws, st);
  // destination rows per workgroup: ~256 workgroups in total, bounded by 128 KB of LDS
  int R = (int)(((long)N * B + 255) / 256);
  const int rmax = (128 * 1024) / (8 * C + 4);
  if (R > rmax) R = rmax;
  if (R < 1) R = 1;
  const size_t lds = (size_t)R * C * 8 + (size_t)R * 4;
  const int grid = cdiv(N, R) * B;
  if (N <= 65536) {
    unsigned short *i16 = reinterpret_cast<unsigned short *>((char *)ws + 256);
    const long E = (long)B * N * k;
    idx_to_u16_kernel<<<cdiv((E + 1) / 2, 256), 256, 0, st>>>(idx, E, i16);
    if (C == 0 && ((long)N * k) % 8 == 0) {                  // in-degrees only
      int Rc = (int)(((long)N * B + 255) / 256);
      if (Rc < 1) Rc = 1;
      indeg_lds_kernel<<<cdiv(N, Rc) * B, 1024, (size_t)Rc * 4, st>>>(i16, B, N, k, Rc, indeg);
      return check_launch("indeg_lds_kernel");
    }
    if (C == 64) {
      GCN_HIP(hipFuncSetAttribute((const void *)reverse_sum_lds_kernel<unsigned short, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      reverse_sum_lds_kernel<unsigned short, true><<<grid, 1024, lds, st>>>(x_pm, i16, (const unsigned int *)ws, B, N, C, k, R, r, indeg);
    } else {
      GCN_HIP(hipFuncSetAttribute((const void *)reverse_sum_lds_kernel<unsigned short, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      reverse_sum_lds_kernel<unsigned short, false><<<grid, 1024, lds, st>>>(x_pm, i16, (const unsigned int *)ws, B, N, C, k, R, r, indeg);
    }
  } else {
    GCN_HIP(hipFuncSetAttribute((const void *)reverse_sum_lds_kernel<int64_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    reverse_sum_lds_kernel<int64_t, false><<<grid, 1024, lds, st>>>(x_pm, idx, (const unsigned int *)ws, B, N, C, k, R, r, indeg);
  }
  return check_launch("reverse_sum_lds_kernel");
}

GCN_EXPORT int gcn_keyedge_fwd(const float *att, const int64_t *kidx, const float *U, const float *V, int B, int N, int k,
                               int NK, int Cout, int G, float *ymax, float *ymin, uint8_t *amax, uint8_t *amin,
                               double *gsum, const float *gamma_route, void *stream) {
  GCN_REQUIRE(att && kidx && U && V && ymax && gsum, "gcn_keyedge_fwd: null pointer");
  GCN_REQUIRE(gamma_route || ymin, "gcn_keyedge_fwd: ymin may be NULL only in routed mode (gamma_route given)");
  GCN_REQUIRE(gamma_route || (amax == nullptr) == (amin == nullptr), "gcn_keyedge_fwd: pass both amax and amin or neither");
  GCN_REQUIRE(B >= 0 && N >= 1 && k >= 1 && k <= 255 && NK >= 1 && Cout >= 1 && G >= 1 && G <= 64 && Cout % G == 0, "gcn_keyedge_fwd: bad shape");
  const size_t lds = sizeof(float) * (((size_t)NK * Cout + 3) & ~(size_t)3) + 16 * 64 * sizeof(float2);    // key table + per-wave (weight, id) slots
  GCN_REQUIRE(lds <= 148 * 1024, "gcn_keyedge_fwd: key table %zu B exceeds LDS", lds);
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(gsum, sizeof(double) * 2 * B * G, st));
  int blocks_per_cloud = (256 + B - 1) / B;               // one 16-wave block per CU (the key table is 61 KB)
  if (blocks_per_cloud > (N + 15) / 16) blocks_per_cloud = (N + 15) / 16;
  const int ppb = (N + blocks_per_cloud - 1) / blocks_per_cloud;
  const dim3 grid(cdiv(N, ppb), B);
#define GCN_KE_FWD(NCHV, RV)                                                                                        \
  {                                                                                                                 \
    GCN_HIP(hipFuncSetAttribute((const void *)keyedge_fwd_kernel<NCHV, RV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    keyedge_fwd_kernel<NCHV, RV><<<grid, 1024, lds, st>>>(att, kidx, U, V, N, k, NK, Cout, G, ppb, ymax, ymin, amax, amin, gsum, gamma_route); \
  }
  if (Cout % 128 == 0) {
    if (gamma_route) GCN_KE_FWD(2, true) else GCN_KE_FWD(2, false)
  } else {
    if (gamma_route) GCN_KE_FWD(1, true) else GCN_KE_FWD(1, false)
  }
#undef GCN_KE_FWD
  return check_launch("keyedge_fwd_kernel");
}

GCN_EXPORT int gcn_cast_pad16(const float *x_pm, long rows, int C, void *x_pm_16, int half, void *stream) {
  GCN_REQUIRE(x_pm && x_pm_16, "gcn_cast_pad_bf16: null pointer");
  GCN_REQUIRE(rows >= 0 && C >= 1 && (half == 0 || half == 1), "gcn_cast_pad_bf16: bad shape");
  if (rows == 0) return GCN_OK;
  const int Cp = padded_channels(C);
  cast_pad_bf16_kernel<<<cdiv(rows * Cp, 256), 256, 0, (hipStream_t)stream>>>(x_pm, rows, C, Cp, (unsigned short *)x_pm_16, half);
  return check_launch("cast_pad_bf16_kernel");
}
GCN_EXPORT int gcn_cast_pad_bf16(const float *x_pm, long rows, int C, void *x_pm_bf16, void *stream) {
  return gcn_cast_pad16(x_pm, rows, C, x_pm_bf16, 0, stream);
}

GCN_EXPORT int gcn_keyedge_bwd(const float *att, const int64_t *kidx, const float *U, const float *V, const float *coef,
                               const int64_t *jsel, const float *Ac, const float *Bc, const float *X, int B, int N, int k,
                               int NK, int Cout, float *datt, float *dV, float *A2, float *dUsp, float *T12, void *stream) {
  GCN_REQUIRE(att && kidx && U && V && coef && jsel && Ac && Bc && X && datt && dV && A2 && dUsp && T12,
              "gcn_keyedge_bwd: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && k >= 1 && k <= 64 && NK >= 1 && Cout >= 1, "gcn_keyedge_bwd: bad shape (need k <= 64)");
  const int NKp = (NK + 63) & ~63;
  const size_t lds = sizeof(float) * (2 * (((size_t)NK * Cout + 3) & ~(size_t)3) + 6 * NKp + 16 * (256 + NKp));
  GCN_REQUIRE(lds <= 158 * 1024, "gcn_keyedge_bwd: key tables %zu B exceed LDS", lds);
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_spans(st, {dUsp, sizeof(float) * (size_t)B * NK * Cout}, {T12, sizeof(float) * (size_t)B * 2 * NK}));
  GCN_HIP(hipFuncSetAttribute((const void *)keyedge_bwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  GCN_HIP(hipFuncSetAttribute((const void *)keyedge_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int blocks_per_cloud = (256 + B - 1) / B;
  if (blocks_per_cloud > (N + 3) / 4) blocks_per_cloud = (N + 3) / 4;
  const int ppb = (N + blocks_per_cloud - 1) / blocks_per_cloud;
  if (Cout % 128 == 0)
    keyedge_bwd_kernel<2><<<dim3(cdiv(N, ppb), B), 1024, lds, st>>>(att, kidx, U, V, coef, jsel, Ac, Bc, X, N, k, NK, Cout, ppb,
                                                                     datt, dV, A2, dUsp, T12);
  else
    keyedge_bwd_kernel<1><<<dim3(cdiv(N, ppb), B), 1024, lds, st>>>(att, kidx, U, V, coef, jsel, Ac, Bc, X, N, k, NK, Cout, ppb,
                                                                     datt, dV, A2, dUsp, T12);
  return check_launch("keyedge_bwd_kernel");
}
