// edgeconv_fwd_impl.h -- (kernels of edgeconv_fwd.hip [bf16 operands] and edgeconv_fwd_f16.hip [IEEE half])
// the grouped (N*k, 2C) x (2C, Cout) contraction of the DGCNN EdgeConv block
// (get_graph_feature M4:93-124 + Conv2d 1x1 + the statistics/extreme half of GroupNorm + LeakyReLU + max over k,
// models/dgcnn-hais-concat-direct-4.py:463-505) for gfx950, bf16 MFMA, k <= 128, up to 256 input channels (BASELINE
// configs[4] names C = 256: KS = 16 k-steps, a 64-KB tile per buffer, one workgroup per CU).
//
// Formulation.  The reference's row is e = [x_j - x_i ; x_i] with W = [W1 | W2]:  y[n,j] = W1.x_j + (W2 - W1).x_i.
// The centre term q[n] = (W2 - W1).x_n is the same for the k rows of a point, so it is contracted ONCE per point
// (edgeconv_center_kernel, a (B*N, C) x (C, Cout) MFMA GEMM, k-fold cheaper than the grouped part) and enters the
// grouped contraction as a per-(point, channel) constant of the k rows -- the matrix cores only run the x_j half,
// K = Cp instead of 2*Cp, with the same algorithmic result.  Because adding a constant is monotone, q is applied
// AFTER the reduction over k:  max_k (s_k + q) = (max_k s_k) + q bitwise, and the GroupNorm sums follow from
// sum_k (s_k + q) = sum_k s_k + k q,  sum_k (s_k + q)^2 = sum_k s_k^2 + 2 q sum_k s_k + k q^2.
//
// The kernel is VALU-issue bound, not MFMA bound (an MFMA holds the SIMD's vector issue for 8 of its 32 cycles;
// every accumulator element needs sum, sum of squares, max and arg-max), so the structure is built around the
// VALU count per tile: fragment addresses, neighbour-id offsets and DMA chunk offsets are per-lane constants hoisted
// out of the persistent loop (ds_read immediates select row block and buffer), tile coordinates advance on the SALU.
//
// Geometry (wave64, v_mfma_f32_32x32x16_bf16):
//   tile      = 128 edge rows = TP points x KP rows (KP = k rounded up to 32; padded slots repeat neighbour 0, which
//               cannot change max/min and is masked out of the sums)
//   workgroup = CW = Cout/32 waves; a wave owns all 128 rows x 32 columns (4 accumulator blocks); W1 fragments live
//               in registers for the whole kernel, the LDS only streams A
//   A in LDS  = [128][Cp] bf16, 16-B chunks XOR-swizzled on the DMA *source* side (conflict-free ds_read_b128),
//               double buffered: tile t+1 is gathered by global_load_lds_dwordx4 while tile t is on the MFMAs
// The two .hip files that include this header carry the `hipcc-flags: -fno-honor-nans` line.
#pragma once
#include "common.h"
#include "edgeconv_fwd.h"

#include <type_traits>

namespace gcn {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// one 32x32x16 matrix step on 16-bit operand fragments: bf16, or IEEE half (F16) -- the operand images differ, the kernel
// around them does not
template <bool F16>
__device__ __forceinline__ f32x16 ec_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// max / min of a value over the two half-waves (lane l and lane l ^ 32), in every lane.  v_permlane32_swap exchanges
// the upper half of its first operand with the lower half of its second (tools/micro/permlane_swap_test.hip); called
// with the same value twice, one result holds the partner's value and the other the lane's own, and the reductions
// take both.  hipcc (ROCm 7.2) folds such a reduction over swap(v, v) down to the lane's own value (seen with r[1]
// alone, with fmax(fmax(v, r[0]), r[1]), and again when a result was written back into the result vector), so the
// second operand and the second result each pass through an empty asm and stay scalars.
struct HalfPair { unsigned int own_or_partner, partner_or_own; };
__device__ __forceinline__ HalfPair half_swap(unsigned int u) {
  unsigned int w = u;
  asm volatile("" : "+v"(w));
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
  unsigned int r0 = r[0], r1 = r[1];
  asm volatile("" : "+v"(r1));
  return HalfPair{r0, r1};
}
__device__ __forceinline__ float both_halves_max(float v) {
  const HalfPair r = half_swap(__builtin_bit_cast(unsigned int, v));
  return __builtin_fmaxf(__builtin_fmaxf(v, __builtin_bit_cast(float, r.own_or_partner)), __builtin_bit_cast(float, r.partner_or_own));
}
__device__ __forceinline__ float both_halves_min(float v) {
  const HalfPair r = half_swap(__builtin_bit_cast(unsigned int, v));
  return __builtin_fminf(__builtin_fminf(v, __builtin_bit_cast(float, r.own_or_partner)), __builtin_bit_cast(float, r.partner_or_own));
}
__device__ __forceinline__ int both_halves_min_i(int v) {
  const HalfPair r = half_swap((unsigned int)v);
  return min(min(v, (int)r.own_or_partner), (int)r.partner_or_own);
}

// v_max3_f32 / v_min3_f32.  This file is compiled with -fno-honor-nans (see the hipcc-flags line at the end of the
// header comment): without it the compiler puts a canonicalising `v_max x, x` in front of every fmaxf on an MFMA result
// (IEEE mode quiets signalling NaNs), ~30 VALU per tile.  Inline asm is NOT an option here: the hazard recogniser
// does not see an asm statement read an MFMA result and leaves out the wait states the matrix pipe needs.
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float min3f(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }

// ------------------------------------------------------------------ centre term q = x . (W2 - W1)^T
// x (R, Cp) bf16 point-major, wp (Cout, 2Cp) bf16 = [W1 | W2 - W1] -> q (R, Cout) f32.  One wave = 32 rows x Cout.
template <int KS, int CW, bool F16>
__global__ __launch_bounds__(256) void edgeconv_center_kernel(const unsigned short *__restrict__ x,
                                                              const unsigned short *__restrict__ wp, long R,
                                                              float *__restrict__ q) {
  constexpr int CP = KS * 16;
  const int lane = lane_id();
  const int lr = lane & 31, lh = lane >> 5;
  const long r0 = ((long)blockIdx.x * 4 + wave_id()) * 32;
  if (r0 >= R) return;
  long row = r0 + lr;
  if (row >= R) row = R - 1;
  f32x16 acc[CW];
#pragma unroll
  for (int cb = 0; cb < CW; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[cb][i] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(x + row * CP + s * 16 + lh * 8);
#pragma unroll
    for (int cb = 0; cb < CW; ++cb) {
      const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(wp + (long)(cb * 32 + lr) * (2 * CP) + CP + s * 16 + lh * 8);
      acc[cb] = ec_mfma<F16>(af, bf, acc[cb]);
    }
  }
#pragma unroll
  for (int cb = 0; cb < CW; ++cb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const long r = r0 + 4 * lh + (i & 3) + 8 * (i >> 2);
      if (r < R) q[r * (CW * 32) + cb * 32 + lr] = acc[cb][i];
    }
}

// ------------------------------------------------------------------ grouped contraction
// KS = Cp/16 k-steps, CW = Cout/32 waves, NB = KP/32 accumulator blocks per point (1, 2, 3, 4).
template <int KS, int CW, int NB, bool KEXACT, bool WITH_ARG, bool ROUTED, bool F16>
__global__ __launch_bounds__(64 * CW, KS > 8 ? 1 : 2) void edgeconv_fwd_q_kernel(EcqArgs a) {
  constexpr int CP = KS * 16;
  constexpr int NC = 2 * KS;               // 16-B chunks per x row
  constexpr int ROW_BYTES = CP * 2;
  constexpr int KP = NB * 32;
  constexpr int TP = NB == 3 ? 1 : 4 / NB; // points per 128-row tile
  constexpr int ROWS_USED = TP * KP;
  constexpr int RPP = 64 / NC;             // rows per 1-KiB DMA piece
  constexpr int PIECES = 128 / RPP;
  constexpr int PPW = PIECES / CW;         // pieces per wave (PIECES in {4,8,16,32}, CW in {2,4})
  constexpr int RPB = NC >= 16 ? 1 : 16 / NC;  // rows per 256-B bank row
  constexpr int A_BYTES = 128 * ROW_BYTES;
  constexpr int COUT = CW * 32;
  static_assert(PIECES % CW == 0 && PPW >= 1, "piece split");
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int lane = lane_id();
  const int cg = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  const int col = cg * 32 + lr;

  // ---- W1 fragments in registers for the whole kernel
  bf16x8 breg[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    breg[s] = *reinterpret_cast<const bf16x8 *>(a.wp + (long)col * (2 * CP) + s * 16 + lh * 8);
  // routed mode: sgn = -1 turns the min into a max of the NEGATED column; the sign is folded into this lane's W1
  // column once (bf16 sign bits; products and sums negate exactly)
  const float sgn = ROUTED ? (a.gamma_route[col] >= 0.f ? 1.f : -1.f) : 1.f;
  if (ROUTED && sgn < 0.f) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      unsigned int *w4 = reinterpret_cast<unsigned int *>(&breg[s]);
#pragma unroll
      for (int i = 0; i < 4; ++i) w4[i] ^= 0x80008000u;
    }
  }

  // Workgroup g runs on XCD g % 8: give the workgroups of one XCD consecutive tile ranges (one cloud per XCD at 8
  // clouds), so a cloud's rows stay in that XCD's L2 across the k-fold gathers.
  const int GD = gridDim.x;
  const int vb = (GD % 8 == 0) ? (int)(blockIdx.x % 8) * (GD / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int t_begin = (int)((long)vb * a.total_tiles / GD);
  const int t_end = (int)((long)(vb + 1) * a.total_tiles / GD);
  if (t_begin >= t_end) return;
  const int tpc = a.tiles_per_cloud, N = a.N, k = a.k;

  // ---- per-lane constants of the DMA pieces: chunk offset in the source row, byte offset of the neighbour id
  // relative to the tile's first point, and the (point, slot) the piece row belongs to
  unsigned int coff[PPW], idoff[PPW];
  int ppt[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = cg + i * CW;
    const int row = p * RPP + lane / NC;
    const int cs = lane % NC;
    coff[i] = (unsigned int)((cs ^ ((row / RPB) & (NC - 1))) * 16);
    int pt = row / KP, j = row % KP;
    if (pt >= TP) pt = TP - 1;                         // NB == 3: rows 96..127 are never gathered
    if (!KEXACT && j >= k) j = 0;
    ppt[i] = pt;
    idoff[i] = (unsigned int)((pt * k + j) * 8);
  }
  // ---- per-lane fragment addresses: row lr of block 0 in buffer 0; block and buffer are immediates
  unsigned int arow[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) arow[s] = (unsigned int)(lr * ROW_BYTES + (((2 * s + lh) ^ ((lr / RPB) & (NC - 1))) << 4));
  // rows of the point's LAST block that are real neighbours (!KEXACT): register i is row rlast + off(i)
  const int rlast = (NB - 1) * 32 + 4 * lh;
  float nv_last = 16.f;
  if (!KEXACT) {
    int c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) c += (rlast + (i & 3) + 8 * (i >> 2)) < k ? 1 : 0;
    nv_last = (float)c;
  }

  // tile coordinates (cloud b, tile-in-cloud m) of tiles t, t+1, t+2: advanced on the SALU, clamped at t_end - 1
  int b0 = t_begin / tpc, m0 = t_begin % tpc;
  auto step = [&](int &b, int &m, int t) {             // (b, m) of tile t -> tile t + 1 (or unchanged at the end)
    if (t + 1 < t_end) {
      ++m;
      if (m == tpc) { m = 0; ++b; }
    }
  };
  int b1 = b0, m1 = m0;
  step(b1, m1, t_begin);
  int b2 = b1, m2 = m1;
  step(b2, m2, t_begin + 1);

  int grow[PPW];
  float qn[TP];
  auto load_ids = [&](int b, int m) {
    const int n0 = m * TP;
    const unsigned char *base = reinterpret_cast<const unsigned char *>(a.idx) + ((long)b * N + n0) * k * 8;
    if (n0 + TP <= N) {
#pragma unroll
      for (int i = 0; i < PPW; ++i) grow[i] = *reinterpret_cast<const int *>(base + idoff[i]);   // low dword of the int64 id
    } else {                                             // last tile of a cloud with N % TP != 0: clamp the point
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int pt = n0 + ppt[i] < N ? ppt[i] : 0;
        grow[i] = *reinterpret_cast<const int *>(base + (idoff[i] - (unsigned int)((ppt[i] - pt) * k * 8)));
      }
    }
  };
  auto load_q = [&](int b, int m) {
    const int n0 = m * TP;
#pragma unroll
    for (int pt = 0; pt < TP; ++pt) {
      const int n = n0 + pt < N ? n0 + pt : N - 1;
      qn[pt] = a.q ? *reinterpret_cast<const float *>(reinterpret_cast<const unsigned char *>(a.q + ((long)b * N + n) * COUT) + (unsigned int)(col * 4)) : 0.f;
    }
  };
  auto issue_gather = [&](int b, int buf) {
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(a.x) + (long)b * a.NX * ROW_BYTES;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int p = cg + i * CW;
      if (p * RPP < ROWS_USED) {
        const unsigned char *src = xb + ((unsigned int)grow[i] * (unsigned int)ROW_BYTES + coff[i]);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds + buf * A_BYTES + p * 1024), 16, 0, 0);
      }
    }
  };

  float s1 = 0.f, s2 = 0.f;
  int cur_b = b0;
  const int cpg = a.Cout / a.G;  // channels per group (multiple of 32 -> a wave's columns are in one group)
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

  load_ids(b0, m0);
  load_q(b0, m0);
  issue_gather(b0, 0);
  load_ids(b1, m1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

  auto tile = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    if (b0 != cur_b) {
      flush_stats(cur_b);
      cur_b = b0;
    }
    float qc[TP];
#pragma unroll
    for (int pt = 0; pt < TP; ++pt) qc[pt] = qn[pt];
    if (t + 1 < t_end) issue_gather(b1, BUF ^ 1);       // ids were fetched one tile ago
    load_ids(b2, m2);                                   // consumed next iteration
    load_q(b1, m1);
    const int n0 = m0 * TP;

    // ---- K loop: the fragments of k-step s+1 are in flight while the MFMAs of step s issue
    constexpr int NRB = (ROWS_USED + 31) / 32;          // accumulator blocks in use (3 when k in (64, 96])
    f32x16 acc[NRB];
    bf16x8 af[2][NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
      af[0][rb] = *reinterpret_cast<const bf16x8 *>(lds + arow[0] + (BUF * A_BYTES + rb * 32 * ROW_BYTES));
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
          af[(s + 1) & 1][rb] = *reinterpret_cast<const bf16x8 *>(lds + arow[s + 1] + (BUF * A_BYTES + rb * 32 * ROW_BYTES));
      }
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
        acc[rb] = ec_mfma<F16>(af[s & 1][rb], breg[s], s == 0 ? zero16 : acc[rb]);
    }

    // ---- epilogue, straight-line over the blocks so that their dependency chains interleave: per (block, column)
    // sum and sum of squares (GroupNorm), max / min over the 32 rows, and the row attaining it.  Blocks of a point
    // past the end of the cloud (tail tile) are computed like the others (their rows are clamped copies) and masked.
    float vld[TP];                                       // 1 for a real point, 0 past the end of the cloud
#pragma unroll
    for (int pt = 0; pt < TP; ++pt) vld[pt] = n0 + pt < N ? 1.f : 0.f;
    float ps[NRB], pq[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) { ps[rb] = 0.f; pq[rb] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        float v = acc[rb][i];
        if (!KEXACT && (rb % NB) == NB - 1) v = (rlast + (i & 3) + 8 * (i >> 2)) < k ? v : 0.f;
        ps[rb] += v;
        pq[rb] = fmaf(v, v, pq[rb]);
      }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const float nv = (!KEXACT && (rb % NB) == NB - 1) ? nv_last : 16.f;
      const float q = qc[rb / NB];
      const float sp = ROUTED ? ps[rb] * sgn : ps[rb];  // acc holds sgn * s in routed mode
      const float ty = fmaf(nv, q, sp);                 // sum of y = s + q over this lane's rows
      s1 = fmaf(vld[rb / NB], ty, s1);
      s2 = fmaf(vld[rb / NB], fmaf(q, sp + ty, pq[rb]), s2);   // sum y^2 = sum s^2 + 2 q sum s + nv q^2
    }
    // extremes: lane-level max3 chains, then the other half-wave's value (rows 4h.. of each register live in the two
    // halves), then the point's NB blocks.  both_halves_* are symmetric in the two swap results, so every lane ends
    // up with the extreme over all KP rows of its (point, column).
    float bmx[NRB], bmn[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      bmx[rb] = max3f(acc[rb][0], acc[rb][1], acc[rb][2]);
      if (!ROUTED) bmn[rb] = min3f(acc[rb][0], acc[rb][1], acc[rb][2]);
    }
#pragma unroll
    for (int i = 3; i < 15; i += 2)
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) {
        bmx[rb] = max3f(bmx[rb], acc[rb][i], acc[rb][i + 1]);
        if (!ROUTED) bmn[rb] = min3f(bmn[rb], acc[rb][i], acc[rb][i + 1]);
      }
    float pmx[TP], pmn[TP];
#pragma unroll
    for (int pt = 0; pt < TP; ++pt) {
#pragma unroll
      for (int q_ = 0; q_ < NB; ++q_) {
        const int rb = pt * NB + q_;
        const float m = __builtin_fmaxf(bmx[rb], acc[rb][15]);
        pmx[pt] = q_ == 0 ? m : __builtin_fmaxf(pmx[pt], m);
        if (!ROUTED) {
          const float n_ = __builtin_fminf(bmn[rb], acc[rb][15]);
          pmn[pt] = q_ == 0 ? n_ : __builtin_fminf(pmn[pt], n_);
        }
      }
      pmx[pt] = both_halves_max(pmx[pt]);
      if (!ROUTED) pmn[pt] = both_halves_min(pmn[pt]);
    }
    // the row attaining it: lowest register index in this lane that equals the point's extreme (64 = none, which
    // maps to a row >= 128), lowest row over the point's blocks and the two halves -> lowest slot on ties.  Padded
    // slots repeat slot 0 and sit at higher rows, so the result is always a real neighbour slot (< k).
    int pax[TP], pan[TP];
    if (WITH_ARG) {
      int bax[NRB], ban[NRB];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb) { bax[rb] = 64; ban[rb] = 64; }
#pragma unroll
      for (int i = 15; i >= 0; --i)
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
          bax[rb] = acc[rb][i] == pmx[rb / NB] ? i : bax[rb];
          if (!ROUTED) ban[rb] = acc[rb][i] == pmn[rb / NB] ? i : ban[rb];
        }
#pragma unroll
      for (int pt = 0; pt < TP; ++pt) {
#pragma unroll
        for (int q_ = 0; q_ < NB; ++q_) {
          const int rb = pt * NB + q_;
          const int rbase = q_ * 32 + 4 * lh;           // point-row of register 0
          const int rx = rbase + (bax[rb] & 3) + 8 * (bax[rb] >> 2);
          pax[pt] = q_ == 0 ? rx : min(pax[pt], rx);
          if (!ROUTED) {
            const int rn = rbase + (ban[rb] & 3) + 8 * (ban[rb] >> 2);
            pan[pt] = q_ == 0 ? rn : min(pan[pt], rn);
          }
        }
        pax[pt] = both_halves_min_i(pax[pt]);
        if (!ROUTED) pan[pt] = both_halves_min_i(pan[pt]);
      }
    }
#pragma unroll
    for (int pt = 0; pt < TP; ++pt) {
      if (lh == 0 && n0 + pt < N) {
        const long o = ((long)b0 * N + n0 + pt) * COUT + col;
        a.ymax[o] = ROUTED ? fmaf(pmx[pt], sgn, qc[pt]) : pmx[pt] + qc[pt];
        if (!ROUTED) a.ymin[o] = pmn[pt] + qc[pt];
        if (WITH_ARG) {
          a.amax[o] = (unsigned char)pax[pt];
          if (!ROUTED) a.amin[o] = (unsigned char)pan[pt];
        }
      }
    }
    // the DMA of tile t+1 must have landed (and every wave must be done reading this buffer) before the next tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    b0 = b1; m0 = m1; b1 = b2; m1 = m2;
    step(b2, m2, t + 2);
  };

  for (int t = t_begin; t < t_end; t += 2) {
    tile(t, std::integral_constant<int, 0>{});
    if (t + 1 < t_end) tile(t + 1, std::integral_constant<int, 1>{});
  }
  flush_stats(cur_b);
}

template <int KS, int CW, int NB, bool F16>
static int launch_q_nb(EcqArgs &a, bool with_arg, hipStream_t st) {
  constexpr int A_BYTES = 128 * KS * 32;
  constexpr int TP = NB == 3 ? 1 : 4 / NB;
  const int lds_bytes = 2 * A_BYTES;
  a.TP = TP;
  a.tiles_per_cloud = (a.N + TP - 1) / TP;
  a.total_tiles = a.B * a.tiles_per_cloud;
  // persistent grid: two waves per SIMD (8 waves per CU), bounded by the LDS
  int wg_per_cu = 8 / CW;
  const int by_lds = (160 * 1024) / lds_bytes;
  if (wg_per_cu > by_lds) wg_per_cu = by_lds;
  const int resident = 256 * wg_per_cu;
  const int grid = a.total_tiles < resident ? a.total_tiles : resident;
  using kern_t = void (*)(EcqArgs);
  const bool kexact = a.k == NB * 32;
  const bool routed = a.gamma_route != nullptr;
  kern_t kern;
#define ECQ_PICK(KE) \
  (routed ? (with_arg ? (kern_t)edgeconv_fwd_q_kernel<KS, CW, NB, KE, true, true, F16> : (kern_t)edgeconv_fwd_q_kernel<KS, CW, NB, KE, false, true, F16>) \
          : (with_arg ? (kern_t)edgeconv_fwd_q_kernel<KS, CW, NB, KE, true, false, F16> : (kern_t)edgeconv_fwd_q_kernel<KS, CW, NB, KE, false, false, F16>))
  kern = kexact ? ECQ_PICK(true) : ECQ_PICK(false);
#undef ECQ_PICK
  GCN_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  kern<<<grid, 64 * CW, lds_bytes, st>>>(a);
  return check_launch("edgeconv_fwd_q_kernel");
}

template <int KS, int CW, bool F16>
static int launch_q(EcqArgs &a, bool with_arg, hipStream_t st) {
  switch ((a.k + 31) / 32) {
    case 1: return launch_q_nb<KS, CW, 1, F16>(a, with_arg, st);
    case 2: return launch_q_nb<KS, CW, 2, F16>(a, with_arg, st);
    case 3: return launch_q_nb<KS, CW, 3, F16>(a, with_arg, st);
    case 4: return launch_q_nb<KS, CW, 4, F16>(a, with_arg, st);
  }
  set_error("edgeconv_fwd_q: k=%d > 128", a.k);
  return GCN_EINVAL;
}

// centre term launcher shared by the two operand types
template <bool F16>
static int launch_center(const void *x16, const void *wp16, long rows, int C, int Cout, float *q, hipStream_t st) {
  int Cp = 16;
  while (Cp < C) Cp <<= 1;
  const int ks = Cp / 16;
  const int grid = cdiv(rows, 128);
#define ECC_CASE(KSV, CWV)                                                                                         \
  if (ks == KSV && Cout == CWV * 32) {                                                                             \
    edgeconv_center_kernel<KSV, CWV, F16><<<grid, 256, 0, st>>>((const unsigned short *)x16,                       \
                                                                (const unsigned short *)wp16, rows, q);            \
    return check_launch("edgeconv_center_kernel");                                                                 \
  }
  if constexpr (!F16) { ECC_CASE(1, 2) ECC_CASE(2, 2) ECC_CASE(1, 4) ECC_CASE(2, 4) }
  ECC_CASE(4, 2) ECC_CASE(8, 2) ECC_CASE(4, 4) ECC_CASE(8, 4) ECC_CASE(16, 4)
#undef ECC_CASE
  set_error("gcn_edgeconv_center: unsupported configuration (C=%d, Cout=%d%s)", C, Cout, F16 ? ", half operands need C >= 33" : "");
  return GCN_EINVAL;
}

template <bool F16>
static int launch_fwd_q_t(EcqArgs &a, int Cp, bool with_arg, hipStream_t st) {
  const int ks = Cp / 16;
#define ECQ_CASE(KSV, CWV) \
  if (ks == KSV && a.Cout == CWV * 32) return launch_q<KSV, CWV, F16>(a, with_arg, st);
  if constexpr (!F16) { ECQ_CASE(1, 2) ECQ_CASE(2, 2) ECQ_CASE(1, 4) ECQ_CASE(2, 4) }
  ECQ_CASE(4, 2) ECQ_CASE(8, 2) ECQ_CASE(4, 4) ECQ_CASE(8, 4) ECQ_CASE(16, 4)
#undef ECQ_CASE
  set_error("gcn_edgeconv_fwd(%s): unsupported configuration Cp=%d Cout=%d", F16 ? "f16" : "bf16", Cp, a.Cout);
  return GCN_EINVAL;
}

}  // namespace gcn
