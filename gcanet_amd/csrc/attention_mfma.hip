// attention_mfma.hip -- flash-style scaled-dot-product attention on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, f32 accumulation) for the long-sequence attention stacks of the reference:
// models/transformer.py:36-75 (self-attention over all n points, materialises (b,h,n,n) scores) and
// models/query_decoder.py:5-72 (nn.MultiheadAttention, 100 queries x N points per cloud).
// BASELINE config 5 (N = 16384, 8 heads) cannot afford the score matrix: nothing of size Lq x Lk ever
// reaches HBM here, forward or backward.  The exact-f32 kernel of attention.hip stays the parity path.
//
// Common shape of the three kernels (forward, dQ, dK/dV): a workgroup of 4 waves owns 128 rows of ONE
// side (32 per wave, operand fragments in registers for the whole kernel) and streams 64-row tiles of
// the OTHER side through LDS by LDS-DMA (global_load_lds_dwordx4, double buffered, one barrier per
// tile).  Scores are always computed TRANSPOSED to what the row softmax would suggest -- the streamed
// side on the MFMA row index, the owned side on the lane -- so that
//   * every per-row quantity of the owned side (running max, sum, LSE, delta) is one value per lane,
//   * the accumulator tile is, register for register, the B operand of the next product
//     (P^T for O^T = V^T.P^T;  dS^T for dQ^T = K^T.dS^T;  P, dS for dV^T = dO^T.P, dK^T = Q^T.dS).
// The contraction index of that next product then runs over accumulator registers in C-layout order
// (row = (r&3) + 8(r>>2) + 4(lane>>5)), so the "transposed" LDS images (V^T, K^T, Q^T, dO^T) are stored
// with the rows of every 16-group permuted to that order (attn_pack_kernel) and one ds_read_b128
// yields the matching A operand.  LDS images are lane-linear (DMA), the XOR swizzle that makes the
// b128 reads conflict-free sits on the DMA source address.
#include "common.h"

namespace gcn {

typedef __attribute__((ext_vector_type(8))) short a_bf16x8;
typedef __attribute__((ext_vector_type(16))) float a_f32x16;
typedef __attribute__((ext_vector_type(2))) float a_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 a_bf16x2;

__device__ __forceinline__ unsigned int pack_bf16x2(float a, float b) {
  a_f32x2 v = {a, b};
  a_bf16x2 h = __builtin_convertvector(v, a_bf16x2);   // v_cvt_pk_bf16_f32 (RNE)
  return __builtin_bit_cast(unsigned int, h);
}
__device__ __forceinline__ unsigned short to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// 16-bit operand type of the kernels: bf16 (F16 = false) or IEEE half (F16 = true, BASELINE config 5 "fp16+MFMA":
// 11 significand bits instead of 8, at the price of range -- values below 6e-8 flush, which the probabilities and
// their gradients tolerate).  The images, fragments and LDS layouts are identical; only the conversions and the MFMA
// opcode (v_mfma_f32_32x32x16_f16) differ.
typedef __attribute__((ext_vector_type(8))) _Float16 a_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 a_f16x2;
template <bool F16>
__device__ __forceinline__ unsigned int pack_x2(float a, float b) {
  if (F16) {
    a_f32x2 v = {a, b};
    a_f16x2 h = __builtin_convertvector(v, a_f16x2);   // round to nearest even
    return __builtin_bit_cast(unsigned int, h);
  }
  return pack_bf16x2(a, b);
}
template <bool F16>
__device__ __forceinline__ unsigned short to_16(float f) {
  if (F16) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(unsigned short, h);
  }
  return to_bf16(f);
}
template <bool F16>
__device__ __forceinline__ a_f32x16 mfma_16(a_bf16x8 a, a_bf16x8 b, a_f32x16 c) {
  if (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(a_f16x8, a), __builtin_bit_cast(a_f16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// position (within a group of 16) -> row of the group, in accumulator order: lane half h = p>>3 holds
// rows 4h + (i&3) + 8(i>>2), i = p&7
__device__ __host__ __forceinline__ int attn_row_of_pos(int p) {
  const int h = p >> 3, i = p & 7;
  return 4 * h + (i & 3) + 8 * (i >> 2);
}

// src (BH, L, D) f32 -> row-major image dst_rm (BH, Lp, D) bf16 (zero rows past L, values * mul)
// and/or permuted transposed image dst_t (BH, D, Lp) bf16.  One block = 64 rows of one bh.
template <int D, bool F16>
__global__ __launch_bounds__(256) void attn_pack_kernel(const float *__restrict__ src, int L, int Lp, float mul,
                                                        unsigned short *__restrict__ dst_rm,
                                                        unsigned short *__restrict__ dst_t) {
  __shared__ float tile[64][D + 1];
  const int bh = blockIdx.y, l0 = blockIdx.x * 64;
  const float *s = src + (long)bh * L * D;
  for (int e = threadIdx.x; e < 64 * D; e += 256) {
    const int r = e / D, d = e % D;
    const float v = (l0 + r < L) ? s[(long)(l0 + r) * D + d] * mul : 0.f;
    tile[r][d] = v;
    if (dst_rm) dst_rm[((long)bh * Lp + l0 + r) * D + d] = to_16<F16>(v);
  }
  if (!dst_t) return;
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * D; e += 256) {
    const int d = e / 64, p = e % 64;
    const int r = (p & ~15) + attn_row_of_pos(p & 15);
    dst_t[((long)bh * D + d) * Lp + l0 + p] = to_16<F16>(tile[r][d]);
  }
}

// ---- LDS-DMA of one 64-row tile -----------------------------------------------------------------
// Row-major image tile: 64 rows x D bf16 (RB = 2D bytes per row), 16-B chunks XOR-swizzled so that 16
// consecutive rows reading the same logical chunk land on 16 different 16-B slots of the 256-B bank row.
template <int D>
__device__ __forceinline__ int rm_swz(int row) {
  constexpr int RB = 2 * D, RPB = 256 / RB, CPR = RB / 16;
  return (row / RPB) & (CPR - 1);
}
template <int D>
__device__ __forceinline__ void dma_rm_tile(const unsigned short *__restrict__ img_row0, unsigned char *lds, int wave,
                                            int lane, int nwaves = 4) {
  constexpr int RB = 2 * D, CPR = RB / 16, RPP = 1024 / RB, PIECES = 64 * RB / 1024;
  for (int p = wave; p < PIECES; p += nwaves) {
    const int row = p * RPP + lane / CPR, pos = lane % CPR;
    const int chunk = pos ^ rm_swz<D>(row);
    const unsigned char *src = reinterpret_cast<const unsigned char *>(img_row0) + (long)row * RB + chunk * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(lds + p * 1024), 16, 0, 0);
  }
}
// Transposed image tile: D rows (channels) x 64 positions bf16 = 128-B rows, row stride Lp elements in global.
template <int D>
__device__ __forceinline__ void dma_t_tile(const unsigned short *__restrict__ img_col0, long Lp, unsigned char *lds,
                                           int wave, int lane, int nwaves = 4) {
  constexpr int PIECES = D / 8;
  for (int p = wave; p < PIECES; p += nwaves) {
    const int row = p * 8 + lane / 8, pos = lane % 8;
    const int chunk = pos ^ ((row >> 1) & 7);
    const unsigned char *src = reinterpret_cast<const unsigned char *>(img_col0 + (long)row * Lp) + chunk * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)(lds + p * 1024), 16, 0, 0);
  }
}
// A operand from a row-major tile: row (0..63), logical 16-B chunk c (= k-step*2 + lane half)
template <int D>
__device__ __forceinline__ a_bf16x8 lds_rm_frag(const unsigned char *lds, int row, int c) {
  return *reinterpret_cast<const a_bf16x8 *>(lds + row * (2 * D) + ((c ^ rm_swz<D>(row)) << 4));
}
// A operand from a transposed tile: row = channel, logical chunk c (8 positions)
__device__ __forceinline__ a_bf16x8 lds_t_frag(const unsigned char *lds, int row, int c) {
  return *reinterpret_cast<const a_bf16x8 *>(lds + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}

// ---- forward --------------------------------------------------------------------------------------
// qb (BH,Lqp,D) bf16 pre-scaled by scale*log2(e); kb (BH,Lkp,D); vt (BH,D,Lkp) permuted; mask bytes
// (1 = masked out) (Lq,Lk) [+ bh stride]; out (BH,Lq,D) f32; lse (BH,Lq) f32 natural-log units.
template <int D, bool MASK, bool F16>
__global__ __launch_bounds__(512, 2) void attn_fwd_mfma_kernel(const unsigned short *__restrict__ qb,
                                                               const unsigned short *__restrict__ kb,
                                                               const unsigned short *__restrict__ vt,
                                                               const unsigned char *__restrict__ mask, long mask_bh_stride,
                                                               int Lq, int Lk, int Lqp, int Lkp,
                                                               float *__restrict__ out, float *__restrict__ lse,
                                                               float *__restrict__ part) {
  // gridDim.z > 1 (few query rows, many keys -- the query decoder's cross attention): workgroup z streams its share of
  // the key tiles and leaves (unnormalised O, running maximum, running sum) in `part`; attn_fwd_combine_kernel joins them.
  constexpr int KS = D / 16;            // k-steps of S^T = K.Q^T
  constexpr int DB = D / 32;            // 32-row blocks of O^T
  constexpr int KT_BYTES = 64 * D * 2;  // K tile; the V^T tile is the same size
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (K tile | V^T tile)

  const int lane = lane_id(), wave = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  int tile_, bh;
  xcd_major_tile_cloud(tile_, bh);
  const int nwaves = (int)(blockDim.x >> 6);          // 4 or 8 waves (32 query rows each) share the streamed K / V tiles
  const int qrow = tile_ * (32 * nwaves) + wave * 32 + lr;     // < Lqp by construction
  const unsigned short *kimg = kb + (long)bh * Lkp * D;
  const unsigned short *vimg = vt + (long)bh * D * Lkp;

  a_bf16x8 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s)
    qf[s] = *reinterpret_cast<const a_bf16x8 *>(qb + ((long)bh * Lqp + qrow) * D + 16 * s + 8 * lh);

  a_f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -__builtin_inff(), l_run = 0.f;

  const int tiles_all = Lkp / 64, tps = (tiles_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int tbeg = (int)blockIdx.z * tps, ntiles = min(tiles_all, tbeg + tps);     // this workgroup's key tiles [tbeg, ntiles)
  auto issue = [&](int t, int buf) {
    unsigned char *base = smem + buf * 2 * KT_BYTES;
    dma_rm_tile<D>(kimg + (long)t * 64 * D, base, wave, lane, nwaves);
    dma_t_tile<D>(vimg + (long)t * 64, Lkp, base + KT_BYTES, wave, lane, nwaves);
  };
  if (tbeg < ntiles) issue(tbeg, tbeg & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = tbeg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    const unsigned char *kt = smem + buf * 2 * KT_BYTES;
    const unsigned char *vtile = kt + KT_BYTES;
    const int key0 = t * 64;

    a_f32x16 sc[2];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kb2][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s)
        sc[kb2] = mfma_16<F16>(lds_rm_frag<D>(kt, 32 * kb2 + lr, 2 * s + lh), qf[s], sc[kb2]);
    }
    // keys past Lk (zero padding of the last tile) and user mask -> -inf
    if (MASK || key0 + 64 > Lk) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + 32 * kb2 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          bool dead = key >= Lk;
          if (MASK && !dead && qrow < Lq) dead = mask[(long)bh * mask_bh_stride + (long)qrow * Lk + key] != 0;
          if (dead) sc[kb2][r] = -__builtin_inff();
        }
    }
    float mx = sc[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -__builtin_inff() ? 0.f : m_new;     // fully masked so far: p = 0 everywhere
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);        // m_run = -inf -> 0
    m_run = m_new;
    float ps = 0.f;
    unsigned int pf[2][2][4];
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(sc[kb2][r] - m_use);
        const float p1 = __builtin_amdgcn_exp2f(sc[kb2][r + 1] - m_use);
        ps += p0 + p1;
        pf[kb2][r >> 3][(r & 7) >> 1] = pack_x2<F16>(p0, p1);
      }
    l_run = l_run * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {                 // wave-uniform skip once the maxima settle
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          a_bf16x8 pb;
          unsigned int *pw = reinterpret_cast<unsigned int *>(&pb);
          pw[0] = pf[kb2][tt][0]; pw[1] = pf[kb2][tt][1]; pw[2] = pf[kb2][tt][2]; pw[3] = pf[kb2][tt][3];
          o[d] = mfma_16<F16>(lds_t_frag(vtile, 32 * d + lr, 4 * kb2 + 2 * tt + lh), pb, o[d]);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32);
  if (gridDim.z > 1) {
    if (qrow < Lq) {
      const long row = ((long)blockIdx.z * gridDim.y + bh) * Lq + qrow;
      float *prow = part + row * (D + 4);                      // D values, maximum, sum, 2 pad floats: 16-byte rows
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          prow[32 * d + 8 * g + 4 * lh + 0] = o[d][4 * g];
          prow[32 * d + 8 * g + 4 * lh + 1] = o[d][4 * g + 1];
          prow[32 * d + 8 * g + 4 * lh + 2] = o[d][4 * g + 2];
          prow[32 * d + 8 * g + 4 * lh + 3] = o[d][4 * g + 3];
        }
      if (lh == 0) { prow[D] = m_run; prow[D + 1] = l_tot; }
    }
    return;
  }
  if (qrow < Lq) {
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    float *orow = out + ((long)bh * Lq + qrow) * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v4 = make_float4(o[d][4 * g] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
        *reinterpret_cast<float4 *>(orow + 32 * d + 8 * g + 4 * lh) = v4;
      }
    if (lse && lh == 0) lse[(long)bh * Lq + qrow] = (m_run + __builtin_amdgcn_logf(l_tot)) * 0.6931471805599453f;
  }
}

// joins the key-split partials: m = max_z m_z, l = sum_z l_z 2^(m_z - m), O = sum_z O_z 2^(m_z - m) / l; one thread per
// (row, 4 channels); a split whose keys were all masked carries m_z = -inf and weight 0
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(const float *__restrict__ part, int splits, long rows,
                                                               float *__restrict__ out, float *__restrict__ lse) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = e / (D / 4);
  const int c4 = (int)(e % (D / 4));
  if (row >= rows) return;
  float m = -__builtin_inff();
  for (int z = 0; z < splits; ++z) m = fmaxf(m, part[((long)z * rows + row) * (D + 4) + D]);
  const float mu = m == -__builtin_inff() ? 0.f : m;
  float l = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < splits; ++z) {
    const float *p = part + ((long)z * rows + row) * (D + 4);
    const float w = __builtin_amdgcn_exp2f(p[D] - mu);         // -inf -> 0
    l = fmaf(p[D + 1], w, l);
    const float4 v = *reinterpret_cast<const float4 *>(p + 4 * c4);
    acc.x = fmaf(v.x, w, acc.x); acc.y = fmaf(v.y, w, acc.y); acc.z = fmaf(v.z, w, acc.z); acc.w = fmaf(v.w, w, acc.w);
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  *reinterpret_cast<float4 *>(out + row * D + 4 * c4) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (lse && c4 == 0) lse[row] = (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f;
}

// sum of the key-split partial dQ tiles
__global__ __launch_bounds__(256) void attn_sum_splits_kernel(const float4 *__restrict__ part, int splits, long n4, float4 *__restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n4) return;
  float4 a = part[e];
  for (int z = 1; z < splits; ++z) {
    const float4 v = part[(long)z * n4 + e];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  out[e] = a;
}

// Key splits of the forward and dQ kernels: only when the query side cannot fill the chip (fewer than 128 workgroups)
// and there are enough key tiles to share out; at most 16.
static inline int attn_key_splits(long row_wgs, int key_tiles) {
  if (row_wgs >= 128 || key_tiles < 32) return 1;
  long s = (256 + row_wgs - 1) / row_wgs;
  if (s > key_tiles / 8) s = key_tiles / 8;
  if (s > 16) s = 16;
  return s < 2 ? 1 : (int)s;
}

// ---- backward -------------------------------------------------------------------------------------
// delta[q] = sum_d dO[q,d] * O[q,d]  and  lse2[q] = lse[q] * log2(e)  (padded rows and fully masked rows:
// delta = 0, lse2 = +inf so that p = exp2(s - lse2) = 0).  8 lanes per row.
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const float *__restrict__ dout, const float *__restrict__ out,
                                                         const float *__restrict__ lse, int Lq, int Lqp,
                                                         float *__restrict__ delta, float *__restrict__ lse2) {
  const int bh = blockIdx.y;
  const int row = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
  float s = 0.f;
  if (row < Lq) {
    const float *a = dout + ((long)bh * Lq + row) * D, *b = out + ((long)bh * Lq + row) * D;
#pragma unroll
    for (int c = sub * 4; c < D; c += 32) {
      const float4 x = *reinterpret_cast<const float4 *>(a + c), y = *reinterpret_cast<const float4 *>(b + c);
      s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
  }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  if (sub == 0 && row < Lqp) {
    float l2 = __builtin_inff();
    if (row < Lq) {
      const float l = lse[(long)bh * Lq + row];
      if (l > -__builtin_inff()) l2 = l * 1.4426950408889634f;
    }
    delta[(long)bh * Lqp + row] = s;
    lse2[(long)bh * Lqp + row] = l2;
  }
}

__device__ __forceinline__ a_bf16x8 make_frag(const unsigned int *w) {
  a_bf16x8 f;
  unsigned int *pw = reinterpret_cast<unsigned int *>(&f);
  pw[0] = w[0]; pw[1] = w[1]; pw[2] = w[2]; pw[3] = w[3];
  return f;
}

// dQ: the workgroup owns 128 queries (lane = query) and streams key tiles: K, V (row-major) and K^T.
//   S^T = K.Qs^T, dP^T = V.dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T.dS^T;  dQ = scale * dQ^T.
template <int D, bool MASK, bool F16>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const unsigned short *__restrict__ qs,
                                                             const unsigned short *__restrict__ kr,
                                                             const unsigned short *__restrict__ vr,
                                                             const unsigned short *__restrict__ ktr,
                                                             const unsigned short *__restrict__ dor,
                                                             const float *__restrict__ lse2, const float *__restrict__ delta,
                                                             const unsigned char *__restrict__ mask, long mask_bh_stride,
                                                             int Lq, int Lk, int Lqp, int Lkp, float scale,
                                                             float *__restrict__ dq) {
  // gridDim.z > 1: workgroup z takes its share of the key tiles and writes a partial dQ (dq then points at
  // [z][bh][Lq][D] partial tiles, summed by attn_sum_splits_kernel)
  constexpr int KS = D / 16, DB = D / 32, TB = 64 * D * 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (K | V | K^T)
  const int lane = lane_id(), wave = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  int tile_, bh;
  xcd_major_tile_cloud(tile_, bh);
  const int qrow = tile_ * 128 + wave * 32 + lr;
  const unsigned short *kimg = kr + (long)bh * Lkp * D, *vimg = vr + (long)bh * Lkp * D;
  const unsigned short *ktimg = ktr + (long)bh * D * Lkp;

  a_bf16x8 qf[KS], dof[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    qf[s] = *reinterpret_cast<const a_bf16x8 *>(qs + ((long)bh * Lqp + qrow) * D + 16 * s + 8 * lh);
    dof[s] = *reinterpret_cast<const a_bf16x8 *>(dor + ((long)bh * Lqp + qrow) * D + 16 * s + 8 * lh);
  }
  const float my_lse2 = lse2[(long)bh * Lqp + qrow], my_delta = delta[(long)bh * Lqp + qrow];
  a_f32x16 acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int tiles_all = Lkp / 64, tps = (tiles_all + (int)gridDim.z - 1) / (int)gridDim.z;
  const int tbeg = (int)blockIdx.z * tps, ntiles = min(tiles_all, tbeg + tps);
  auto issue = [&](int t, int buf) {
    unsigned char *base = smem + buf * 3 * TB;
    dma_rm_tile<D>(kimg + (long)t * 64 * D, base, wave, lane);
    dma_rm_tile<D>(vimg + (long)t * 64 * D, base + TB, wave, lane);
    dma_t_tile<D>(ktimg + (long)t * 64, Lkp, base + 2 * TB, wave, lane);
  };
  if (tbeg < ntiles) issue(tbeg, tbeg & 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = tbeg; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    const unsigned char *kt = smem + buf * 3 * TB, *vtile = kt + TB, *ktt = kt + 2 * TB;
    const int key0 = t * 64;
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      a_f32x16 sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = -my_lse2; dp[r] = -my_delta; }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sc = mfma_16<F16>(lds_rm_frag<D>(kt, 32 * kb2 + lr, 2 * s + lh), qf[s], sc);
        dp = mfma_16<F16>(lds_rm_frag<D>(vtile, 32 * kb2 + lr, 2 * s + lh), dof[s], dp);
      }
      unsigned int dsw[2][4];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int key = key0 + 32 * kb2 + ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * lh;
          bool dead = key >= Lk;
          if (MASK && !dead && qrow < Lq) dead = mask[(long)bh * mask_bh_stride + (long)qrow * Lk + key] != 0;
          const float p = dead ? 0.f : __builtin_amdgcn_exp2f(sc[r + u]);
          ds2[u] = p * dp[r + u];
        }
        dsw[r >> 3][(r & 7) >> 1] = pack_x2<F16>(ds2[0], ds2[1]);
      }
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
          acc[d] = mfma_16<F16>(lds_t_frag(ktt, 32 * d + lr, 4 * kb2 + 2 * tt + lh),
                                                           make_frag(dsw[tt]), acc[d]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (qrow < Lq) {
    float *orow = dq + (((long)blockIdx.z * gridDim.y + bh) * Lq + qrow) * D;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4 *>(orow + 32 * d + 8 * g + 4 * lh) =
            make_float4(acc[d][4 * g] * scale, acc[d][4 * g + 1] * scale, acc[d][4 * g + 2] * scale, acc[d][4 * g + 3] * scale);
  }
}

// dK, dV: the workgroup owns 128 keys (lane = key) and streams query tiles: Qs, dO (row-major), Qs^T, dO^T
// and the tile's lse2 / delta.   S = Qs.K^T - lse2, dP = dO.V^T - delta (row constants enter as the initial
// accumulator), P = exp2(S), dS = P o dP;  dV^T += dO^T.P,  dK^T += Qs^T.dS;  dK = ln2 * dK^T (Qs carries log2 e).
template <int D, bool MASK, bool F16>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const unsigned short *__restrict__ qs,
                                                              const unsigned short *__restrict__ qst,
                                                              const unsigned short *__restrict__ kr,
                                                              const unsigned short *__restrict__ vr,
                                                              const unsigned short *__restrict__ dor,
                                                              const unsigned short *__restrict__ dot,
                                                              const float *__restrict__ lse2, const float *__restrict__ delta,
                                                              const unsigned char *__restrict__ mask, long mask_bh_stride,
                                                              int Lq, int Lk, int Lqp, int Lkp,
                                                              float *__restrict__ dk, float *__restrict__ dv) {
  constexpr int KS = D / 16, DB = D / 32, TB = 64 * D * 2, BUF = 4 * TB + 1024;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // 2 x (Qs | dO | Qs^T | dO^T | lse2,delta)
  const int lane = lane_id(), wave = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  int tile_, bh;
  xcd_major_tile_cloud(tile_, bh);
  const int krow = tile_ * 128 + wave * 32 + lr;      // < Lkp
  const unsigned short *qimg = qs + (long)bh * Lqp * D, *doimg = dor + (long)bh * Lqp * D;
  const unsigned short *qtimg = qst + (long)bh * D * Lqp, *dotimg = dot + (long)bh * D * Lqp;
  const float *l2 = lse2 + (long)bh * Lqp, *dl = delta + (long)bh * Lqp;

  a_bf16x8 kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    kf[s] = *reinterpret_cast<const a_bf16x8 *>(kr + ((long)bh * Lkp + krow) * D + 16 * s + 8 * lh);
    vf[s] = *reinterpret_cast<const a_bf16x8 *>(vr + ((long)bh * Lkp + krow) * D + 16 * s + 8 * lh);
  }
  a_f32x16 dkt[DB], dvt[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[d][r] = 0.f; dvt[d][r] = 0.f; }

  const int ntiles = Lqp / 64;
  auto issue = [&](int t, int buf) {
    unsigned char *base = smem + buf * BUF;
    dma_rm_tile<D>(qimg + (long)t * 64 * D, base, wave, lane);
    dma_rm_tile<D>(doimg + (long)t * 64 * D, base + TB, wave, lane);
    dma_t_tile<D>(qtimg + (long)t * 64, Lqp, base + 2 * TB, wave, lane);
    dma_t_tile<D>(dotimg + (long)t * 64, Lqp, base + 3 * TB, wave, lane);
    if (wave == 0)       // 64 floats each: one dword per lane
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(l2 + t * 64 + lane),
                                       (__attribute__((address_space(3))) void *)(base + 4 * TB), 4, 0, 0);
    else if (wave == 1)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(dl + t * 64 + lane),
                                       (__attribute__((address_space(3))) void *)(base + 4 * TB + 256), 4, 0, 0);
  };
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    const unsigned char *qt = smem + buf * BUF, *dot_rm = qt + TB, *qtt = qt + 2 * TB, *dott = qt + 3 * TB;
    const float *lt = reinterpret_cast<const float *>(qt + 4 * TB), *dt = lt + 64;
    const int q0 = t * 64;
#pragma unroll
    for (int qb2 = 0; qb2 < 2; ++qb2) {
      a_f32x16 sc, dp;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 a = *reinterpret_cast<const float4 *>(lt + 32 * qb2 + 8 * g + 4 * lh);
        const float4 b = *reinterpret_cast<const float4 *>(dt + 32 * qb2 + 8 * g + 4 * lh);
        sc[4 * g] = -a.x; sc[4 * g + 1] = -a.y; sc[4 * g + 2] = -a.z; sc[4 * g + 3] = -a.w;
        dp[4 * g] = -b.x; dp[4 * g + 1] = -b.y; dp[4 * g + 2] = -b.z; dp[4 * g + 3] = -b.w;
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sc = mfma_16<F16>(lds_rm_frag<D>(qt, 32 * qb2 + lr, 2 * s + lh), kf[s], sc);
        dp = mfma_16<F16>(lds_rm_frag<D>(dot_rm, 32 * qb2 + lr, 2 * s + lh), vf[s], dp);
      }
      unsigned int pw[2][4], dsw[2][4];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float p2[2], ds2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float p = __builtin_amdgcn_exp2f(sc[r + u]);          // padded / fully masked rows: lse2 = +inf -> 0
          if (MASK) {
            const int qq = q0 + 32 * qb2 + ((r + u) & 3) + 8 * ((r + u) >> 2) + 4 * lh;
            if (qq < Lq && krow < Lk && mask[(long)bh * mask_bh_stride + (long)qq * Lk + krow] != 0) p = 0.f;
          }
          p2[u] = p;
          ds2[u] = p * dp[r + u];
        }
        pw[r >> 3][(r & 7) >> 1] = pack_x2<F16>(p2[0], p2[1]);
        dsw[r >> 3][(r & 7) >> 1] = pack_x2<F16>(ds2[0], ds2[1]);
      }
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          dvt[d] = mfma_16<F16>(lds_t_frag(dott, 32 * d + lr, 4 * qb2 + 2 * tt + lh),
                                                           make_frag(pw[tt]), dvt[d]);
          dkt[d] = mfma_16<F16>(lds_t_frag(qtt, 32 * d + lr, 4 * qb2 + 2 * tt + lh),
                                                           make_frag(dsw[tt]), dkt[d]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (krow < Lk) {
    float *krow_o = dk + ((long)bh * Lk + krow) * D, *vrow_o = dv + ((long)bh * Lk + krow) * D;
    const float ln2 = 0.6931471805599453f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4 *>(krow_o + 32 * d + 8 * g + 4 * lh) =
            make_float4(dkt[d][4 * g] * ln2, dkt[d][4 * g + 1] * ln2, dkt[d][4 * g + 2] * ln2, dkt[d][4 * g + 3] * ln2);
        *reinterpret_cast<float4 *>(vrow_o + 32 * d + 8 * g + 4 * lh) =
            make_float4(dvt[d][4 * g], dvt[d][4 * g + 1], dvt[d][4 * g + 2], dvt[d][4 * g + 3]);
      }
  }
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
// operand images of gcn_attention_ws_bytes (the key-split partials start behind them, 256-byte aligned)
static inline long attn_ws_base(int BH, int Lq, int Lk, int D) {
  const long Lqp = round_up(Lq, 128), Lkp = round_up(Lk, 128);
  return ((2L * BH * D * (4 * Lqp + 4 * Lkp) + 8L * BH * Lqp + 1024) + 255) & ~255L;
}

template <int D, bool F16>
static int run_bwd(const float *q, const float *k, const float *v, const float *out, const float *dout, const float *lse,
                   const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, float scale, float *dq, float *dk,
                   float *dv, unsigned char *wsb, hipStream_t st) {
  const int Lqp = round_up(Lq, 128), Lkp = round_up(Lk, 128);
  const size_t nq = (size_t)BH * Lqp * D, nk = (size_t)BH * Lkp * D;
  unsigned short *qs = (unsigned short *)wsb, *qst = qs + nq, *dor = qst + nq, *dot = dor + nq;
  unsigned short *kr = dot + nq, *ktr = kr + nk, *vr = ktr + nk;
  float *delta = (float *)(vr + nk), *lse2 = delta + (size_t)BH * Lqp;
  const int splits = attn_key_splits((long)BH * (Lqp / 128), Lkp / 64);
  float *dq_out = dq;
  if (splits > 1) dq_out = reinterpret_cast<float *>(wsb + attn_ws_base(BH, Lq, Lk, D));    // partial dQ tiles [split][BH][Lq][D]
  const dim3 gq(Lqp / 128, BH, splits);
  attn_pack_kernel<D, F16><<<dim3(Lqp / 64, BH), 256, 0, st>>>(q, Lq, Lqp, scale * 1.4426950408889634f, qs, qst);
  attn_pack_kernel<D, F16><<<dim3(Lqp / 64, BH), 256, 0, st>>>(dout, Lq, Lqp, 1.f, dor, dot);
  attn_pack_kernel<D, F16><<<dim3(Lkp / 64, BH), 256, 0, st>>>(k, Lk, Lkp, 1.f, kr, ktr);
  attn_pack_kernel<D, F16><<<dim3(Lkp / 64, BH), 256, 0, st>>>(v, Lk, Lkp, 1.f, vr, nullptr);
  attn_delta_kernel<D><<<dim3(Lqp / 32, BH), 256, 0, st>>>(dout, out, lse, Lq, Lqp, delta, lse2);
  const long ms = mask_per_bh ? (long)Lq * Lk : 0;
  const int lds_q = 2 * 3 * 64 * D * 2, lds_kv = 2 * (4 * 64 * D * 2 + 1024);
  if (mask) {
    GCN_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<D, true, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv));
    attn_bwd_dq_kernel<D, true, F16><<<gq, 256, lds_q, st>>>(qs, kr, vr, ktr, dor, lse2, delta, mask, ms, Lq, Lk, Lqp, Lkp, scale, dq_out);
    attn_bwd_dkv_kernel<D, true, F16><<<dim3(Lkp / 128, BH), 256, lds_kv, st>>>(qs, qst, kr, vr, dor, dot, lse2, delta, mask, ms, Lq, Lk, Lqp, Lkp, dk, dv);
  } else {
    GCN_HIP(hipFuncSetAttribute((const void *)attn_bwd_dkv_kernel<D, false, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv));
    attn_bwd_dq_kernel<D, false, F16><<<gq, 256, lds_q, st>>>(qs, kr, vr, ktr, dor, lse2, delta, nullptr, 0, Lq, Lk, Lqp, Lkp, scale, dq_out);
    attn_bwd_dkv_kernel<D, false, F16><<<dim3(Lkp / 128, BH), 256, lds_kv, st>>>(qs, qst, kr, vr, dor, dot, lse2, delta, nullptr, 0, Lq, Lk, Lqp, Lkp, dk, dv);
  }
  if (splits > 1) {
    const long n4 = (long)BH * Lq * D / 4;
    attn_sum_splits_kernel<<<(int)((n4 + 255) / 256), 256, 0, st>>>(reinterpret_cast<const float4 *>(dq_out), splits, n4,
                                                                     reinterpret_cast<float4 *>(dq));
  }
  return check_launch("attn_bwd kernels");
}


template <int D, bool F16>
static int run_fwd(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh, int BH, int Lq,
                   int Lk, float scale, float *out, float *lse, unsigned short *ws, hipStream_t st) {
  const int Lqp = round_up(Lq, 128), Lkp = round_up(Lk, 64);
  unsigned short *qb = ws, *kb = qb + (size_t)BH * Lqp * D, *vt = kb + (size_t)BH * Lkp * D;
  attn_pack_kernel<D, F16><<<dim3(Lqp / 64, BH), 256, 0, st>>>(q, Lq, Lqp, scale * 1.4426950408889634f, qb, nullptr);
  attn_pack_kernel<D, F16><<<dim3(Lkp / 64, BH), 256, 0, st>>>(k, Lk, Lkp, 1.f, kb, nullptr);
  attn_pack_kernel<D, F16><<<dim3(Lkp / 64, BH), 256, 0, st>>>(v, Lk, Lkp, 1.f, nullptr, vt);
  const int lds = 2 * 2 * 64 * D * 2;
  const long ms = mask_per_bh ? (long)Lq * Lk : 0;
  // eight waves (256 query rows) per streamed tile when the padded length allows it: half the LDS-DMA work per row
  const int threads = (Lqp % 256 == 0) ? 512 : 256;
  const int splits = attn_key_splits((long)BH * (Lqp / (threads / 2)), Lkp / 64);
  float *part = splits > 1 ? reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(ws) + attn_ws_base(BH, Lq, Lk, D)) : nullptr;
  const dim3 grid(Lqp / (threads / 2), BH, splits);
  if (mask)
    attn_fwd_mfma_kernel<D, true, F16><<<grid, threads, lds, st>>>(qb, kb, vt, mask, ms, Lq, Lk, Lqp, Lkp, out, lse, part);
  else
    attn_fwd_mfma_kernel<D, false, F16><<<grid, threads, lds, st>>>(qb, kb, vt, nullptr, 0, Lq, Lk, Lqp, Lkp, out, lse, part);
  if (splits > 1) {
    const long rows = (long)BH * Lq;
    attn_fwd_combine_kernel<D><<<(int)((rows * (D / 4) + 255) / 256), 256, 0, st>>>(part, splits, rows, out, lse);
  }
  return check_launch("attn_fwd_mfma_kernel");
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_attention_ws_bytes(int BH, int Lq, int Lk, int D) {
  if (BH < 0 || Lq < 1 || Lk < 1 || D < 1) return -1;
  const long Lqp = round_up(Lq, 128), Lkp = round_up(Lk, 128);
  // forward: Qs, K, V^T images; backward: Qs, Qs^T, dO, dO^T, K, K^T, V (all bf16), delta and lse2 (f32); then up to 16
  // key-split partials of (D + 4) floats per query row (forward) / D floats (dQ) when few query rows meet many keys
  (void)Lqp; (void)Lkp;
  return attn_ws_base(BH, Lq, Lk, D) + 16L * BH * Lq * (D + 4) * 4;
}

static int attention_fwd_16(bool f16, const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                            int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *ws, void *stream) {
  GCN_REQUIRE(q && k && v && out && ws, "gcn_attention_fwd_bf16: null pointer");
  GCN_REQUIRE(BH >= 0 && Lq >= 1 && Lk >= 1, "gcn_attention_fwd_bf16: bad shape");
  GCN_REQUIRE(D == 32 || D == 64, "gcn_attention_fwd_bf16: head dim %d unsupported (32, 64)", D);
  GCN_REQUIRE(((uintptr_t)ws & 15) == 0 && ((uintptr_t)out & 15) == 0, "gcn_attention_fwd_bf16: ws/out must be 16-B aligned");
  if (BH == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (f16) {
    if (D == 32) return run_fwd<32, true>(q, k, v, mask, mask_per_bh, BH, Lq, Lk, scale, out, lse, (unsigned short *)ws, st);
    return run_fwd<64, true>(q, k, v, mask, mask_per_bh, BH, Lq, Lk, scale, out, lse, (unsigned short *)ws, st);
  }
  if (D == 32) return run_fwd<32, false>(q, k, v, mask, mask_per_bh, BH, Lq, Lk, scale, out, lse, (unsigned short *)ws, st);
  return run_fwd<64, false>(q, k, v, mask, mask_per_bh, BH, Lq, Lk, scale, out, lse, (unsigned short *)ws, st);
}

GCN_EXPORT int gcn_attention_fwd_bf16(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                                      int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *ws,
                                      void *stream) {
  return attention_fwd_16(false, q, k, v, mask, mask_per_bh, BH, Lq, Lk, D, scale, out, lse, ws, stream);
}

GCN_EXPORT int gcn_attention_fwd_f16(const float *q, const float *k, const float *v, const uint8_t *mask, int mask_per_bh,
                                     int BH, int Lq, int Lk, int D, float scale, float *out, float *lse, void *ws,
                                     void *stream) {
  return attention_fwd_16(true, q, k, v, mask, mask_per_bh, BH, Lq, Lk, D, scale, out, lse, ws, stream);
}

static int attention_bwd_16(bool f16, const float *q, const float *k, const float *v, const float *out, const float *dout,
                            const float *lse, const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, int D,
                            float scale, float *dq, float *dk, float *dv, void *ws, void *stream) {
  GCN_REQUIRE(q && k && v && out && dout && lse && dq && dk && dv && ws, "gcn_attention_bwd_bf16: null pointer");
  GCN_REQUIRE(BH >= 0 && Lq >= 1 && Lk >= 1, "gcn_attention_bwd_bf16: bad shape");
  GCN_REQUIRE(D == 32 || D == 64, "gcn_attention_bwd_bf16: head dim %d unsupported (32, 64)", D);
  GCN_REQUIRE(((uintptr_t)ws & 15) == 0 && (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv | (uintptr_t)out | (uintptr_t)dout) & 15) == 0,
              "gcn_attention_bwd_bf16: buffers must be 16-B aligned");
  if (BH == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  if (f16) {
    if (D == 32) return run_bwd<32, true>(q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, scale, dq, dk, dv, (unsigned char *)ws, st);
    return run_bwd<64, true>(q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, scale, dq, dk, dv, (unsigned char *)ws, st);
  }
  if (D == 32) return run_bwd<32, false>(q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, scale, dq, dk, dv, (unsigned char *)ws, st);
  return run_bwd<64, false>(q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, scale, dq, dk, dv, (unsigned char *)ws, st);
}

GCN_EXPORT int gcn_attention_bwd_bf16(const float *q, const float *k, const float *v, const float *out, const float *dout,
                                      const float *lse, const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, int D,
                                      float scale, float *dq, float *dk, float *dv, void *ws, void *stream) {
  return attention_bwd_16(false, q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, D, scale, dq, dk, dv, ws, stream);
}

GCN_EXPORT int gcn_attention_bwd_f16(const float *q, const float *k, const float *v, const float *out, const float *dout,
                                     const float *lse, const uint8_t *mask, int mask_per_bh, int BH, int Lq, int Lk, int D,
                                     float scale, float *dq, float *dk, float *dv, void *ws, void *stream) {
  return attention_bwd_16(true, q, k, v, out, dout, lse, mask, mask_per_bh, BH, Lq, Lk, D, scale, dq, dk, dv, ws, stream);
}
