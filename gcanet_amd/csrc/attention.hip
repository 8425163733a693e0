// attention.hip -- fused scaled-dot-product attention forward (online softmax, no (n x m) score
// matrix in HBM) for the attention stacks of the reference: models/transformer.py:36-75 (`Attention`:
// einsum scores * dim**-0.5 -> softmax -> einsum with V, materialises (b,h,n,n)) and
// models/query_decoder.py:5-72 (nn.MultiheadAttention cross/self attention, per-cloud Python loop).
//
// This round's kernel is the exact-f32 version (parity path, within 1e-4 of the f32 reference):
//   one wave = 16 query rows, one 64-key tile per step, lane = key for the scores and lane = channel
//   for the P.V accumulation; K/V tiles staged in LDS and shared by the 4 waves of the workgroup.
// A bf16 MFMA flash kernel (S^T = K.Q^T so that row reductions stay in-register) is the planned
// replacement for the long-sequence Transformer case.
#include "common.h"

namespace gcn {

// q (BH, Lq, D), k (BH, Lk, D), v (BH, Lk, D) f32 contiguous; mask (Lq, Lk) bytes or null (1 = masked
// out, shared by all BH when mask_bh_stride == 0); out (BH, Lq, D); lse (BH, Lq) optional.
template <int D>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                            const float *__restrict__ v, const unsigned char *__restrict__ mask,
                                                            long mask_bh_stride, int Lq, int Lk, float scale,
                                                            float *__restrict__ out, float *__restrict__ lse) {
  constexpr int TK = 64;  // keys per tile
  constexpr int QW = 16;  // queries per wave
  __shared__ float ks[TK][D + 1];
  __shared__ float vs[TK][D];
  __shared__ float qs[4][QW][D];
  const int lane = lane_id(), wave = wave_id();
  const int bh = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * QW;
  const float *qb = q + (long)bh * Lq * D, *kb = k + (long)bh * Lk * D, *vb = v + (long)bh * Lk * D;
  for (int e = lane; e < QW * D; e += 64) {
    const int r = e / D, d = e % D;
    qs[wave][r][d] = (q0 + r < Lq) ? qb[(long)(q0 + r) * D + d] * scale : 0.f;
  }
  float m_run[QW], l_run[QW], o[QW];  // o: this lane's channel (lane < D) of each query row
#pragma unroll
  for (int r = 0; r < QW; ++r) { m_run[r] = -__builtin_inff(); l_run[r] = 0.f; o[r] = 0.f; }

  for (int j0 = 0; j0 < Lk; j0 += TK) {
    __syncthreads();
    for (int e = threadIdx.x; e < TK * D; e += 256) {
      const int r = e / D, d = e % D;
      const bool ok = j0 + r < Lk;
      ks[r][d] = ok ? kb[(long)(j0 + r) * D + d] : 0.f;
      vs[r][d] = ok ? vb[(long)(j0 + r) * D + d] : 0.f;
    }
    __syncthreads();
    const bool kvalid = j0 + lane < Lk;
#pragma unroll
    for (int r = 0; r < QW; ++r) {
      if (q0 + r >= Lq) continue;  // wave-uniform
      float s = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; ++d) s = fmaf(qs[wave][r][d], ks[lane][d], s);
      bool dead = !kvalid;
      if (mask && kvalid) dead = mask[(long)bh * mask_bh_stride + (long)(q0 + r) * Lk + j0 + lane] != 0;
      s = dead ? -__builtin_inff() : s;
      float mx = s;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      const float m_new = fmaxf(m_run[r], mx);
      const float corr = m_new == -__builtin_inff() ? 1.f : __expf(m_run[r] - m_new);
      const float p = (dead || m_new == -__builtin_inff()) ? 0.f : __expf(s - m_new);
      float ps = p;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) ps += __shfl_xor(ps, off);
      l_run[r] = l_run[r] * corr + ps;
      m_run[r] = m_new;
      float acc = o[r] * corr;
      if (lane < D) {
#pragma unroll 8
        for (int t = 0; t < TK; ++t) acc = fmaf(readlane_f(p, t), vs[t][lane], acc);
      }
      o[r] = acc;
    }
  }
#pragma unroll
  for (int r = 0; r < QW; ++r) {
    if (q0 + r >= Lq) continue;
    if (lane < D) out[((long)bh * Lq + q0 + r) * D + lane] = l_run[r] > 0.f ? o[r] / l_run[r] : 0.f;
    if (lse && lane == 0) lse[(long)bh * Lq + q0 + r] = m_run[r] + __logf(l_run[r]);
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_attention_fwd(const float *q, const float *k, const float *v, const uint8_t *mask,
                                 int mask_per_bh, int BH, int Lq, int Lk, int D, float scale, float *out, float *lse,
                                 void *stream) {
  GCN_REQUIRE(q && k && v && out, "gcn_attention_fwd: null pointer");
  GCN_REQUIRE(BH >= 0 && Lq >= 1 && Lk >= 1, "gcn_attention_fwd: bad shape");
  GCN_REQUIRE(D == 8 || D == 16 || D == 32 || D == 64, "gcn_attention_fwd: head dim %d unsupported (8, 16, 32, 64)", D);
  if (BH == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(cdiv(Lq, 64), BH);
  const long ms = mask_per_bh ? (long)Lq * Lk : 0;
  switch (D) {
    case 8: attention_fwd_kernel<8><<<grid, 256, 0, st>>>(q, k, v, mask, ms, Lq, Lk, scale, out, lse); break;
    case 16: attention_fwd_kernel<16><<<grid, 256, 0, st>>>(q, k, v, mask, ms, Lq, Lk, scale, out, lse); break;
    case 32: attention_fwd_kernel<32><<<grid, 256, 0, st>>>(q, k, v, mask, ms, Lq, Lk, scale, out, lse); break;
    default: attention_fwd_kernel<64><<<grid, 256, 0, st>>>(q, k, v, mask, ms, Lq, Lk, scale, out, lse); break;
  }
  return check_launch("attention_fwd_kernel");
}
