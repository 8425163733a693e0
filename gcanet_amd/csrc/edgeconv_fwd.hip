// edgeconv_fwd.hip -- the grouped (N*k, 2C) x (2C, Cout) contraction of the DGCNN EdgeConv block
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
// hipcc-flags: -fno-honor-nans
#include "edgeconv_fwd_impl.h"

namespace gcn {

int launch_edgeconv_fwd_q(EcqArgs &a, int Cp, bool with_arg, hipStream_t st) { return launch_fwd_q_t<false>(a, Cp, with_arg, st); }

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_edgeconv_center(const void *x_pm_bf16, const void *wp_bf16, long rows, int C, int Cout, float *q,
                                   void *stream) {
  GCN_REQUIRE(x_pm_bf16 && wp_bf16 && q, "gcn_edgeconv_center: null pointer");
  GCN_REQUIRE(rows >= 0 && C >= 1 && C <= 256, "gcn_edgeconv_center: bad shape (C <= 256)");
  GCN_REQUIRE(Cout == 64 || Cout == 128, "gcn_edgeconv_center: Cout must be 64 or 128, got %d", Cout);
  if (rows == 0) return GCN_OK;
  return launch_center<false>(x_pm_bf16, wp_bf16, rows, C, Cout, q, (hipStream_t)stream);
}
