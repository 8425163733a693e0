// edgeconv_fwd_f16.hip -- the EdgeConv matrix-core kernels of edgeconv_fwd_impl.h on IEEE-half operands
// (v_mfma_f32_32x32x16_f16): BASELINE configs[4] names "fp16+MFMA" for its C = 256 layer.  Same kernels, same f32
// accumulation and epilogue; the operand images come from gcn_edgeconv_pack_*16(..., half = 1).  Instantiated for 33..256
// input channels (Cp in {64,128,256}); narrower layers keep bf16 or f32.  A separate translation unit so that the 80
// extra kernel instantiations compile next to the bf16 ones, not after them.
// hipcc-flags: -fno-honor-nans
#include "edgeconv_fwd_impl.h"

namespace gcn {

int launch_edgeconv_fwd_q_f16(EcqArgs &a, int Cp, bool with_arg, hipStream_t st) { return launch_fwd_q_t<true>(a, Cp, with_arg, st); }

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_edgeconv_center_f16(const void *x_pm_f16, const void *wp_f16, long rows, int C, int Cout, float *q,
                                       void *stream) {
  GCN_REQUIRE(x_pm_f16 && wp_f16 && q, "gcn_edgeconv_center_f16: null pointer");
  GCN_REQUIRE(rows >= 0 && C >= 33 && C <= 256, "gcn_edgeconv_center_f16: bad shape (33 <= C <= 256)");
  GCN_REQUIRE(Cout == 64 || Cout == 128, "gcn_edgeconv_center_f16: Cout must be 64 or 128, got %d", Cout);
  if (rows == 0) return GCN_OK;
  return launch_center<true>(x_pm_f16, wp_f16, rows, C, Cout, q, (hipStream_t)stream);
}
