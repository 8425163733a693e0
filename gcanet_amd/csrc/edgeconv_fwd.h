// edgeconv_fwd.h -- launcher of the grouped EdgeConv contraction (edgeconv_fwd.hip), shared with edgeconv.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcn {

struct EcqArgs {
  const unsigned short *x;   // (B,NX,Cp) bf16 point-major rows
  const unsigned short *wp;  // (Cout, 2Cp) bf16 = [W1 | W2 - W1]; the kernel contracts the W1 half
  const int64_t *idx;        // (B,N,k) neighbour ids (rows of x within the cloud)
  const float *q;            // (B,N,Cout) f32 centre term (W2 - W1).x_n, or NULL (= 0)
  int B, N, NX, k, Cout, G, TP;
  int tiles_per_cloud, total_tiles;
  float *ymax, *ymin;        // (B,N,Cout)
  unsigned char *amax, *amin;
  double *gsum;              // (B,G,2)
  const float *gamma_route;  // non-null: keep only the extreme GroupNorm+LeakyReLU will route
};

// k <= 128, Cp in {16,32,64,128}, Cout in {64,128}
int launch_edgeconv_fwd_q(EcqArgs &a, int Cp, bool with_arg, hipStream_t st);
// the same on IEEE-half operand images (edgeconv_fwd_f16.hip): Cp in {64,128,256}
int launch_edgeconv_fwd_q_f16(EcqArgs &a, int Cp, bool with_arg, hipStream_t st);

}  // namespace gcn
