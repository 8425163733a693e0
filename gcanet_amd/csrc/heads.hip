// heads.hip -- small per-point epilogues of the prediction heads.
//   param_normalise: the three direction triples of the 22-channel parameter head (plane normal 4:7, cylinder axis
//   8:11, cone axis 15:18) are scaled to unit length, everything else passes through (M4:664-676:
//   `v / (torch.norm(v, dim=-1, keepdim=True).repeat(1,1,3) + 1e-12)` followed by a 7-way torch.cat).  In torch
//   that is ~15 tiny kernels forward and ~40 backward (slice / norm / div / cat and their autograd nodes) on a
//   (B*N, 22) tensor; here one pass each way, one thread per point.
#include "common.h"

namespace gcn {

__device__ __forceinline__ bool is_triple_start(int c) { return c == 4 || c == 8 || c == 15; }

__global__ __launch_bounds__(256) void param_normalise_fwd_kernel(const float *__restrict__ p, long R, float *__restrict__ out) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float v[22];
#pragma unroll
  for (int c = 0; c < 22; ++c) v[c] = p[r * 22 + c];
#pragma unroll
  for (int c = 0; c < 22; ++c) {
    if (is_triple_start(c)) {
      const float nrm = sqrtf(v[c] * v[c] + v[c + 1] * v[c + 1] + v[c + 2] * v[c + 2]) + 1e-12f;
      v[c] /= nrm; v[c + 1] /= nrm; v[c + 2] /= nrm;
    }
  }
#pragma unroll
  for (int c = 0; c < 22; ++c) out[r * 22 + c] = v[c];
}

// y = v / (|v| + eps):  dv = g / (|v|+eps) - v (g.v) / (|v| (|v|+eps)^2)     (dv = g/(|v|+eps) when |v| = 0)
__global__ __launch_bounds__(256) void param_normalise_bwd_kernel(const float *__restrict__ p, const float *__restrict__ go,
                                                                  long R, float *__restrict__ gi) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= R) return;
  float v[22], g[22];
#pragma unroll
  for (int c = 0; c < 22; ++c) { v[c] = p[r * 22 + c]; g[c] = go[r * 22 + c]; }
#pragma unroll
  for (int c = 0; c < 22; ++c) {
    if (is_triple_start(c)) {
      const float n0 = sqrtf(v[c] * v[c] + v[c + 1] * v[c + 1] + v[c + 2] * v[c + 2]);
      const float ne = n0 + 1e-12f;
      const float gv = g[c] * v[c] + g[c + 1] * v[c + 1] + g[c + 2] * v[c + 2];
      const float k2 = n0 > 0.f ? gv / (n0 * ne * ne) : 0.f;
#pragma unroll
      for (int d = 0; d < 3; ++d) g[c + d] = g[c + d] / ne - v[c + d] * k2;
    }
  }
#pragma unroll
  for (int c = 0; c < 22; ++c) gi[r * 22 + c] = g[c];
}

// row_normalise: y = x / |x| over the last dimension (the feature normalisation in front of the offset module's cosine
// similarity, M4:326-342: `f / f.norm(dim=-1, keepdim=True)`, no epsilon -- a zero row gives NaN there and here).
// One wave per row, lane = channel (+ 64 per step); backward dx = (g - y (y.g)) / |x|.  torch: norm + div forward, about
// ten small kernels backward.
__global__ __launch_bounds__(256) void row_normalise_fwd_kernel(const float *__restrict__ x, long R, int C, float *__restrict__ y) {
  const long r = (long)blockIdx.x * 4 + wave_id();
  if (r >= R) return;
  const int lane = lane_id();
  const float *xr = x + r * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(xr[c], xr[c], s);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float n = sqrtf(s);
  for (int c = lane; c < C; c += 64) y[r * C + c] = xr[c] / n;
}

__global__ __launch_bounds__(256) void row_normalise_bwd_kernel(const float *__restrict__ x, const float *__restrict__ go,
                                                                long R, int C, float *__restrict__ gi) {
  const long r = (long)blockIdx.x * 4 + wave_id();
  if (r >= R) return;
  const int lane = lane_id();
  const float *xr = x + r * C, *gr = go + r * C;
  float s = 0.f, d = 0.f;
  for (int c = lane; c < C; c += 64) { s = fmaf(xr[c], xr[c], s); d = fmaf(xr[c], gr[c], d); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); d += __shfl_xor(d, o); }
  const float n = sqrtf(s);
  const float k = d / (n * n * n);                         // x (x.g) / |x|^3
  for (int c = lane; c < C; c += 64) gi[r * C + c] = gr[c] / n - xr[c] * k;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_row_normalise_fwd(const float *x, long R, int C, float *y, void *stream) {
  GCN_REQUIRE(x && y, "gcn_row_normalise_fwd: null pointer");
  GCN_REQUIRE(R >= 0 && C >= 1, "gcn_row_normalise_fwd: bad shape");
  if (R == 0) return GCN_OK;
  row_normalise_fwd_kernel<<<(int)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, R, C, y);
  return check_launch("row_normalise_fwd_kernel");
}

GCN_EXPORT int gcn_row_normalise_bwd(const float *x, const float *grad_out, long R, int C, float *grad_in, void *stream) {
  GCN_REQUIRE(x && grad_out && grad_in, "gcn_row_normalise_bwd: null pointer");
  GCN_REQUIRE(R >= 0 && C >= 1, "gcn_row_normalise_bwd: bad shape");
  if (R == 0) return GCN_OK;
  row_normalise_bwd_kernel<<<(int)((R + 3) / 4), 256, 0, (hipStream_t)stream>>>(x, grad_out, R, C, grad_in);
  return check_launch("row_normalise_bwd_kernel");
}

GCN_EXPORT int gcn_param_normalise_fwd(const float *p, long R, float *out, void *stream) {
  GCN_REQUIRE(p && out, "gcn_param_normalise_fwd: null pointer");
  GCN_REQUIRE(R >= 0, "gcn_param_normalise_fwd: bad shape");
  if (R == 0) return GCN_OK;
  param_normalise_fwd_kernel<<<cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(p, R, out);
  return check_launch("param_normalise_fwd_kernel");
}

GCN_EXPORT int gcn_param_normalise_bwd(const float *p, const float *grad_out, long R, float *grad_in, void *stream) {
  GCN_REQUIRE(p && grad_out && grad_in, "gcn_param_normalise_bwd: null pointer");
  GCN_REQUIRE(R >= 0, "gcn_param_normalise_bwd: bad shape");
  if (R == 0) return GCN_OK;
  param_normalise_bwd_kernel<<<cdiv(R, 256), 256, 0, (hipStream_t)stream>>>(p, grad_out, R, grad_in);
  return check_launch("param_normalise_bwd_kernel");
}
