// optim.hip -- Adam (torch.optim.Adam of the reference's trainer, option_new.py:83-90: no weight decay by default,
// no amsgrad) over ONE flat parameter / gradient / moment buffer.  The model has 57 parameter tensors with 1.5 M
// elements: torch's multi-tensor kernels take ~0.2 ms per step for 42 MB of traffic; with the gradients already packed
// by parallel.FlatGradDP and the parameters made views of one buffer (gcanet_amd/optim.py) it is one 16-byte-wide
// elementwise pass.  The step count lives on the device so that the call can sit inside a captured HIP graph.
#include "common.h"

namespace gcn {

// state[0] = t (as float, for readers), state[1] = 1 - b1^t, state[2] = 1 - b2^t, state[3] = t as a 32-bit integer (bits).
// The corrections are computed in double as torch.optim.Adam computes them on the host: 1 - 0.999^t in f32 loses ~6e-5
// of relative accuracy at small t, and a float counter stops counting at 2^24.
__global__ void adam_prepare_kernel(float *state, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  unsigned int t = __float_as_uint(state[3]);
  if (t == 0u && state[0] > 0.f) t = (unsigned int)state[0];      // a state written by the CPU path / an older version
  t += 1u;
  state[3] = __uint_as_float(t);
  state[0] = (float)t;
  state[1] = (float)(1.0 - pow((double)b1, (double)t));
  state[2] = (float)(1.0 - pow((double)b2, (double)t));
}

__global__ __launch_bounds__(256) void adam_flat_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                        float wd, const float *__restrict__ state) {
  const float bc1 = state[1], bc2 = state[2];
  const float step_size = lr / bc1, rs2 = 1.f / sqrtf(bc2);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4 *>(p)[i], mv = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
    const float4 gv = reinterpret_cast<const float4 *>(g)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w}, ww[4] = {vv.x, vv.y, vv.z, vv.w};
    const float gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] + wd * pp[e];
      mm[e] = mm[e] + (gr - mm[e]) * (1.f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
      ww[e] = ww[e] * b2 + (1.f - b2) * gr * gr;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(ww[e]) * rs2 + eps;
      pp[e] = pp[e] - step_size * (mm[e] / denom);
    }
    reinterpret_cast<float4 *>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4 *>(m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    reinterpret_cast<float4 *>(v)[i] = make_float4(ww[0], ww[1], ww[2], ww[3]);
  }
  // tail (n % 4 elements) by the first threads of workgroup 0
  if (blockIdx.x == 0 && (long)threadIdx.x < n - (n4 << 2)) {
    const long i = (n4 << 2) + threadIdx.x;
    const float gr = g[i] + wd * p[i];
    const float mn = m[i] + (gr - m[i]) * (1.f - b1);
    const float vn = v[i] * b2 + (1.f - b2) * gr * gr;
    m[i] = mn; v[i] = vn;
    p[i] = p[i] - step_size * (mn / (sqrtf(vn) * rs2 + eps));
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_adam_flat(float *p, const float *g, float *m, float *v, long n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float *state, void *stream) {
  GCN_REQUIRE(p && g && m && v && state, "gcn_adam_flat: null pointer");
  GCN_REQUIRE(n >= 0 && lr >= 0.f && beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "gcn_adam_flat: bad hyper-parameter");
  GCN_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "gcn_adam_flat: buffers must be 16-byte aligned");
  if (n == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  adam_prepare_kernel<<<1, 64, 0, st>>>(state, beta1, beta2);
  const long n4 = n >> 2;
  const int grid = (int)(n4 / 256 + 1 > 2048 ? 2048 : n4 / 256 + 1);
  adam_flat_kernel<<<grid, 256, 0, st>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, state);
  return check_launch("adam_flat_kernel");
}
