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

// multi_mean_square: L = sum_t mean(v_t^2) over several tensors (f32 or bf16) in one pass each way -- the synthetic
// objective bench.py puts on the five outputs of the hot path (every output receives a gradient, as under the
// reference's losses), where torch needs one reduction per tensor forward and two scaling passes backward (~0.15 ms for
// 28 MB).  The tensor list travels in the kernel arguments (no device table: nothing to upload, graph-capturable);
// workgroups take <= MSQ_CHUNK-element chunks; forward: per-chunk partial sums (f64) folded in chunk order by the last
// workgroup to finish (deterministic); backward: grad_t = v_t * (2 / numel_t) * g.
constexpr int MSQ_MAX = 8;
constexpr long MSQ_CHUNK = 16384;
struct MsqArgs {
  const void *v[MSQ_MAX];
  void *grad[MSQ_MAX];
  long numel[MSQ_MAX];
  int chunk0[MSQ_MAX + 1];   // first chunk of tensor t (prefix sums); chunk0[nt] = total
  int bf16[MSQ_MAX];
  int nt;
};

__device__ __forceinline__ int msq_find(const MsqArgs &a, int chunk) {
  int t = 0;
#pragma unroll
  for (int i = 1; i < MSQ_MAX; ++i) t += (i < a.nt && chunk >= a.chunk0[i]) ? 1 : 0;
  return t;
}

__device__ __forceinline__ float msq_load(const void *v, int bf16, long e) {
  return bf16 ? __uint_as_float((unsigned int)((const unsigned short *)v)[e] << 16) : ((const float *)v)[e];
}

__global__ __launch_bounds__(256) void multi_mean_square_fwd_kernel(MsqArgs a, double *__restrict__ part,
                                                                   unsigned int *__restrict__ done, float *__restrict__ loss) {
  const int t = msq_find(a, blockIdx.x);
  const void *v = a.v[t];
  const int bf = a.bf16[t];
  const long n = a.numel[t], first = (long)(blockIdx.x - a.chunk0[t]) * MSQ_CHUNK;
  const long end = first + MSQ_CHUNK < n ? first + MSQ_CHUNK : n;
  const int nseg = a.chunk0[a.nt];
  float acc = 0.f;
  for (long e = first + threadIdx.x; e < end; e += 256) {
    const float x = msq_load(v, bf, e);
    acc = fmaf(x, x, acc);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float wsum[4];
  __shared__ bool last;
  if (lane_id() == 0) wsum[wave_id()] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = ((double)wsum[0] + (double)wsum[1] + (double)wsum[2] + (double)wsum[3]) / (double)n;
    __threadfence();
    last = atomicAdd(done, 1u) == (unsigned int)nseg - 1u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double s = 0.0;                                         // fixed order: lane-strided chunks, then a butterfly
  for (int i = threadIdx.x; i < nseg; i += 256) s += __builtin_nontemporal_load(part + i);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double wd[4];
  if (lane_id() == 0) wd[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    *loss = (float)(wd[0] + wd[1] + wd[2] + wd[3]);
    *done = 0u;                                           // ready for the next call (graph replays included)
  }
}

__global__ __launch_bounds__(256) void multi_mean_square_bwd_kernel(MsqArgs a, const float *__restrict__ gout) {
  const int t = msq_find(a, blockIdx.x);
  const void *v = a.v[t];
  void *gr = a.grad[t];
  const int bf = a.bf16[t];
  const long n = a.numel[t], first = (long)(blockIdx.x - a.chunk0[t]) * MSQ_CHUNK;
  const long end = first + MSQ_CHUNK < n ? first + MSQ_CHUNK : n;
  const float k = (float)(2.0 / (double)n) * gout[0];
  for (long e = first + threadIdx.x; e < end; e += 256) {
    const float g = msq_load(v, bf, e) * k;
    if (bf) {
      const __bf16 h = (__bf16)g;
      ((unsigned short *)gr)[e] = __builtin_bit_cast(unsigned short, h);
    } else {
      ((float *)gr)[e] = g;
    }
  }
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

static int msq_pack(MsqArgs &a, const void *const *v, void *const *grad, const long *numel, const int *is_bf16, int nt) {
  a.nt = nt;
  int c = 0;
  for (int t = 0; t < MSQ_MAX; ++t) {
    const bool on = t < nt;
    a.v[t] = on ? v[t] : nullptr;
    a.grad[t] = (on && grad) ? grad[t] : nullptr;
    a.numel[t] = on ? numel[t] : 1;
    a.bf16[t] = on ? is_bf16[t] : 0;
    a.chunk0[t] = c;
    if (on) c += (int)((numel[t] + MSQ_CHUNK - 1) / MSQ_CHUNK);
  }
  a.chunk0[MSQ_MAX] = c;
  for (int t = nt; t <= MSQ_MAX; ++t) a.chunk0[t] = c;
  return c;
}

GCN_EXPORT int gcn_multi_mean_square_ws_chunks(const long *numel, int nt) {
  long c = 0;
  for (int t = 0; t < nt; ++t) c += (numel[t] + MSQ_CHUNK - 1) / MSQ_CHUNK;
  return (int)c;
}

GCN_EXPORT int gcn_multi_mean_square_fwd(const void *const *v, const long *numel, const int *is_bf16, int nt, double *part,
                                         unsigned int *done, float *loss, void *stream) {
  GCN_REQUIRE(nt >= 0 && nt <= MSQ_MAX, "gcn_multi_mean_square_fwd: 0..%d tensors per call", MSQ_MAX);
  GCN_REQUIRE(loss && (nt == 0 || (v && numel && is_bf16 && part && done)), "gcn_multi_mean_square_fwd: null pointer");
  for (int t = 0; t < nt; ++t) GCN_REQUIRE(numel[t] >= 1 && v[t], "gcn_multi_mean_square_fwd: empty or null tensor %d", t);
  MsqArgs a;
  const int nseg = msq_pack(a, v, nullptr, numel, is_bf16, nt);
  if (nseg == 0) {
    GCN_HIP(fill_dev(loss, 0, sizeof(float), (hipStream_t)stream));
    return GCN_OK;
  }
  multi_mean_square_fwd_kernel<<<nseg, 256, 0, (hipStream_t)stream>>>(a, part, done, loss);
  return check_launch("multi_mean_square_fwd_kernel");
}

GCN_EXPORT int gcn_multi_mean_square_bwd(const void *const *v, void *const *grad, const long *numel, const int *is_bf16, int nt,
                                         const float *grad_loss, void *stream) {
  GCN_REQUIRE(nt >= 0 && nt <= MSQ_MAX, "gcn_multi_mean_square_bwd: 0..%d tensors per call", MSQ_MAX);
  GCN_REQUIRE(nt == 0 || (v && grad && numel && is_bf16 && grad_loss), "gcn_multi_mean_square_bwd: null pointer");
  for (int t = 0; t < nt; ++t) GCN_REQUIRE(numel[t] >= 1 && v[t] && grad[t], "gcn_multi_mean_square_bwd: empty or null tensor %d", t);
  MsqArgs a;
  const int nseg = msq_pack(a, v, grad, numel, is_bf16, nt);
  if (nseg == 0) return GCN_OK;
  multi_mean_square_bwd_kernel<<<nseg, 256, 0, (hipStream_t)stream>>>(a, grad_loss);
  return check_launch("multi_mean_square_bwd_kernel");
}
