// normaledge.hip -- the 7-channel normal-feature EdgeConv of the reference (get_graph_feature_with_normals_g,
// M4:164-205: edge feature [clamp(n_i.n_j, +-0.99), n_j - n_i, n_i]; Conv2d(7->64,1x1) + GroupNorm + LeakyReLU +
// max over k, M4:575-577,691-693) without the (B,7,N,k) tensor (117 MB at B=8, N=8192, k=64) or its gather /
// cat / permute passes: the edge feature is rebuilt in registers from the point rows, forward and backward.
// K = 7 is far too thin for the matrix cores (a 32x32x16 tile would be 56 % padding): exact f32 FMAs on the VALU,
// which also makes this block independent of the bf16 switch.
#include "common.h"

namespace gcn {

constexpr int NE_F = 7;

// pts (B,N,6) point-major [xyz, normal]; idx (B,N,k) int64; W (Cout,7) f32.  Outputs as gcn_edgeconv_fwd:
// ymax/ymin (B,N,Cout), amax/amin (B,N,Cout) u8 slots, gsum (B,G,2) f64 sums of y and y^2.  k <= 256.
// The neighbour loop is VALU-bound: it runs on PAIRS of neighbours with packed f32 (v_pk_fma_f32: the four taps, the
// sum and the sum of squares of two neighbours per instruction; the pairs lie in LDS as [x_j, x_j+1] so that a
// broadcast read delivers them in adjacent registers), and in ROUTED mode (gamma_route given, see keyedge_fwd_kernel)
// only the extreme the sign of gamma selects is tracked, with the sign folded into the weights (exact).
typedef float ne_f2 __attribute__((ext_vector_type(2)));

// Workgroups of 16 waves: the GroupNorm sums leave a workgroup as ONE device atomic per (group, statistic), and all
// workgroups reach that point together -- with 4-wave workgroups 128 of them queued on each address (~0.4 us apiece).
template <bool ROUTED>
__global__ __launch_bounds__(1024) void normal_edge_fwd_kernel(const float *__restrict__ pts, const int64_t *__restrict__ idx,
                                                              const float *__restrict__ W, int N, int k, int Cout, int G,
                                                              int pts_per_block, float *__restrict__ ymax,
                                                              float *__restrict__ ymin, unsigned char *__restrict__ amax,
                                                              unsigned char *__restrict__ amin, double *__restrict__ gsum,
                                                              const float *__restrict__ gamma_route) {
  __shared__ double red[128];                 // (group, stat) partial sums of this workgroup, G <= 64
  extern __shared__ float4 efs[];             // [wave][2][P]: per neighbour pair {ang_j, ang_j+1, d0_j, d0_j+1}, {d1.., d2..}
  const int lane = lane_id(), wave = wave_id();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int n_lo = tile * pts_per_block, n_hi = min(n_lo + pts_per_block, N);
  const int cpg = Cout / G;
  const float *pb = pts + (long)b * N * 6;
  if (threadIdx.x < 2 * G) red[threadIdx.x] = 0.0;
  __syncthreads();
  const int P = (k + 1) >> 1;                 // neighbour pairs
  const float4 *pa = efs + (long)wave * 2 * P, *pbq = pa + P;
  float *efa = reinterpret_cast<float *>(efs + (long)wave * 2 * P), *efb = efa + 4 * P;
  for (int c0 = 0; c0 < Cout; c0 += 64) {
    const int c = min(c0 + lane, Cout - 1);
    const bool cv = c0 + lane < Cout;
    const bool neg = ROUTED && gamma_route[c] < 0.f;
    float w[NE_F];
#pragma unroll
    for (int f = 0; f < NE_F; ++f) {
      w[f] = W[c * NE_F + f];
      if (neg) w[f] = -w[f];
    }
    ne_f2 w2[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) w2[f] = ne_f2{w[f], w[f]};
    ne_f2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
    for (int n = n_lo + wave; n < n_hi; n += 16) {
      const long pn = (long)b * N + n;
      const float ni0 = pb[(long)n * 6 + 3], ni1 = pb[(long)n * 6 + 4], ni2 = pb[(long)n * 6 + 5];
      float ang[4], d0[4], d1[4], d2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = min(q * 64 + lane, k - 1);
        const long m = idx[pn * k + jj];
        const float a0 = pb[m * 6 + 3], a1 = pb[m * 6 + 4], a2 = pb[m * 6 + 5];
        const float dot = (ni0 * a0 + ni1 * a1) + ni2 * a2;
        ang[q] = fminf(fmaxf(dot, -0.99f), 0.99f);
        d0[q] = a0 - ni0; d1[q] = a1 - ni1; d2[q] = a2 - ni2;
        if (q * 64 + 64 >= k) break;                                  // wave-uniform
      }
      // The k edge features go through LDS and come back as broadcast reads (four v_readlane per neighbour before),
      // and the centre-normal taps, constant over the neighbours, are summed once per (point, channel).
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = q * 64 + lane;
        if (j < k) {
          const int o = (j >> 1) * 4 + (j & 1);
          efa[o] = ang[q]; efa[o + 2] = d0[q];
          efb[o] = d1[q]; efb[o + 2] = d2[q];
        }
        if (q * 64 + 64 >= k) break;
      }
      __builtin_amdgcn_wave_barrier();
      const float base = fmaf(w[6], ni2, fmaf(w[5], ni1, w[4] * ni0));
      const ne_f2 base2 = {base, base};
      float mx = -__builtin_inff(), mn = __builtin_inff();
      int ax = 0, an = 0;
      auto track = [&](float y, int j) {
        if (y > mx) { mx = y; ax = j; }
        if (!ROUTED && y < mn) { mn = y; an = j; }
      };
      int j = 0;
      for (; j + 1 < k; j += 2) {
        const float4 ea = pa[j >> 1], eb = pbq[j >> 1];
        ne_f2 y = __builtin_elementwise_fma(w2[0], ne_f2{ea.x, ea.y}, base2);
        y = __builtin_elementwise_fma(w2[1], ne_f2{ea.z, ea.w}, y);
        y = __builtin_elementwise_fma(w2[2], ne_f2{eb.x, eb.y}, y);
        y = __builtin_elementwise_fma(w2[3], ne_f2{eb.z, eb.w}, y);
        track(y.x, j);
        track(y.y, j + 1);
        s1 += y;
        s2 = __builtin_elementwise_fma(y, y, s2);
      }
      if (j < k) {                                                    // odd k: the last neighbour alone
        const float4 ea = pa[j >> 1], eb = pbq[j >> 1];
        float y = fmaf(w[0], ea.x, base);
        y = fmaf(w[1], ea.z, y); y = fmaf(w[2], eb.x, y); y = fmaf(w[3], eb.z, y);
        track(y, j);
        s1.x += y;
        s2.x = fmaf(y, y, s2.x);
      }
      if (cv) {
        if (ROUTED) {
          ymax[pn * Cout + c] = neg ? -mx : mx;
          amax[pn * Cout + c] = (unsigned char)ax;
        } else {
          ymax[pn * Cout + c] = mx; ymin[pn * Cout + c] = mn;
          amax[pn * Cout + c] = (unsigned char)ax; amin[pn * Cout + c] = (unsigned char)an;
        }
      }
    }
    float t1f = s1.x + s1.y, t2f = s2.x + s2.y;
    if (neg) t1f = -t1f;                                              // the sums are those of y itself
    if (!cv) { t1f = 0.f; t2f = 0.f; }
    // one f64 atomic pair per GroupNorm group and wave (same-address f64 atomics serialise at ~0.45 us each:
    // a per-lane version of this epilogue cost 7 ms)
    const int seg = (cpg % 64) == 0 ? 64 : cpg;          // lanes per group inside this 64-channel chunk
    if ((seg & (seg - 1)) == 0 && seg <= 64) {
      double t1 = (double)t1f, t2 = (double)t2f;
      for (int o = seg >> 1; o >= 1; o >>= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
      if ((lane & (seg - 1)) == 0 && cv) {
        atomicAdd(&red[(c / cpg) * 2], t1);
        atomicAdd(&red[(c / cpg) * 2 + 1], t2);
      }
    } else if (cv) {
      atomicAdd(&red[(c / cpg) * 2], (double)t1f);
      atomicAdd(&red[(c / cpg) * 2 + 1], (double)t2f);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * G) atomicAdd(gsum + (long)b * G * 2 + threadIdx.x, red[threadIdx.x]);
}

// Weight-gradient pieces of the block for dy = coef*[j == jsel] + Ac + Bc*y (the inputs carry no gradient):
//   dWsp[b,c,f] = sum_n coef[n,c] * ef[n, jsel[n,c], f];   esum[b,f] = sum_{n,j} ef;   gram[b,f,g] = sum_{n,j} ef_f ef_g
// (dW = sum_b dWsp_b + Ac^T esum + sum_b Bc_b o (W gram_b) is finished by the caller: 64x7 numbers).  k <= 64 per pass.
// 16-wave workgroups and per-CLOUD accumulators: every workgroup closes with one device atomic per output, and they all
// get there together -- 512 four-wave workgroups on the same 448 addresses spent most of the kernel's 72 us queueing.
__global__ __launch_bounds__(1024) void normal_edge_bwd_kernel(const float *__restrict__ pts, const int64_t *__restrict__ idx,
                                                              const float *__restrict__ coef, const int64_t *__restrict__ jsel,
                                                              int N, int k, int Cout, int pts_per_block,
                                                              float *__restrict__ dWsp, float *__restrict__ esum,
                                                              float *__restrict__ gram) {
  const int lane = lane_id(), wave = wave_id();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int n_lo = tile * pts_per_block, n_hi = min(n_lo + pts_per_block, N);
  const float *pb = pts + (long)b * N * 6;
  float es[NE_F], gr[28];
#pragma unroll
  for (int f = 0; f < NE_F; ++f) es[f] = 0.f;
#pragma unroll
  for (int f = 0; f < 28; ++f) gr[f] = 0.f;
  float dw[2][NE_F];                       // Cout <= 128: two channel chunks per lane
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int f = 0; f < NE_F; ++f) dw[h][f] = 0.f;

  for (int n = n_lo + wave; n < n_hi; n += 16) {
    const long pn = (long)b * N + n;
    const float ni0 = pb[(long)n * 6 + 3], ni1 = pb[(long)n * 6 + 4], ni2 = pb[(long)n * 6 + 5];
    for (int j0 = 0; j0 < k; j0 += 64) {
      const bool jv = j0 + lane < k;
      const long m = idx[pn * k + min(j0 + lane, k - 1)];
      const float a0 = pb[m * 6 + 3], a1 = pb[m * 6 + 4], a2 = pb[m * 6 + 5];
      const float dot = (ni0 * a0 + ni1 * a1) + ni2 * a2;
      float e[NE_F];
      e[0] = fminf(fmaxf(dot, -0.99f), 0.99f);
      e[1] = a0 - ni0; e[2] = a1 - ni1; e[3] = a2 - ni2;
      e[4] = ni0; e[5] = ni1; e[6] = ni2;
      if (jv) {
        int t = 0;
#pragma unroll
        for (int f = 0; f < NE_F; ++f) {
          es[f] += e[f];
#pragma unroll
          for (int g = f; g < NE_F; ++g) { gr[t] = fmaf(e[f], e[g], gr[t]); ++t; }
        }
      }
      // routed part: lane = channel picks the edge its maximum came from
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = h * 64 + lane;
        if (h * 64 < Cout) {
          const int cc = min(c, Cout - 1);
          const int js = (int)jsel[pn * Cout + cc];
          const float cf = (c < Cout && js >= j0 && js < j0 + 64) ? coef[pn * Cout + cc] : 0.f;
          const int src = (js - j0) & 63;
#pragma unroll
          for (int f = 0; f < 4; ++f) dw[h][f] = fmaf(cf, __shfl(e[f], src), dw[h][f]);
          dw[h][4] = fmaf(cf, ni0, dw[h][4]);
          dw[h][5] = fmaf(cf, ni1, dw[h][5]);
          dw[h][6] = fmaf(cf, ni2, dw[h][6]);
        }
      }
    }
  }
  // workgroup-level sums in LDS first
  __shared__ float red[NE_F + NE_F * NE_F];
  float *res = red, *rgr = res + NE_F;
  for (int i = threadIdx.x; i < NE_F + NE_F * NE_F; i += 1024) red[i] = 0.f;
  __syncthreads();
  // per-wave slots, added in wave order on the way out (7 ds_add_f32 per lane from 16 waves onto the same words were
  // ~7000 LDS atomics per workgroup at 0.33 lanes/clk; stride 7 between lanes is conflict free)
  __shared__ float wdw[16][128 * NE_F];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = h * 64 + lane;
    if (c < Cout) {
#pragma unroll
      for (int f = 0; f < NE_F; ++f) wdw[wave_id()][c * NE_F + f] = dw[h][f];
    }
  }
#pragma unroll
  for (int f = 0; f < NE_F; ++f) {
    float v = es[f];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) atomicAdd(&res[f], v);
  }
  int t = 0;
#pragma unroll
  for (int f = 0; f < NE_F; ++f)
#pragma unroll
    for (int g = f; g < NE_F; ++g) {
      float v = gr[t++];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) {
        atomicAdd(&rgr[f * NE_F + g], v);
        if (g != f) atomicAdd(&rgr[g * NE_F + f], v);
      }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < Cout * NE_F; i += 1024) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += wdw[w][i];
    atomicAdd(dWsp + (long)b * Cout * NE_F + i, t);
  }
  if (threadIdx.x < NE_F) atomicAdd(esum + (long)b * NE_F + threadIdx.x, res[threadIdx.x]);
  if (threadIdx.x < NE_F * NE_F) atomicAdd(gram + (long)b * NE_F * NE_F + threadIdx.x, rgr[threadIdx.x]);
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_normal_edge_fwd(const float *pts, const int64_t *idx, const float *W, int B, int N, int k, int Cout, int G,
                                   float *ymax, float *ymin, uint8_t *amax, uint8_t *amin, double *gsum,
                                   const float *gamma_route, void *stream) {
  GCN_REQUIRE(pts && idx && W && ymax && amax && gsum, "gcn_normal_edge_fwd: null pointer");
  GCN_REQUIRE(gamma_route || (ymin && amin), "gcn_normal_edge_fwd: ymin / amin may be NULL only in routed mode (gamma_route given)");
  GCN_REQUIRE(B >= 0 && N >= 1 && k >= 1 && k <= 256 && Cout >= 1 && G >= 1 && G <= 64 && Cout % G == 0, "gcn_normal_edge_fwd: bad shape");
  if (B == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GCN_HIP(zero_dev(gsum, sizeof(double) * 2 * B * G, st));
  int blocks_per_cloud = (256 + B - 1) / B;                // one 16-wave workgroup per CU
  if (blocks_per_cloud > (N + 15) / 16) blocks_per_cloud = (N + 15) / 16;
  const int ppb = (N + blocks_per_cloud - 1) / blocks_per_cloud;
  const size_t lds = sizeof(float4) * 16 * 2 * (size_t)((k + 1) / 2);
  GCN_HIP(hipFuncSetAttribute((const void *)normal_edge_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  GCN_HIP(hipFuncSetAttribute((const void *)normal_edge_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (gamma_route)
    normal_edge_fwd_kernel<true><<<dim3(cdiv(N, ppb), B), 1024, lds, st>>>(pts, idx, W, N, k, Cout, G, ppb, ymax, ymin, amax, amin, gsum, gamma_route);
  else
    normal_edge_fwd_kernel<false><<<dim3(cdiv(N, ppb), B), 1024, lds, st>>>(pts, idx, W, N, k, Cout, G, ppb, ymax, ymin, amax, amin, gsum, gamma_route);
  return check_launch("normal_edge_fwd_kernel");
}

GCN_EXPORT int gcn_normal_edge_bwd(const float *pts, const int64_t *idx, const float *coef, const int64_t *jsel, int B, int N,
                                   int k, int Cout, float *dWsp, float *esum, float *gram, void *stream) {
  GCN_REQUIRE(pts && idx && coef && jsel && dWsp && esum && gram, "gcn_normal_edge_bwd: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && k >= 1 && Cout >= 1 && Cout <= 128, "gcn_normal_edge_bwd: bad shape (Cout <= 128)");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) return GCN_OK;
  GCN_HIP(zero_spans(st, {dWsp, sizeof(float) * (size_t)B * Cout * NE_F}, {esum, sizeof(float) * B * NE_F}, {gram, sizeof(float) * B * NE_F * NE_F}));
  int blocks_per_cloud = (256 + B - 1) / B;
  if (blocks_per_cloud > (N + 15) / 16) blocks_per_cloud = (N + 15) / 16;
  const int ppb = (N + blocks_per_cloud - 1) / blocks_per_cloud;
  normal_edge_bwd_kernel<<<dim3(cdiv(N, ppb), B), 1024, 0, st>>>(pts, idx, coef, jsel, N, k, Cout, ppb, dWsp, esum, gram);
  return check_launch("normal_edge_bwd_kernel");
}
