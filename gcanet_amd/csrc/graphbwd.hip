// graphbwd.hip -- fused pieces of the closed-form backward of the grouped blocks
// (max_k LeakyReLU(GroupNorm(conv(edge features)))), see gcanet_amd/dgcnn.py:
//   route_bwd_kernel     one pass over (B,N,Cout): selected extreme -> yhat, z, routed gradient;
//                        per-channel dgamma/dbeta, per-(cloud,group) S1/S2, coef = rstd*gamma*g,
//                        the selected neighbour slot / id and the sparse scatter Dsp[m_sel] += coef
//   edge_combine_kernel  D1/D2 (the per-point sums of dy over incoming / outgoing edges)
//   edge_wgrad_kernel    every row reduction of the weight gradient in one f32-MFMA pass
//   (the transposed aggregation r = Adj^T.x lives in edgeconv.hip: reverse_sum_lds_kernel)
// The reference has no counterpart kernels: its autograd walks the materialised (B,Cout,N,k) tensor.
#include "common.h"

namespace gcn {

__global__ __launch_bounds__(256) void route_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ ymax,
                                                        const float *__restrict__ ymin, const unsigned char *__restrict__ amax,
                                                        const unsigned char *__restrict__ amin, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *__restrict__ mean_rstd,
                                                        const int64_t *__restrict__ idx, int N, int k, int Cout, int G,
                                                        float slope, int rows_per_block, float *__restrict__ coef,
                                                        int64_t *__restrict__ jsel, int64_t *__restrict__ msel,
                                                        float *__restrict__ dsp, float *__restrict__ dgamma,
                                                        float *__restrict__ dbeta, double *__restrict__ S) {
  extern __shared__ double sm[];             // 2*G doubles, then 2*Cout floats
  float *cs = reinterpret_cast<float *>(sm + 2 * G);
  for (int i = threadIdx.x; i < 2 * G; i += 256) sm[i] = 0.0;
  for (int i = threadIdx.x; i < 2 * Cout; i += 256) cs[i] = 0.f;
  __syncthreads();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int r0 = tile * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const int cpg = Cout / G;
  // thread -> fixed channel (Cout <= 256 and divides 256, or a multiple of 256)
  const int nct = Cout <= 256 ? Cout : 256;
  const int rstep = 256 / nct > 0 ? 256 / nct : 1;
  for (int c = threadIdx.x % nct; c < Cout; c += 256) {
    const int g = c / cpg;
    const float mean = mean_rstd[((long)b * G + g) * 2], rstd = mean_rstd[((long)b * G + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    const bool pos = ga >= 0.f;
    float dg = 0.f, db = 0.f, s1 = 0.f, s2 = 0.f;
    for (int n = r0 + threadIdx.x / nct; n < r1; n += rstep) {
      const long o = ((long)b * N + n) * Cout + c;
      const float ys = (pos || ymin == nullptr) ? ymax[o] : ymin[o];   // ymin == NULL: routed forward
      const int js = (pos || amin == nullptr) ? amax[o] : amin[o];
      const float yh = (ys - mean) * rstd;
      const float z = yh * ga + be;
      const float gz = dout[o] * (z > 0.f ? 1.f : slope);
      const float t = gz * ga;
      db += gz;
      dg = fmaf(gz, yh, dg);
      s1 += t;
      s2 = fmaf(t, yh, s2);
      const float cf = t * rstd;
      coef[o] = cf;
      if (jsel) jsel[o] = js;
      if (idx) {
        const int64_t m = idx[((long)b * N + n) * k + js];
        if (msel) msel[o] = m;
        if (dsp) atomicAdd(dsp + ((long)b * N + m) * Cout + c, cf);
      }
    }
    atomicAdd(&cs[c], dg);
    atomicAdd(&cs[Cout + c], db);
    atomicAdd(&sm[g * 2], (double)s1);
    atomicAdd(&sm[g * 2 + 1], (double)s2);
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * G) atomicAdd(S + (long)b * G * 2 + threadIdx.x, sm[threadIdx.x]);
  for (int i = threadIdx.x; i < Cout; i += 256) {
    atomicAdd(dgamma + i, cs[i]);
    atomicAdd(dbeta + i, cs[Cout + i]);
  }
}

// Affine part of the GroupNorm backward per (cloud, channel), in double like the statistics:
//   Bg = -(rstd^2) * S2 / M,  Ag = -(rstd * S1) / M - Bg * mean;   dy += Ac + Bc * y
__global__ void gn_affine_kernel(const float *__restrict__ mean_rstd, const double *__restrict__ S, int BC, int Cout,
                                 int G, double count, float *__restrict__ Ac, float *__restrict__ Bc) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BC) return;
  const int b = e / Cout, g = (e % Cout) / (Cout / G);
  const double mu = (double)mean_rstd[((long)b * G + g) * 2], rs = (double)mean_rstd[((long)b * G + g) * 2 + 1];
  const double Bg = (-(rs * rs)) * S[((long)b * G + g) * 2 + 1] / count;
  const double Ag = (-(rs * S[((long)b * G + g) * 2])) / count - Bg * mu;
  Ac[e] = (float)Ag;
  Bc[e] = (float)Bg;
}

// D2[n,c] = coef + k*A + B*(SW + k*XW) ; D1[m,c] = Dsp + indeg*(A + B*P1) + B*RW      (all (B,N,Cout))
__global__ __launch_bounds__(256) void edge_combine_kernel(const float *__restrict__ coef, const float *__restrict__ dsp,
                                                           const float *__restrict__ indeg, const float *__restrict__ Ac,
                                                           const float *__restrict__ Bc, const float *__restrict__ P1,
                                                           const float *__restrict__ SW, const float *__restrict__ XW,
                                                           const float *__restrict__ RW, int N, int Cout, float kf,
                                                           float *__restrict__ D1, float *__restrict__ D2) {
  const int b = blockIdx.y;
  const long per = (long)N * Cout;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < per; e += (long)gridDim.x * 256) {
    const int c = (int)(e % Cout);
    const long n = e / Cout;
    const long o = (long)b * per + e;
    const float a = Ac[(long)b * Cout + c], bb = Bc[(long)b * Cout + c];
    const float dg = indeg[(long)b * N + n];
    D2[o] = coef[o] + kf * a + bb * (SW[o] + kf * XW[o]);
    D1[o] = dsp[o] + dg * (a + bb * P1[o]) + bb * RW[o];
  }
}

// ------------------------------------------------------------------ fused weight-gradient reductions
// All row-reductions the closed-form EdgeConv backward needs, in ONE pass over the point rows, on the f32
// matrix cores (v_mfma_f32_16x16x4_f32: A[i][k] and B[k][j] are both "row n0+k, column 16t+(lane&15)" reads of
// the row-major operands, so no transposes are needed for X^T.Y products):
//   M1 += Dsp^T x, M2 += D2^T x  (Cout x C, summed over clouds);  G11[b] = x^T diag(indeg) x,  G21[b] = x^T s
//   (C x C per cloud);  ssum[b] = sum_n s[n,:].
// Workgroup = 4 waves on one row chunk of one cloud; wave w owns output-row tiles {w*OW..} of M1/M2 and
// c-tile w of G11/G21; partial tiles are added with f32 atomics (256 workgroups -> ~8k adds per element set).
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;

template <int CT, int OT>
__global__ __launch_bounds__(256) void edge_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ sx,
                                                         const float *__restrict__ dsp, const float *__restrict__ d2,
                                                         const float *__restrict__ indeg, int N, int C, int Cout,
                                                         int rows_per_block, float *__restrict__ M1, float *__restrict__ M2,
                                                         float *__restrict__ G11, float *__restrict__ G21,
                                                         float *__restrict__ ssum) {
  constexpr int OW = OT / 4;                   // o-tiles per wave
  constexpr bool HAS_G = true;
  const int lane = lane_id(), wave = wave_id();
  const int li = lane & 15, lk = lane >> 4;
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int r0 = tile * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const float *xb = x + (long)b * N * C, *sb = sx + (long)b * N * C;
  const float *pb = dsp + (long)b * N * Cout, *qb = d2 + (long)b * N * Cout;
  const float *ib = indeg + (long)b * N;
  const bool gwave = wave < CT;                // waves that own a c-tile of G11/G21 (CT = 1: wave 0 only)

  wg_f32x4 m1[OW][CT], m2[OW][CT], g11[CT], g21[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
#pragma unroll
    for (int o = 0; o < OW; ++o) { m1[o][t] = {0.f, 0.f, 0.f, 0.f}; m2[o][t] = {0.f, 0.f, 0.f, 0.f}; }
    g11[t] = {0.f, 0.f, 0.f, 0.f};
    g21[t] = {0.f, 0.f, 0.f, 0.f};
  }
  float ss[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) ss[t] = 0.f;

  // The loop is latency-bound (13 small loads, then 24 MFMAs): operands of the NEXT four rows are fetched before
  // the current MFMAs issue.  Loads are unconditional on clamped addresses and masked afterwards (predicated
  // loads serialise: one branch + s_waitcnt vmcnt(0) each).
  float xn[CT], sn[CT], pn[OW], qn[OW], dgn;
  auto fetch = [&](int n0) {
    const int n = min(n0 + lk, r1 - 1);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int c = min(16 * t + li, C - 1);
      xn[t] = xb[(long)n * C + c];
      sn[t] = sb[(long)n * C + c];
    }
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      const int oc = 16 * (wave * OW + o) + li;
      pn[o] = pb[(long)n * Cout + oc];
      qn[o] = qb[(long)n * Cout + oc];
    }
    dgn = ib[n];
  };
  if (r0 < r1) fetch(r0);
  for (int n0 = r0; n0 < r1; n0 += 4) {
    const bool ok = n0 + lk < r1;
    float xv[CT], sv[CT], pa[OW], qa[OW];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const bool okc = ok && 16 * t + li < C;
      xv[t] = okc ? xn[t] : 0.f;
      sv[t] = okc ? sn[t] : 0.f;
    }
#pragma unroll
    for (int o = 0; o < OW; ++o) {
      pa[o] = ok ? pn[o] : 0.f;
      qa[o] = ok ? qn[o] : 0.f;
    }
    const float dg = ok ? dgn : 0.f;
    if (n0 + 4 < r1) fetch(n0 + 4);
#pragma unroll
    for (int t = 0; t < CT; ++t) {
#pragma unroll
      for (int o = 0; o < OW; ++o) {
        m1[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[o], xv[t], m1[o][t], 0, 0, 0);
        m2[o][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[o], xv[t], m2[o][t], 0, 0, 0);
      }
    }
    if (HAS_G && gwave) {
      float xa = 0.f;
#pragma unroll
      for (int t = 0; t < CT; ++t) xa = (t == wave) ? xv[t] : xa;       // this wave's c-tile as the A operand
      const float xd = xa * dg;
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        g11[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xd, xv[t], g11[t], 0, 0, 0);
        g21[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, sv[t], g21[t], 0, 0, 0);
      }
    }
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < CT; ++t) ss[t] += sv[t];
    }
  }
  // C/D layout of 16x16: col = lane&15, row = 4*(lane>>4) + r
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const int c = 16 * t + li;
    if (c < C) {
#pragma unroll
      for (int o = 0; o < OW; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oc = 16 * (wave * OW + o) + 4 * lk + r;
          atomicAdd(M1 + (long)oc * C + c, m1[o][t][r]);
          atomicAdd(M2 + (long)oc * C + c, m2[o][t][r]);
        }
      if (gwave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cr = 16 * wave + 4 * lk + r;
          if (cr < C) {
            atomicAdd(G11 + ((long)b * C + cr) * C + c, g11[t][r]);
            atomicAdd(G21 + ((long)b * C + cr) * C + c, g21[t][r]);
          }
        }
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      float v = ss[t];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int c = 16 * t + li;
      if (lk == 0 && c < C) atomicAdd(ssum + (long)b * C + c, v);
    }
  }
}

// dW (Cout, 2C) = [dW1 - dWd | dWd],  dWd = M2,
// dW1[o,d] = M1[o,d] + sum_b Ac[b,o] ssum[b,d] + sum_b Bc[b,o] sum_c (W1[o,c] G11[b,c,d] + Wd[o,c] G21[b,c,d])
// One workgroup per output row o: W row in LDS, thread = (d, cloud slice); G reads coalesced over d.
__global__ __launch_bounds__(256) void edge_wgrad_finish_kernel(const float *__restrict__ W, const float *__restrict__ Ac,
                                                                const float *__restrict__ Bc, const float *__restrict__ M1,
                                                                const float *__restrict__ M2, const float *__restrict__ G11,
                                                                const float *__restrict__ G21, const float *__restrict__ ssum,
                                                                int B, int C, int Cout, float *__restrict__ dW) {
  __shared__ float w1[64], wd[64], part[256];
  const int o = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    w1[c] = W[(long)o * 2 * C + c];
    wd[c] = W[(long)o * 2 * C + C + c] - w1[c];
  }
  __syncthreads();
  const int slices = 256 / C > 0 ? 256 / C : 1;          // C <= 64 here
  const int d = threadIdx.x % C, sl = threadIdx.x / C;
  float acc = 0.f;
  if (sl < slices) {
    for (int b = sl; b < B; b += slices) {
      float t = 0.f;
      for (int c = 0; c < C; ++c) {
        t = fmaf(w1[c], G11[((long)b * C + c) * C + d], t);
        t = fmaf(wd[c], G21[((long)b * C + c) * C + d], t);
      }
      acc = fmaf(Bc[(long)b * Cout + o], t, acc);
      acc = fmaf(Ac[(long)b * Cout + o], ssum[(long)b * C + d], acc);
    }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < C) {
    float tot = M1[(long)o * C + d];
    for (int s2 = 0; s2 < slices; ++s2) tot += part[s2 * C + d];
    const float dwd = M2[(long)o * C + d];
    dW[(long)o * 2 * C + d] = tot - dwd;
    dW[(long)o * 2 * C + C + d] = dwd;
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_route_bwd(const float *dout_pm, const float *ymax, const float *ymin, const uint8_t *amax,
                             const uint8_t *amin, const float *gamma, const float *beta, const float *mean_rstd,
                             const int64_t *idx, int B, int N, int k, int Cout, int G, float slope, float *coef,
                             int64_t *jsel, int64_t *msel, float *dsp, float *dgamma, float *dbeta, double *S,
                             double count_per_group, float *Ac, float *Bc, void *stream) {
  GCN_REQUIRE(dout_pm && ymax && amax && gamma && beta && mean_rstd && coef && dgamma && dbeta && S,
              "gcn_route_bwd: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && Cout >= 1 && G >= 1 && Cout % G == 0, "gcn_route_bwd: bad shape");
  GCN_REQUIRE((Cout <= 256 && 256 % Cout == 0) || Cout % 256 == 0, "gcn_route_bwd: Cout=%d unsupported", Cout);
  GCN_REQUIRE(!(dsp || msel) || idx, "gcn_route_bwd: dsp/msel need idx");
  GCN_REQUIRE((Ac == nullptr) == (Bc == nullptr) && (!Ac || count_per_group > 0), "gcn_route_bwd: Ac/Bc come together, count > 0");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) { GCN_HIP(zero_spans(st, {dgamma, sizeof(float) * Cout}, {dbeta, sizeof(float) * Cout})); return GCN_OK; }
  GCN_HIP(zero_spans(st, {dgamma, sizeof(float) * Cout}, {dbeta, sizeof(float) * Cout}, {S, sizeof(double) * 2 * B * G}));
  if (dsp) GCN_HIP(hipMemsetAsync(dsp, 0, sizeof(float) * (size_t)B * N * Cout, st));
  // (more, shorter blocks are slower: the end-of-block dgamma/dbeta/S atomics all land on the same few lines)
  int blocks = (512 + B - 1) / B;
  int rows = (N + blocks - 1) / blocks;
  if (rows < 8) rows = 8;
  route_bwd_kernel<<<dim3(cdiv(N, rows), B), 256, sizeof(double) * 2 * G + sizeof(float) * 2 * Cout, st>>>(
      dout_pm, ymax, ymin, amax, amin, gamma, beta, mean_rstd, idx, N, k, Cout, G, slope, rows, coef, jsel, msel, dsp,
      dgamma, dbeta, S);
  int rc = check_launch("route_bwd_kernel");
  if (rc || !(Ac && Bc)) return rc;
  gn_affine_kernel<<<cdiv((long)B * Cout, 256), 256, 0, st>>>(mean_rstd, S, B * Cout, Cout, G, count_per_group, Ac, Bc);
  return check_launch("gn_affine_kernel");
}

GCN_EXPORT int gcn_edge_combine(const float *coef, const float *dsp, const float *indeg, const float *Ac, const float *Bc,
                                const float *P1, const float *SW, const float *XW, const float *RW, int B, int N, int k,
                                int Cout, float *D1, float *D2, void *stream) {
  GCN_REQUIRE(coef && dsp && indeg && Ac && Bc && P1 && SW && XW && RW && D1 && D2, "gcn_edge_combine: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && Cout >= 1, "gcn_edge_combine: bad shape");
  if (B == 0) return GCN_OK;
  const long per = (long)N * Cout;
  const int g = (int)((per + 255) / 256 > 1024 ? 1024 : (per + 255) / 256);
  edge_combine_kernel<<<dim3(g, B), 256, 0, (hipStream_t)stream>>>(coef, dsp, indeg, Ac, Bc, P1, SW, XW, RW, N, Cout, (float)k, D1, D2);
  return check_launch("edge_combine_kernel");
}

GCN_EXPORT long gcn_edge_wgrad_ws_floats(int B, int C, int Cout) {
  if (B < 0 || C < 1 || Cout < 1) return -1;
  return 2L * Cout * C + 2L * B * C * C + (long)B * C;
}

GCN_EXPORT int gcn_edge_wgrad(const float *x_pm, const float *s_pm, const float *dsp, const float *d2, const float *indeg,
                              const float *W, const float *Ac, const float *Bc, int B, int N, int C, int Cout, float *dW,
                              float *ws, void *stream) {
  GCN_REQUIRE(x_pm && s_pm && dsp && d2 && indeg && W && Ac && Bc && dW && ws, "gcn_edge_wgrad: null pointer");
  GCN_REQUIRE(B >= 1 && N >= 1, "gcn_edge_wgrad: bad shape");
  GCN_REQUIRE((C <= 16 || C == 64) && (Cout == 64 || Cout == 128),
              "gcn_edge_wgrad: supported C <= 16 or C == 64, Cout in {64,128}; got C=%d Cout=%d", C, Cout);
  hipStream_t st = (hipStream_t)stream;
  const long nws = gcn_edge_wgrad_ws_floats(B, C, Cout);
  GCN_HIP(hipMemsetAsync(ws, 0, sizeof(float) * nws, st));
  float *M1 = ws, *M2 = M1 + (long)Cout * C, *G11 = M2 + (long)Cout * C, *G21 = G11 + (long)B * C * C,
        *ssum = G21 + (long)B * C * C;
  int blocks = (512 + B - 1) / B;
  int rows = (N + blocks - 1) / blocks;
  rows = (rows + 3) & ~3;
  const dim3 grid(cdiv(N, rows), B);
#define GCN_WG(CT, OT) edge_wgrad_kernel<CT, OT><<<grid, 256, 0, st>>>(x_pm, s_pm, dsp, d2, indeg, N, C, Cout, rows, M1, M2, G11, G21, ssum)
  if (C <= 16 && Cout == 64) GCN_WG(1, 4);
  else if (C <= 16) GCN_WG(1, 8);
  else if (Cout == 64) GCN_WG(4, 4);
  else GCN_WG(4, 8);
#undef GCN_WG
  int rc = check_launch("edge_wgrad_kernel");
  if (rc) return rc;
  edge_wgrad_finish_kernel<<<Cout, 256, 0, st>>>(W, Ac, Bc, M1, M2, G11, G21, ssum, B, C, Cout, dW);
  return check_launch("edge_wgrad_finish_kernel");
}
