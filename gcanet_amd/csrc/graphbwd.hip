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

// Destination-partitioned staging of the sparse scatter (see dsp_bucket_sum_kernel)
constexpr int DSP_CAP = 512;                 // entries one route_bwd workgroup may file per partition (mean 256)
struct DspBuckets {
  float *stag_coef;            // (B, P, tiles, DSP_CAP)
  unsigned short *stag_idx;    // same shape: accumulator index (row inside the partition) * Cout + channel  (< 16384)
  int *counts;                 // (B, P, tiles)
  unsigned int *ovf_cnt;       // (B) zeroed by the host
  uint2 *ovf;                  // (B, N*Cout): {row * Cout + channel, coef bits}
  unsigned int *absmax;        // max |coef| bits, zeroed by the host
  int rshift, P;               // rows per partition = 1 << rshift, partitions per cloud
};

// RB_NT threads per workgroup: the loop is a chain of dependent loads (extreme -> neighbour id -> staging slot); with 256
// threads the 512 workgroups of the bench shape left 1.7 waves per SIMD resident, 85 % of their cycles waiting (SQ counters)
constexpr int RB_NT = 1024;
__global__ __launch_bounds__(RB_NT) void route_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ ymax,
                                                        const float *__restrict__ ymin, const unsigned char *__restrict__ amax,
                                                        const unsigned char *__restrict__ amin, const float *__restrict__ gamma,
                                                        const float *__restrict__ beta, const float *__restrict__ mean_rstd,
                                                        const int64_t *__restrict__ idx, int N, int k, int Cout, int G,
                                                        float slope, int rows_per_block, float *__restrict__ coef,
                                                        int64_t *__restrict__ jsel, int64_t *__restrict__ msel,
                                                        float *__restrict__ dsp, float *__restrict__ dgamma,
                                                        float *__restrict__ dbeta, double *__restrict__ S,
                                                        DspBuckets bk, double *__restrict__ part_s,
                                                        float *__restrict__ part_c) {
  extern __shared__ double sm[];             // 2*G doubles, then 2*Cout floats
  __shared__ float amax_s[RB_NT / 64];
  __shared__ int bcnt[64];                   // entries this workgroup has filed per destination partition
  float cfmax = 0.f;                         // max |coef| of this thread (scale of the fixed-point scatter)
  if (bk.stag_coef && threadIdx.x < 64) bcnt[threadIdx.x] = 0;
  float *cs = reinterpret_cast<float *>(sm + 2 * G);
  for (int i = threadIdx.x; i < 2 * G; i += RB_NT) sm[i] = 0.0;
  for (int i = threadIdx.x; i < 2 * Cout; i += RB_NT) cs[i] = 0.f;
  __syncthreads();
  int tile, b;
  xcd_tile_cloud(tile, b);
  const int r0 = tile * rows_per_block, r1 = min(r0 + rows_per_block, N);
  const int cpg = Cout / G;
  // thread -> fixed channel (Cout <= 256 and divides 256, or a multiple of 256)
  const int nct = Cout <= 256 ? Cout : 256;
  const int rstep = RB_NT / nct;
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
        if (bk.stag_coef) {                    // file (accumulator, coef) under the destination's partition;
          const int part = (int)m >> bk.rshift;  // dsp_bucket_sum_kernel adds the entries up afterwards
          const int slot = atomicAdd(&bcnt[part], 1);
          if (slot < DSP_CAP) {
            const long e = (((long)b * bk.P + part) * gridDim.x + tile) * DSP_CAP + slot;
            bk.stag_coef[e] = cf;
            bk.stag_idx[e] = (unsigned short)((((int)m & ((1 << bk.rshift) - 1)) * Cout) | c);
          } else {                              // more than DSP_CAP entries for one partition from this tile (hub graphs)
            const unsigned int oslot = atomicAdd(bk.ovf_cnt + b, 1u);
            bk.ovf[(long)b * N * Cout + oslot] = make_uint2((unsigned int)m * (unsigned int)Cout + (unsigned int)c, __float_as_uint(cf));
          }
          cfmax = fmaxf(cfmax, fabsf(cf));
        } else if (dsp) {
          atomicAdd(dsp + ((long)b * N + m) * Cout + c, cf);
        }
      }
    }
    atomicAdd(&cs[c], dg);
    atomicAdd(&cs[Cout + c], db);
    atomicAdd(&sm[g * 2], (double)s1);
    atomicAdd(&sm[g * 2 + 1], (double)s2);
  }
  if (bk.stag_coef) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cfmax = fmaxf(cfmax, __shfl_xor(cfmax, o));
    if (lane_id() == 0) amax_s[wave_id()] = cfmax;
  }
  __syncthreads();
  if (bk.stag_coef) {
    if (threadIdx.x == 0)                      // non-negative floats order like their bit patterns
    {
      float mx = 0.f;
#pragma unroll
      for (int w = 0; w < RB_NT / 64; ++w) mx = fmaxf(mx, amax_s[w]);
      atomicMax(bk.absmax, __float_as_uint(mx));
    }
    if ((int)threadIdx.x < bk.P) bk.counts[((long)b * bk.P + threadIdx.x) * gridDim.x + tile] = min(bcnt[threadIdx.x], DSP_CAP);
  }
  if (part_s) {
    // per-workgroup partials, folded by fold_partials_kernel (common.h; 512 workgroups adding to the same 2 Cout + 2 G addresses
    // cost ~25 us of contention per launch and made the sums order-dependent)
    const long blk = (long)b * gridDim.x + tile;
    if ((int)threadIdx.x < 2 * G) part_s[blk * 2 * G + threadIdx.x] = sm[threadIdx.x];
    for (int i = threadIdx.x; i < 2 * Cout; i += RB_NT) part_c[blk * 2 * Cout + i] = cs[i];
    return;
  }
  if ((int)threadIdx.x < 2 * G) atomicAdd(S + (long)b * G * 2 + threadIdx.x, sm[threadIdx.x]);
  for (int i = threadIdx.x; i < Cout; i += RB_NT) {
    atomicAdd(dgamma + i, cs[i]);
    atomicAdd(dbeta + i, cs[Cout + i]);
  }
}


// Dsp[m,c] = sum over the points n whose selected neighbour for channel c is m of coef[n,c] -- the sparse part of the
// EdgeConv input gradient.  As 8.4 M scattered global f32 atomics (one per (n,c), every one to a different line) it
// ran at ~28 G atomics/s: 0.27 ms of route_bwd's 0.36 at B=8, N=8192, Cout=128.  Now route_bwd_kernel FILES each
// coefficient under its destination partition (R = 16384/Cout consecutive rows of a cloud): a per-workgroup LDS counter
// per partition hands out the slot inside that workgroup's segment of the partition's staging area (plain stores that
// fill whole lines while the workgroup runs), and this kernel -- one workgroup per (cloud, partition) -- adds up
// exactly its own entries in 64-bit fixed-point LDS accumulators (order-independent: bitwise reproducible) and writes
// every Dsp element once (no zero fill of the (B,N,Cout) buffer).  A segment holds DSP_CAP = 2x the mean; entries
// beyond it (graphs with hubs) go to a per-cloud overflow list that every partition of the cloud filters.
template <int COUT>
__global__ __launch_bounds__(1024) void dsp_bucket_sum_kernel(DspBuckets bk, int B, int N, int tiles, float *__restrict__ dsp) {
  constexpr int R = 16384 / COUT;
  extern __shared__ unsigned long long qacc[];                 // [R][COUT]
  const int b = blockIdx.x % B, part = blockIdx.x / B, m0 = part * R;
  for (int i = threadIdx.x; i < R * COUT; i += 1024) qacc[i] = 0ull;
  const float mx = __uint_as_float(*bk.absmax);
  int ex = 0;
  if (mx > 0.f) (void)frexpf(mx, &ex);                         // mx < 2^ex
  int S = 62 - ex - (32 - __clz(N));                           // at most N addends per accumulator
  S = S < 0 ? 0 : (S > 60 ? 60 : S);
  const double scale = ldexp(1.0, S), inv = ldexp(1.0, -S);
  __syncthreads();
  const int wave = wave_id(), lane = lane_id();
  const long seg0 = ((long)b * bk.P + part) * tiles;
  for (int t = wave; t < tiles; t += 16) {                     // a wave takes whole segments: coalesced reads
    const int cnt = bk.counts[seg0 + t];
    const float *sc = bk.stag_coef + (seg0 + t) * DSP_CAP;
    const unsigned short *si = bk.stag_idx + (seg0 + t) * DSP_CAP;
    for (int i = lane; i < cnt; i += 64)
      atomicAdd(&qacc[si[i]], (unsigned long long)__double2ll_rn((double)sc[i] * scale));
  }
  const unsigned int novf = bk.ovf_cnt[b];
  if (novf) {                                                  // block-uniform, normally zero
    const uint2 *ov = bk.ovf + (long)b * N * COUT;
    for (unsigned int i = threadIdx.x; i < novf; i += 1024) {
      const uint2 e = ov[i];
      const unsigned int row = e.x / COUT - (unsigned int)m0;
      if (row < (unsigned int)R)
        atomicAdd(&qacc[row * COUT + e.x % COUT], (unsigned long long)__double2ll_rn((double)__uint_as_float(e.y) * scale));
    }
  }
  __syncthreads();
  float *ob = dsp + ((long)b * N + m0) * COUT;
  for (int i = threadIdx.x; i < R * COUT; i += 1024) ob[i] = (float)((double)(long long)qacc[i] * inv);
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
    const float dg = D1 ? indeg[(long)b * N + n] : 0.f;
    D2[o] = coef[o] + kf * a + bb * (SW[o] + kf * XW[o]);
    if (D1) D1[o] = dsp[o] + dg * (a + bb * P1[o]) + bb * RW[o];
  }
}

// ------------------------------------------------------------------ fused weight-gradient reductions
// All row-reductions the closed-form EdgeConv backward needs, in ONE pass over the point rows, on the f32
// matrix cores (v_mfma_f32_16x16x4_f32: A[i][k] and B[k][j] are both "row n0+k, column 16t+(lane&15)" reads of
// the row-major operands, so no transposes are needed for X^T.Y products):
//   M1 += Dsp^T x, M2 += D2^T x  (Cout x C, summed over clouds);  G11[b] = x^T diag(indeg) x,  G21[b] = x^T s
//   (C x C per cloud);  ssum[b] = sum_n s[n,:].
// Workgroup = 4 waves on one row chunk of one cloud; wave w owns output-row tiles {w*OW..} of M1/M2 and
// c-tile w of G11/G21; every workgroup stores its partial tiles, edge_wgrad_fold_kernel adds them.
typedef __attribute__((ext_vector_type(4))) float wg_f32x4;

template <int CT, int OT>
__global__ __launch_bounds__(256) void edge_wgrad_kernel(const float *__restrict__ x, const float *__restrict__ sx,
                                                         const float *__restrict__ dsp, const float *__restrict__ d2,
                                                         const float *__restrict__ indeg, int N, int C, int Cout,
                                                         int rows_per_block, float *__restrict__ part) {
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
  // This workgroup's partial [M1 | M2 | G11 | G21 | ssum], every element written once; edge_wgrad_fold_kernel adds the
  // partials in a fixed order.  (As float atomics these were 12.6 M device atomics per launch at C = 64, Cout = 128 --
  // 49 MB of write traffic, 512 workgroups queueing on the same addresses at the end of the kernel.)
  float *M1 = part + ((long)b * gridDim.x + tile) * (2L * Cout * C + 2L * C * C + C);
  float *M2 = M1 + (long)Cout * C, *G11 = M2 + (long)Cout * C, *G21 = G11 + (long)C * C, *ssum = G21 + (long)C * C;
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
          M1[(long)oc * C + c] = m1[o][t][r];
          M2[(long)oc * C + c] = m2[o][t][r];
        }
      if (gwave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cr = 16 * wave + 4 * lk + r;
          if (cr < C) {
            G11[(long)cr * C + c] = g11[t][r];
            G21[(long)cr * C + c] = g21[t][r];
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
      if (lk == 0 && c < C) ssum[c] = v;
    }
  }
}

// fin = [M1 | M2 (Cout,C) | G11 | G21 (B,C,C) | ssum (B,C)] from the workgroups' partials: M1, M2 over all B * nblk of
// them, the rest over the nblk of their cloud.  64 outputs per workgroup, each wave a quarter of the partials, 16 loads
// in flight; fixed order.
__global__ __launch_bounds__(256) void edge_wgrad_fold_kernel(const float *__restrict__ part, int B, int nblk, int C, int Cout,
                                                              float *__restrict__ fin) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long CC = (long)C * C, n_m = 2L * Cout * C, n_g = 2 * CC + C, PSZ = n_m + n_g;
  const long NO = n_m + (long)B * n_g;
  const long i = (long)blockIdx.x * 64 + lane;
  long off = 0, dst = 0;
  int p0 = 0, np = 0;
  if (i < n_m) {
    off = i; np = B * nblk; dst = i;
  } else if (i < NO) {
    const long j = i - n_m;
    const int b = (int)(j / n_g);
    const long e = j % n_g;
    off = n_m + e; p0 = b * nblk; np = nblk;
    dst = e < CC ? n_m + b * CC + e : (e < 2 * CC ? n_m + B * CC + b * CC + (e - CC) : n_m + 2 * B * CC + (long)b * C + (e - 2 * CC));
  }
  const int k0 = p0 + (int)((long)np * wave / 4), k1 = p0 + (int)((long)np * (wave + 1) / 4);
  float acc = 0.f;
  int k = k0;
  for (; k + 16 <= k1; k += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(long)(k + u) * PSZ + off];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  for (; k < k1; ++k) acc += part[(long)k * PSZ + off];
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < NO) fin[dst] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
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

struct DspWs { size_t counts, stag_idx, stag_coef, ovf, total; };
static DspWs dsp_ws_layout(int B, int N, int Cout, int tiles) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t P = (size_t)N / (16384 / Cout);
  DspWs w{};
  size_t o = 256;                                               // zeroed header
  w.counts = o; o += al(sizeof(int) * (size_t)B * P * tiles);
  w.stag_idx = o; o += al(sizeof(unsigned short) * (size_t)B * P * tiles * DSP_CAP);
  w.stag_coef = o; o += al(sizeof(float) * (size_t)B * P * tiles * DSP_CAP);
  w.ovf = o; o += al(sizeof(uint2) * (size_t)B * N * Cout);
  w.total = o;
  return w;
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_route_bwd_part_bytes(int B, int N, int Cout, int G) {
  if (B < 0 || N < 1 || Cout < 1 || G < 1) return -1;
  int blocks = (512 + (B > 0 ? B : 1) - 1) / (B > 0 ? B : 1);
  int rows = (N + blocks - 1) / blocks;
  if (rows < 8) rows = 8;
  const long tiles = cdiv(N, rows);
  return 8L * 2 * G * B * tiles + 4L * 2 * Cout * B * tiles;
}

GCN_EXPORT long gcn_route_bwd_ws_bytes(int B, int N, int Cout) {
  if (B < 0 || N < 1 || Cout < 1) return -1;
  if (!(Cout == 64 || Cout == 128) || N % (16384 / Cout) != 0) return 256;
  int blocks = (512 + B - 1) / B;
  int rows = (N + blocks - 1) / blocks;
  if (rows < 8) rows = 8;
  return (long)dsp_ws_layout(B, N, Cout, cdiv(N, rows)).total;
}

GCN_EXPORT int gcn_route_bwd(const float *dout_pm, const float *ymax, const float *ymin, const uint8_t *amax,
                             const uint8_t *amin, const float *gamma, const float *beta, const float *mean_rstd,
                             const int64_t *idx, int B, int N, int k, int Cout, int G, float slope, float *coef,
                             int64_t *jsel, int64_t *msel, float *dsp, float *dgamma, float *dbeta, double *S,
                             double count_per_group, float *Ac, float *Bc, void *dsp_ws, void *part_ws, void *stream) {
  GCN_REQUIRE(dout_pm && ymax && amax && gamma && beta && mean_rstd && coef && dgamma && dbeta && S,
              "gcn_route_bwd: null pointer");
  GCN_REQUIRE(B >= 0 && N >= 1 && Cout >= 1 && G >= 1 && Cout % G == 0, "gcn_route_bwd: bad shape");
  GCN_REQUIRE((Cout <= 256 && 256 % Cout == 0) || Cout % 256 == 0, "gcn_route_bwd: Cout=%d unsupported", Cout);
  GCN_REQUIRE(!(dsp || msel) || idx, "gcn_route_bwd: dsp/msel need idx");
  GCN_REQUIRE((Ac == nullptr) == (Bc == nullptr) && (!Ac || count_per_group > 0), "gcn_route_bwd: Ac/Bc come together, count > 0");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) { GCN_HIP(zero_spans(st, {dgamma, sizeof(float) * Cout}, {dbeta, sizeof(float) * Cout})); return GCN_OK; }
  // dsp_ws (gcn_route_bwd_ws_bytes): zeroed header [0,4) max |coef| bits, [16, 16+4B) overflow counts; then the
  // per-(cloud, partition, tile) counts, the staging area and the overflow list -- the partitioned LDS sum
  // (dsp_bucket_sum_kernel); without it, or for other shapes, the coefficients are scattered with f32 atomics
  // (more, shorter blocks are slower: the end-of-block dgamma/dbeta/S atomics all land on the same few lines)
  int blocks = (512 + B - 1) / B;
  int rows = (N + blocks - 1) / blocks;
  if (rows < 8) rows = 8;
  const int tiles = cdiv(N, rows);
  const bool lds_scatter = dsp && dsp_ws && (Cout == 64 || Cout == 128) && N <= 65536 && N % (16384 / Cout) == 0 &&
                           N / (16384 / Cout) <= 64 && B <= 60 && ((uintptr_t)dsp_ws & 15) == 0;
  DspBuckets bk{};
  if (lds_scatter) {
    const DspWs w = dsp_ws_layout(B, N, Cout, tiles);
    char *base = (char *)dsp_ws;
    bk.absmax = (unsigned int *)base;
    bk.ovf_cnt = (unsigned int *)(base + 16);
    bk.counts = (int *)(base + w.counts);
    bk.stag_idx = (unsigned short *)(base + w.stag_idx);
    bk.stag_coef = (float *)(base + w.stag_coef);
    bk.ovf = (uint2 *)(base + w.ovf);
    bk.P = N / (16384 / Cout);
    bk.rshift = Cout == 64 ? 8 : 7;
  }
  double *part_s = nullptr;
  float *part_c = nullptr;
  if (part_ws) {                              // gcn_route_bwd_part_bytes: the sums are written, nothing to zero
    GCN_REQUIRE(((uintptr_t)part_ws & 7) == 0, "gcn_route_bwd: part_ws must be 8-byte aligned");
    part_s = (double *)part_ws;
    part_c = (float *)(part_s + 2L * G * B * tiles);
    GCN_HIP(zero_spans(st, {lds_scatter ? dsp_ws : nullptr, lds_scatter ? 256u : 0u}));
  } else {
    GCN_HIP(zero_spans(st, {dgamma, sizeof(float) * Cout}, {dbeta, sizeof(float) * Cout}, {S, sizeof(double) * 2 * B * G},
                       {lds_scatter ? dsp_ws : nullptr, lds_scatter ? 256u : 0u}));
  }
  if (dsp && !lds_scatter) GCN_HIP(zero_dev(dsp, sizeof(float) * (size_t)B * N * Cout, st));
  route_bwd_kernel<<<dim3(cdiv(N, rows), B), RB_NT, sizeof(double) * 2 * G + sizeof(float) * 2 * Cout, st>>>(
      dout_pm, ymax, ymin, amax, amin, gamma, beta, mean_rstd, idx, N, k, Cout, G, slope, rows, coef, jsel, msel, dsp,
      dgamma, dbeta, S, bk, part_s, part_c);
  int rc = check_launch("route_bwd_kernel");
  if (!rc && part_ws) {
    fold_partials_kernel<<<cdiv(2 * Cout, 64) + cdiv(B * 2 * G, 64), 256, 0, st>>>(part_s, part_c, tiles, B, Cout, G, S, dgamma, dbeta);
    rc = check_launch("fold_partials_kernel");
  }
  if (!rc && lds_scatter) {
    const int ldsb = 16384 * 8;
    if (Cout == 64) {
      GCN_HIP(hipFuncSetAttribute((const void *)dsp_bucket_sum_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
      dsp_bucket_sum_kernel<64><<<bk.P * B, 1024, ldsb, st>>>(bk, B, N, tiles, dsp);
    } else {
      GCN_HIP(hipFuncSetAttribute((const void *)dsp_bucket_sum_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
      dsp_bucket_sum_kernel<128><<<bk.P * B, 1024, ldsb, st>>>(bk, B, N, tiles, dsp);
    }
    rc = check_launch("dsp_bucket_sum_kernel");
  }
  if (rc || !(Ac && Bc)) return rc;
  gn_affine_kernel<<<cdiv((long)B * Cout, 256), 256, 0, st>>>(mean_rstd, S, B * Cout, Cout, G, count_per_group, Ac, Bc);
  return check_launch("gn_affine_kernel");
}

GCN_EXPORT int gcn_edge_combine(const float *coef, const float *dsp, const float *indeg, const float *Ac, const float *Bc,
                                const float *P1, const float *SW, const float *XW, const float *RW, int B, int N, int k,
                                int Cout, float *D1, float *D2, void *stream) {
  GCN_REQUIRE(coef && Ac && Bc && SW && XW && D2, "gcn_edge_combine: null pointer");
  GCN_REQUIRE(!D1 || (dsp && indeg && P1 && RW), "gcn_edge_combine: D1 needs dsp, indeg, P1 and RW");
  GCN_REQUIRE(B >= 0 && N >= 1 && Cout >= 1, "gcn_edge_combine: bad shape");
  if (B == 0) return GCN_OK;
  const long per = (long)N * Cout;
  const int g = (int)((per + 255) / 256 > 1024 ? 1024 : (per + 255) / 256);
  edge_combine_kernel<<<dim3(g, B), 256, 0, (hipStream_t)stream>>>(coef, dsp, indeg, Ac, Bc, P1, SW, XW, RW, N, Cout, (float)k, D1, D2);
  return check_launch("edge_combine_kernel");
}

GCN_EXPORT long gcn_edge_wgrad_ws_floats(int B, int C, int Cout) {
  if (B < 0 || C < 1 || Cout < 1) return -1;
  // the five result arrays + one partial set per workgroup (at most ceil(512 / B) workgroups per cloud)
  const long psz = 2L * Cout * C + 2L * C * C + C;
  return 2L * Cout * C + 2L * B * C * C + (long)B * C + (long)B * ((512 + B - 1) / B) * psz;
}

GCN_EXPORT int gcn_edge_wgrad(const float *x_pm, const float *s_pm, const float *dsp, const float *d2, const float *indeg,
                              const float *W, const float *Ac, const float *Bc, int B, int N, int C, int Cout, float *dW,
                              float *ws, void *stream) {
  GCN_REQUIRE(x_pm && s_pm && dsp && d2 && indeg && W && Ac && Bc && dW && ws, "gcn_edge_wgrad: null pointer");
  GCN_REQUIRE(B >= 1 && N >= 1, "gcn_edge_wgrad: bad shape");
  GCN_REQUIRE((C <= 16 || C == 64) && (Cout == 64 || Cout == 128),
              "gcn_edge_wgrad: supported C <= 16 or C == 64, Cout in {64,128}; got C=%d Cout=%d", C, Cout);
  hipStream_t st = (hipStream_t)stream;
  float *M1 = ws, *M2 = M1 + (long)Cout * C, *G11 = M2 + (long)Cout * C, *G21 = G11 + (long)B * C * C,
        *ssum = G21 + (long)B * C * C, *part = ssum + (long)B * C;
  int blocks = (512 + B - 1) / B;
  int rows = (N + blocks - 1) / blocks;
  rows = (rows + 3) & ~3;
  const dim3 grid(cdiv(N, rows), B);           // grid.x <= blocks: the partial sets fit gcn_edge_wgrad_ws_floats
#define GCN_WG(CT, OT) edge_wgrad_kernel<CT, OT><<<grid, 256, 0, st>>>(x_pm, s_pm, dsp, d2, indeg, N, C, Cout, rows, part)
  if (C <= 16 && Cout == 64) GCN_WG(1, 4);
  else if (C <= 16) GCN_WG(1, 8);
  else if (Cout == 64) GCN_WG(4, 4);
  else GCN_WG(4, 8);
#undef GCN_WG
  int rc = check_launch("edge_wgrad_kernel");
  if (rc) return rc;
  {
    const long NO = 2L * Cout * C + (long)B * (2L * C * C + C);
    edge_wgrad_fold_kernel<<<(int)((NO + 63) / 64), 256, 0, st>>>(part, B, (int)grid.x, C, Cout, ws);
    rc = check_launch("edge_wgrad_fold_kernel");
    if (rc) return rc;
  }
  edge_wgrad_finish_kernel<<<Cout, 256, 0, st>>>(W, Ac, Bc, M1, M2, G11, G21, ssum, B, C, Cout, dW);
  return check_launch("edge_wgrad_finish_kernel");
}
