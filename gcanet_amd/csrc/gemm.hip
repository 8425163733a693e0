// gemm.hip -- the per-point 1x1 convolutions of GCANet's heads (Conv1d(kernel 1) on (B,C,N) tensors,
// models/dgcnn-hais-concat-direct-4.py:556-603,644-699,713: 1280->512->256, 256->256->{10,22}, 832->256->64, 262->128,
// 256->1024) as bf16 MFMA GEMMs on POINT-major activations, for gfx950.
//
//   out[m][n] = sum_k A[m][k] * W[n][k] (+ bias[n])      A (M,K) bf16 rows = points, W (N,K) bf16 = the Conv1d weight
//
// Both operands are contiguous along the contraction index, which is exactly the MFMA fragment shape (a lane holds 8
// consecutive k of one row), so tiles go HBM -> LDS by LDS-DMA with no transposition: [rows][64 k] bf16 images,
// 16-byte chunks XOR-swizzled on the source side (conflict-free ds_read_b128), double buffered.  The same kernel is
// the input gradient (dX = dY . W: pass dY and W^T) -- and, with CONTRACT_ROWS, the weight gradient
// dW[n][k] = sum_m dY[m][n] X[m][k], whose contraction index runs down the ROWS of both row-major operands: there the
// fragments come from the same row-major LDS images through the hardware transpose read ds_read_b64_tr_b16.
//
// Epilogue (forward): bias, optional GroupNorm statistics -- per (cloud, group) sum and sum of squares of the f32
// results, so the separate statistics pass over the output disappears -- and bf16 or f32 stores.  The product is
// formed as D[n][m] (W on the MFMA row index) so that a lane owns ONE output row and four consecutive columns per
// accumulator group: 8- or 16-byte stores instead of 2-byte ones.
#include "common.h"

#include <algorithm>

#include <cstdlib>
#include <type_traits>

namespace gcn {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmArgs {
  const unsigned short *A;   // (M,K)
  const unsigned short *W;   // (Np,K), Np = N rounded up to 32 (rows beyond N are zero)
  const float *bias;         // (N) or null
  void *out;                 // (M,N) bf16 or f32
  double *gsum;              // (M/rows_per_cloud, G, 2) or null
  double *part;              // (M/32, N/32, 2) per-wave partial sums when gsum is wanted
  long M;
  int N, Np, K, out_f32, rows_per_cloud, G;
};

__device__ __forceinline__ unsigned short gemm_f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

// BN = columns per workgroup (32, 64 or 128); workgroup = 128 rows, 4 waves.  BN = 128: waves 2 x 2, a wave owns
// 64 rows x 64 columns (each LDS fragment feeds two MFMAs: with 32 x 128 per wave the kernel was bound by the LDS
// reads of the four-times re-read W fragments); BN <= 64: waves 4 x 1, 32 rows x BN columns.
// K is walked in steps of 32 through a ring of four LDS stages with THREE stages in flight (the layers are only
// 4-26 steps deep: a two-buffer scheme pays a full memory latency per step).  Counted s_waitcnt vmcnt + raw
// s_barrier: __syncthreads() would drain the ring.
// BM = 256, BN = 256 (one workgroup per CU, 256 accumulator registers per lane, opt-in): a wave owns 128 x 128, so every
// LDS fragment feeds FOUR MFMAs instead of two.
// NWV = 8 (BM = BN = 256, round 3): eight waves in a 2 x 4 grid, a wave owns 128 rows x 64 columns = 128 accumulator
// registers -- no AGPR traffic, TWO waves per SIMD (one wave's LDS / DMA waits hide under the other's MFMAs, which the
// one-wave-per-SIMD 128 x 128 variant could not do) and 12 KB of LDS fragments per 16 MFMAs and wave.
template <int BN, int BM = 128, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, (BM == 256 || NWV == 8) ? 1 : 2) void gemm_bf16_kernel(GemmArgs a) {
  constexpr bool SQ = BN >= 128;                // 2 x 2 waves (NWV = 4)
  constexpr int WR = NWV == 8 ? 2 : (SQ ? 2 : 4);   // wave grid: WR rows x WCN columns
  constexpr int WCN = NWV / WR;
  constexpr int RB = BM / WR / 32;              // 32-row blocks per wave
  constexpr int CB = BN / WCN / 32;             // 32-column blocks per wave
  constexpr int WCOLS = CB * 32;                // columns per wave
  constexpr int NST = 4;
  constexpr int A_BYTES = BM * 64;              // BM rows x 32 k x 2 B
  constexpr int B_BYTES = BN * 64;
  constexpr int ST_BYTES = A_BYTES + B_BYTES;
  constexpr int APW = BM / 16 / NWV;            // A pieces (16 rows x 64 B) per wave and stage
  constexpr int WPW = BN >= 16 * NWV ? BN / 16 / NWV : 1;   // W pieces per wave and stage; BN = 32: waves 2,3 repeat 0,1
  constexpr int DPS = APW + WPW;                // DMA instructions per wave and stage (the vmcnt unit)
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = lane_id(), wave = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  const int wrow0 = (wave / WCN) * (BM / WR);   // this wave's first row / column inside the tile
  const int wcol0 = (wave % WCN) * (BN / WCN);
  // Workgroups are dealt to the 8 XCDs round robin by linear id.  The column blocks of one 128-row tile all read the
  // same A rows: give them to ONE XCD, back to back, so that the tile comes from HBM once and from that XCD's L2 for
  // the other column blocks.
  const int ncb = (a.N + BN - 1) / BN;
  const long nrt = (a.M + BM - 1) / BM;
  const long id = blockIdx.x;
  long rt;
  int cbk;
  if (nrt % 8 == 0) {
    const long j = id >> 3;
    cbk = (int)(j % ncb);
    rt = (j / ncb) * 8 + (id & 7);
  } else {
    cbk = (int)(id % ncb);
    rt = id / ncb;
  }
  const long m0 = rt * BM;
  const int n0 = cbk * BN;
  const int K = a.K;
  const int ktiles = (K + 31) / 32;

  // DMA pieces: 1 KiB = 16 rows x 64 B (4 chunks of 16 B); chunk swizzle (row/4)&3 on the source side
  const int prow = lane >> 2, cs = lane & 3;
  const unsigned short *arow[APW];
#pragma unroll
  for (int i = 0; i < APW; ++i) {
    const int row = (wave + i * NWV) * 16 + prow;
    long m = m0 + row;
    if (m >= a.M) m = a.M - 1;
    arow[i] = a.A + m * K + ((cs ^ ((row >> 2) & 3)) * 8);
  }
  const unsigned short *wrow[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int p = BN >= 16 * NWV ? wave + i * NWV : (wave & 1);
    const int row = p * 16 + prow;
    wrow[i] = a.W + (long)min(n0 + row, a.Np - 1) * K + ((cs ^ ((row >> 2) & 3)) * 8);
  }
  const int cl = cs ^ ((prow >> 2) & 3);                     // this lane's logical chunk (row/4 & 3 == prow/4 & 3)
  auto issue = [&](int kt, int stg) {
    const int kbase = kt * 32;
    const bool ok = kbase + cl * 8 < K;                       // K % 16 == 0: the last step may hold two chunks only
    unsigned char *dst = lds + stg * ST_BYTES;
#pragma unroll
    for (int i = 0; i < APW; ++i)
      if (ok)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(arow[i] + kbase),
                                         (__attribute__((address_space(3))) void *)(dst + (wave + i * NWV) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (ok)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wrow[i] + kbase),
                                         (__attribute__((address_space(3))) void *)(dst + A_BYTES + (BN >= 16 * NWV ? wave + i * NWV : (wave & 1)) * 1024), 16, 0, 0);
  };
  // fragment byte offsets inside an image: row lr (of a 32-row block), k-step s (0,1), half lh; 64-B rows
  unsigned int frag[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) frag[s] = (unsigned int)(lr * 64 + (((2 * s + lh) ^ ((lr >> 2) & 3)) << 4));

  f32x16 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[rb][cb][i] = 0.f;

#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < ktiles) issue(p, p);

  // Software pipeline (round 3).  The fragments of k-step s+1 are read from LDS while the MFMAs of k-step s issue, and the
  // workgroup barrier sits in the MIDDLE of a k-tile: [frags(kt, 0) in registers] -> read frags(kt, 1) | MFMAs(kt, 0)
  // -> stage kt+1 landed? barrier -> refill the ring -> read frags(kt+1, 0) | MFMAs(kt, 1).  With the barrier at the top
  // of the tile (rounds 2) all waves read their fragments at the same time and only then multiplied: PMC showed the
  // matrix pipe 31 % busy at two waves per SIMD, waves waiting 40 % of their cycles -- LDS and MFMA phases in lock step.
  auto wait_landed = [&](int later) {                          // my share of a stage has landed, `later` younger stages may fly
    static_assert(DPS == 3 || DPS == 4 || DPS == 8, "vmcnt literals below");
    if (later >= 2) {
      if (DPS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (DPS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (later == 1) {
      if (DPS == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (DPS == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  bf16x8 xf[2][RB], wf[2][CB];
  auto load_frags = [&](int stg, int s, int set) {
    const unsigned char *ta = lds + stg * ST_BYTES + wrow0 * 64;
    const unsigned char *tw = lds + stg * ST_BYTES + A_BYTES + wcol0 * 64;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xf[set][rb] = *reinterpret_cast<const bf16x8 *>(ta + frag[s] + rb * 32 * 64);
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) wf[set][cb] = *reinterpret_cast<const bf16x8 *>(tw + frag[s] + cb * 32 * 64);
  };
  auto mfmas = [&](int set) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb)
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[set][cb], xf[set][rb], acc[rb][cb], 0, 0, 0);   // D[n][m]
  };
  // prologue: stage 0 landed everywhere, its first k-step in registers
  wait_landed(min(NST - 2, ktiles - 1));
  __builtin_amdgcn_s_barrier();
  load_frags(0, 0, 0);
  auto ktile = [&](int kt, auto stgc) {
    constexpr int STG = decltype(stgc)::value;
    const int ksteps = min(2, (K - kt * 32) >> 4);             // wave-uniform
    if (ksteps == 2) load_frags(STG, 1, 1);
    mfmas(0);
    if (kt + 1 < ktiles) {
      // stage kt+1: tiles kt+2 .. kt+NST-2 were issued after it
      wait_landed(min(NST - 3, ktiles - 2 - kt));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my LDS reads of stage kt are done: the stage may be refilled
      __builtin_amdgcn_s_barrier();                            // stage kt+1 is in for everybody; stages <= kt are free
      if (kt + NST - 1 < ktiles) issue(kt + NST - 1, (STG + NST - 1) % NST);
      load_frags((STG + 1) % NST, 0, 0);
    }
    if (ksteps == 2) mfmas(1);
  };
  for (int kt = 0; kt < ktiles; kt += 4) {
    ktile(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < ktiles) ktile(kt + 1, std::integral_constant<int, 1>{});
    if (kt + 2 < ktiles) ktile(kt + 2, std::integral_constant<int, 2>{});
    if (kt + 3 < ktiles) ktile(kt + 3, std::integral_constant<int, 3>{});
  }

  // ---- epilogue.  Block (rb, cb): lane = output row m0 + wrow0 + rb*32 + lr; register i = column
  // n0 + wcol0 + cb*32 + 8 (i>>2) + 4 lh + (i&3).  A lane's values are spread over its row in 8-byte pieces: stored
  // straight from the registers, every store instruction would touch 32 rows (the 256->512 layer spent more time
  // storing than multiplying).  The wave's tile goes through LDS instead (the stage ring is free now) and leaves as
  // whole 16-byte chunks of consecutive rows: 1 KiB per store instruction.
  __builtin_amdgcn_s_barrier();                               // every wave is done with the ring
  const int esz = a.out_f32 ? 4 : 2;
  const int rstride = WCOLS * esz + 16;                       // + 16 B: the column-wise writes below spread over the banks
  unsigned char *tile = lds + wave * (32 * (WCOLS * 4 + 16));   // one 32-row slab of the wave's tile at a time
  const bool vec_ok = (a.N % (16 / esz)) == 0;                // rows are 16-byte aligned and chunks never straddle N
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const long mrow = m0 + wrow0 + rb * 32 + lr;
    const bool mv = mrow < a.M;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      float bs1 = 0.f, bs2 = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = cb * 32 + 8 * g + 4 * lh;              // column inside the wave's tile
        const int n = n0 + wcol0 + nl;
        float4 v = {acc[rb][cb][4 * g], acc[rb][cb][4 * g + 1], acc[rb][cb][4 * g + 2], acc[rb][cb][4 * g + 3]};
        if (a.bias) {
          if (n + 3 < a.N && (a.N & 3) == 0) {
            const float4 bq = *reinterpret_cast<const float4 *>(a.bias + n);
            v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
          } else {
            v.x += n + 0 < a.N ? a.bias[n + 0] : 0.f;
            v.y += n + 1 < a.N ? a.bias[n + 1] : 0.f;
            v.z += n + 2 < a.N ? a.bias[n + 2] : 0.f;
            v.w += n + 3 < a.N ? a.bias[n + 3] : 0.f;
          }
        }
        if (a.gsum && mv) {
          bs1 += (v.x + v.y) + (v.z + v.w);
          bs2 = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, bs2))));
        }
        if (vec_ok) {
          if (a.out_f32) {
            *reinterpret_cast<float4 *>(tile + lr * rstride + nl * 4) = v;
          } else {
            bf16x4 h = {(short)gemm_f2bf(v.x), (short)gemm_f2bf(v.y), (short)gemm_f2bf(v.z), (short)gemm_f2bf(v.w)};
            *reinterpret_cast<bf16x4 *>(tile + lr * rstride + nl * 2) = h;
          }
        } else if (mv) {                                      // narrow odd-width heads (10, 22, 3 columns): element stores
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < a.N) {
              if (a.out_f32) reinterpret_cast<float *>(a.out)[mrow * a.N + n + e] = vv[e];
              else reinterpret_cast<unsigned short *>(a.out)[mrow * a.N + n + e] = gemm_f2bf(vv[e]);
            }
        }
      }
      if (a.gsum) {
        // one partial per (32-row slab, 32-column block): f64 atomics on the (cloud, group) sums serialise at
        // ~0.45 us per same-address add (2048 workgroups x 16: 100 us); gemm_stats_reduce_kernel folds the partials
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          bs1 += __shfl_xor(bs1, o);
          bs2 += __shfl_xor(bs2, o);
        }
        if (lane == 0 && n0 + wcol0 + cb * 32 < a.N) {
          const long slab = (m0 + wrow0) / 32 + rb;
          double *p = a.part + (slab * (a.N / 32) + ((n0 + wcol0) / 32 + cb)) * 2;
          p[0] = (double)bs1;
          p[1] = (double)bs2;
        }
      }
    }
    if (vec_ok) {
      __builtin_amdgcn_wave_barrier();                        // same wave wrote the slab: in-order LDS queue
      const int cpr = WCOLS * esz / 16;                       // 16-byte chunks per tile row (4 ... 32)
      const int rpi = 64 / cpr;                               // rows per store instruction
      for (int r0 = 0; r0 < 32; r0 += rpi) {
        const int r = r0 + lane / cpr, ch = lane % cpr;
        const int ncol = n0 + wcol0 + ch * (16 / esz);
        const long mr = m0 + wrow0 + rb * 32 + r;
        if (mr < a.M && ncol < a.N) {
          const uint4 d = *reinterpret_cast<const uint4 *>(tile + r * rstride + ch * 16);
          *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(a.out) + (mr * a.N + ncol) * esz) = d;
        }
      }
      __builtin_amdgcn_wave_barrier();                        // the next slab overwrites the tile after these reads
    }
  }
}

// gsum[cloud][group][2] = sum of the partials of the cloud's row slabs and the group's column blocks
__global__ __launch_bounds__(256) void gemm_stats_reduce_kernel(const double *__restrict__ part, double *__restrict__ gsum,
                                                                int slabs_per_cloud, int nb32, int G) {
  __shared__ double red[2][4];
  const int cloud = blockIdx.x / G, grp = blockIdx.x % G;
  const int bpg = nb32 / G;                                   // column blocks per group
  const int total = slabs_per_cloud * bpg;
  double s1 = 0.0, s2 = 0.0;
  for (int e = threadIdx.x; e < total; e += 256) {
    const int slab = e / bpg, cb = e % bpg;
    const double *p = part + (((long)cloud * slabs_per_cloud + slab) * nb32 + grp * bpg + cb) * 2;
    s1 += p[0];
    s2 += p[1];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane_id() == 0) { red[0][wave_id()] = s1; red[1][wave_id()] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    gsum[(long)blockIdx.x * 2] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    gsum[(long)blockIdx.x * 2 + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// ------------------------------------------------------------------ weight gradient: contraction over the rows
// dW[n][k] = sum_m dY[m][n] X[m][k]   (dY (M,N) bf16, X (M,K) bf16, both row-major; dW (N,K) f32).  M is cut into row
// slices; every (tile, slice) workgroup stores its partial tile and gemm_wgrad_fold_kernel adds the slices in a fixed
// order (float atomics onto dW instead: 128 workgroups queueing on the same 16 K addresses made a 45-us floor under
// the narrow layers, and the sum order changed from run to run).  Workgroup tile = 128 (n) x 128 (k); it walks its slice of M in steps of 64 rows:
// the two [64 m][128] images are fetched by LDS-DMA as they lie in memory and read COLUMN-wise with
// ds_read_b64_tr_b16 (per 16 lanes: a 4-row x 16-column block delivered column-major, i.e. 4 consecutive m per lane).
struct WgradArgs {
  const unsigned short *dY;  // (M,N)
  const unsigned short *X;   // (M,K)
  float *dW;                 // (N,K) f32
  float *db;                 // (N) f32 column sums of dY (the bias gradient); may be null
  float *part;               // (slices, N*K + N) f32 partial results of the row slices
  long M, mper;              // rows, rows per slice (a multiple of 32)
  int N, K, msplit;
};

typedef __attribute__((ext_vector_type(4))) short tr_v4i16;
__device__ __forceinline__ bf16x4 lds_tr_read(unsigned int addr) {
  const tr_v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_v4i16 *)(uintptr_t)addr);
  return bf16x4{r[0], r[1], r[2], r[3]};
}

__global__ __launch_bounds__(256, 2) void gemm_wgrad_bf16_kernel(WgradArgs a) {
  // images: [32 rows m][128 cols] bf16 = 256-B rows, 8 KB each; a stage = (dY, X) = 16 KB; ring of four stages.
  // Three stages are in flight while one is consumed: a 64-row double buffer left every step waiting for the whole
  // fetch latency (512 MFMA cycles of work per wave against ~1 us), the matrix pipe was busy 16 % of the time.
  constexpr int IMG = 32 * 256, STAGE = 2 * IMG, NSTG = 4;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = lane_id(), wave = wave_id();
  // workgroup id -> (tile, row slice) with ALL tiles of a row slice on one XCD (ids are dealt round-robin to the 8
  // XCDs): they run concurrently and walk the slice in step, so each [32 x 128] operand image comes from HBM once and
  // from that XCD's L2 for the other tiles.  (Tiles of one slice spread over the XCDs re-read dY K/128 times and X
  // N/128 times from memory: 1.3 GB instead of 0.24 GB at N=512, K=1280.)
  const int tiles_n = (a.N + 127) / 128, tiles = tiles_n * ((a.K + 127) / 128);
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int tile = jj % tiles, zslice = (jj / tiles) * 8 + xcd;
  if (zslice >= a.msplit) return;
  const int tile_k = tile / tiles_n;
  const int n0 = (tile % tiles_n) * 128, k0 = tile_k * 128;
  const long mper = a.mper;
  const long mb = (long)zslice * mper;
  const long me = mb + mper < a.M ? mb + mper : a.M;
  if (mb >= me) return;
  const int steps = (int)((me - mb + 31) / 32);

  // DMA: 1 KiB = 4 rows x 256 B: 8 pieces per image; wave w fetches pieces w and w + 4 of both images (4 requests per
  // stage and wave: the s_waitcnt arithmetic below relies on it); lane -> row p*4 + lane/16, chunk lane%16 (no
  // swizzle: the transposed reads take 4-row x 16-column blocks).  Stages past the end of the slice fetch its last row
  // again (never read), so that every iteration issues the same number of requests.
  const int ch = lane & 15;
  // columns beyond N / K and rows beyond the slice: clamped reads whose products are masked out at the end
  const int ncol = min(n0 + ch * 8, a.N - 8), kcol = min(k0 + ch * 8, a.K - 8);
  auto issue = [&](int st) {
    const long mrow0 = mb + (long)st * 32;
    unsigned char *base = lds + (st & (NSTG - 1)) * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave + i * 4;
      long m = mrow0 + p * 4 + (lane >> 4);
      if (m > me - 1) m = me - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.dY + m * a.N + ncol),
                                       (__attribute__((address_space(3))) void *)(base + p * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.X + m * a.K + kcol),
                                       (__attribute__((address_space(3))) void *)(base + IMG + p * 1024), 16, 0, 0);
    }
  };
  // wave tile: 64 (n) x 64 (k): waves 2 x 2; MFMA D[n][k]: A = dY^T fragment (row n, 8 consecutive m), B = X^T fragment
  const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
  // transposed read: group g = lane/16 of 16 lanes reads the block rows [8*(g/2).. +4 (+4 for the second read)] x
  // columns [16*(g%2) .. +16) of a 32-column operand block; lane i = 4q+p of the group supplies row q, columns 4p..4p+3
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const unsigned int tr_base = (unsigned int)((8 * (g >> 1) + q) * 256 + (16 * (g & 1) + 4 * pp) * 2);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // bias gradient: the k-tile-0 workgroups also contract their dY fragments with a fragment of ones (every column of
  // the product is the column sum of dY); one of the two waves of each n half carries it
  const bool with_db = a.db != nullptr && tile_k == 0 && wk == 0;      // wave-uniform
  f32x16 accb[2];
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (short)0x3F80;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;

  // The fragment reads are inline asm: hipcc orders every LDS read it can see behind ALL outstanding LDS-DMA requests
  // (s_waitcnt vmcnt(0)), and __syncthreads() carries a fence that does the same -- either would empty the ring at
  // every stage.  Here the waits are explicit: vmcnt(8) for the stage about to be read, lgkmcnt(0) for the fragments.
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  auto frag = [](u32x2 lo, u32x2 hi) -> bf16x8 {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const unsigned int ay0 = tr_base + (unsigned int)(wn * 2), ax0 = tr_base + (unsigned int)(IMG + wk * 2);
  issue(0);
  issue(1);
  issue(2);
  for (int st = 0; st < steps; ++st) {
    // this wave's requests of stage st have landed once at most the 8 of stages st+1, st+2 are outstanding; the barrier
    // then says the same of every wave's -- and that every wave is done reading stage st-1, whose buffer stage st+3 takes
    asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    issue(st + 3);
    const unsigned int sb = (unsigned int)((st & (NSTG - 1)) * STAGE);
    const long mrow0 = mb + (long)st * 32;
    const int mvalid = (int)(me - mrow0 < 32 ? me - mrow0 : 32);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {                          // 16 rows m per MFMA k-step
      if (ms * 16 < mvalid) {
        u32x2 y00, y01, y10, y11, x00, x01, x10, x11;
        const unsigned int ay = sb + ay0 + (unsigned int)(ms * 16 * 256), ax = sb + ax0 + (unsigned int)(ms * 16 * 256);
        asm volatile("ds_read_b64_tr_b16 %0, %8\n\t"
                     "ds_read_b64_tr_b16 %1, %8 offset:1024\n\t"
                     "ds_read_b64_tr_b16 %4, %9\n\t"
                     "ds_read_b64_tr_b16 %5, %9 offset:1024\n\t"
                     "ds_read_b64_tr_b16 %2, %8 offset:64\n\t"
                     "ds_read_b64_tr_b16 %3, %8 offset:1088\n\t"
                     "ds_read_b64_tr_b16 %6, %9 offset:64\n\t"
                     "ds_read_b64_tr_b16 %7, %9 offset:1088\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(y00), "=&v"(y01), "=&v"(y10), "=&v"(y11), "=&v"(x00), "=&v"(x01), "=&v"(x10), "=&v"(x11)
                     : "v"(ay), "v"(ax)
                     : "memory");
        bf16x8 fy[2] = {frag(y00, y01), frag(y10, y11)};
        const bf16x8 fx[2] = {frag(x00, x01), frag(x10, x11)};
        // rows beyond the slice were clamped copies: zero their contribution (m index = ms*16 + 8*(lane/32) + e)
        if (mvalid < 32) {
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (ms * 16 + 8 * (lane >> 5) + e >= mvalid) fy[t][e] = 0;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[i], fx[j], acc[i][j], 0, 0, 0);
        if (with_db) {
#pragma unroll
          for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fy[i], ones, accb[i], 0, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the look-ahead requests past the slice's end
  // D[n][k]: lane column = k (lane%32), register e -> row n = 8 (e>>2) + 4 (lane/32) + (e&3)
  const int lr = lane & 31, lh = lane >> 5;
  float *pw = a.part + (size_t)zslice * ((size_t)a.N * a.K + a.N);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kk = k0 + wk + j * 32 + lr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
        if (n < a.N && kk < a.K) pw[(long)n * a.K + kk] = acc[i][j][e];
      }
    }
  if (with_db && lr == 0) {
    float *pb = pw + (size_t)a.N * a.K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wn + i * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
        if (n < a.N) pb[n] = accb[i][e];
      }
  }
}

// (Tried: one WAVE per (128 x 64 tile, row slice) with a private four-stage ring, no barriers, fragment reads of stage
// s+1 issued before the MFMAs of stage s -- 0.75 fragments per MFMA instead of 1.  Same 200 us at N=512, K=1280.
// Ablation there: skeleton (prologue, partial tiles, fold) 68 us, + LDS-DMA 78, + fragment reads 15, + MFMAs 51 = the
// measured total: with one wave per SIMD nothing overlaps -- the address arithmetic of the six requests, the vmcnt wait,
// the reads and the MFMAs of a stage run back to back.  A 128 x 128 wave tile needs all 256 AGPRs as accumulators and
// hipcc then moves ~200 registers between the two halves of the register file per stage, with builtins and with
// "+a"-constrained asm alike.  The four-wave workgroups below at least overlap across waves.)
// dW (and db) = the slices' partial results added in a fixed order.  Workgroup = 64 x 4 consecutive values x 4 slice
// groups: a thread adds every fourth slice, eight 16-byte loads in flight, the four group sums meet in LDS in group
// order.  (One thread walking all ~128 slices of a narrow layer, eight at a time, was 16 dependent memory round trips:
// 8.7 us per launch, eight launches per step, 0.3 resident waves per SIMD.)
__global__ __launch_bounds__(256) void gemm_wgrad_fold_kernel(const float *__restrict__ part, int slices, long nw, long nb,
                                                              float *__restrict__ dW, float *__restrict__ db) {
  __shared__ float4 sg[4][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + e) * 4;
  const long tot = nw + (db ? nb : 0);
  const long pitch = nw + nb;
  float4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < tot) {
    const float4 *p = reinterpret_cast<const float4 *>(part + i);
    int z = grp;
    for (; z + 28 < slices; z += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long)(z + 4 * u) * (pitch / 4)];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; z < slices; z += 4) {
      const float4 v = p[(long)z * (pitch / 4)];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  sg[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < tot) {
    const float4 a0 = sg[0][e], a1 = sg[1][e], a2 = sg[2][e], a3 = sg[3][e];
    float4 t;
    t.x = ((a0.x + a1.x) + a2.x) + a3.x; t.y = ((a0.y + a1.y) + a2.y) + a3.y;
    t.z = ((a0.z + a1.z) + a2.z) + a3.z; t.w = ((a0.w + a1.w) + a2.w) + a3.w;
    float *o = i < nw ? dW + i : db + (i - nw);
    *reinterpret_cast<float4 *>(o) = t;
  }
}

// ------------------------------------------------------------------ weight gradient of the NARROW layers
// dW (N,K) = dY (M,N)^T . X (M,K), db = column sums of dY, for N <= 32 of any value (the 10-, 22- and 3-wide output layers
// of the heads, M4:661,678,448; the 30x30 layers of KPAM, M4:351-373) -- shapes the matrix-core kernel above does not take
// (N % 8, 16-byte rows) and the library serves as split-K bmm + a partial-sum pass + a separate column sum (~55 us a
// layer at M = 65536).  0.7 GFLOP: a streaming VALU kernel.  Workgroup = 256 rows: the dY slab goes to LDS as f32
// ([row][NMAX], zero padded), a thread owns 4 columns of X and rows rl, rl + RL, ...; per row 4 x NMAX FMAs against
// broadcast LDS reads; the row lanes meet by wave butterflies + one LDS tile per wave; per-workgroup partial
// tiles are added in workgroup order by wgrad_narrow_fold_kernel.  Either operand may be bf16 or f32.
constexpr int WN_ROWS = 256;     // rows per workgroup
constexpr int WN_CBG = 16;       // column groups (of 4) per workgroup: 64 columns

template <bool BF>
__device__ __forceinline__ float wn_load(const void *p, long i) {
  return BF ? __uint_as_float((unsigned int)((const unsigned short *)p)[i] << 16) : ((const float *)p)[i];
}

// grid = (row slabs, column blocks).  cbg = column groups per workgroup (a power of two <= 16), RL = 256 / cbg row lanes.
template <int NMAX, bool BFY, bool BFX>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const void *__restrict__ dY, const void *__restrict__ X, long M, int N,
                                                           int K, int cbg, float *__restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float wn_lds[];
  float *sdy = wn_lds;                       // [WN_ROWS][NMAX]
  float *sacc = wn_lds + WN_ROWS * NMAX;     // [4 waves][NMAX][64] + [NMAX]   (Kq = cbg * 4 <= 64 columns of this block)
  const int Kq = cbg * 4;
  const long r0 = (long)blockIdx.x * WN_ROWS;
  const int rows = (int)min((long)WN_ROWS, M - r0);
  // dY slab -> LDS [row][NMAX] f32.  The slab is one contiguous piece of memory: 16-byte chunks, all of a thread's chunks
  // requested before the first is used (a load-convert-store loop paid the memory latency 64 times: 40-60 us per launch)
  for (int i = threadIdx.x; i < WN_ROWS * NMAX / 4; i += 256) reinterpret_cast<float4 *>(sdy)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  {
    constexpr int EPC = BFY ? 8 : 4;                     // elements per 16-byte chunk
    constexpr int CH = (WN_ROWS * 32 / EPC + 255) / 256; // chunks per thread at N = 32
    const long e0 = r0 * N;
    const int ne = rows * N;
    const bool al = (((uintptr_t)dY & 15) == 0);         // slab offset e0 * sizeof is a multiple of 16 (512-row slabs)
    uint4 raw[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int e = (threadIdx.x + u * 256) * EPC;
      raw[u] = make_uint4(0u, 0u, 0u, 0u);
      if (al && e + EPC <= ne)
        raw[u] = *reinterpret_cast<const uint4 *>((const char *)dY + (e0 + e) * (BFY ? 2 : 4));
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int e = (threadIdx.x + u * 256) * EPC;
      if (e >= ne) continue;
      const bool fast = al && e + EPC <= ne;
      const unsigned int w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
      int r = e / N, n = e - r * N;
#pragma unroll
      for (int t = 0; t < EPC; ++t) {
        if (e + t < ne) {
          float v;
          if (fast) v = BFY ? __uint_as_float((t & 1) ? (w[t >> 1] & 0xffff0000u) : (w[t >> 1] << 16)) : __uint_as_float(w[t & 3]);
          else v = wn_load<BFY>(dY, e0 + e + t);
          sdy[r * NMAX + n] = v;
        }
        if (++n == N) { n = 0; ++r; }
      }
    }
  }
  __syncthreads();
  const int cg = threadIdx.x % cbg, rl = threadIdx.x / cbg, RL = 256 / cbg;
  const int cl = cg * 4;                               // column inside the block
  const int c0 = blockIdx.y * Kq + cl;                 // column of X
  float acc[NMAX][4];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f;
  const bool vec = (K % 4 == 0) && c0 < K && (((uintptr_t)X & 15) == 0);
  auto loadx = [&](int r, float (&xv)[4]) {
    const long o = (r0 + r) * K + c0;
    if (vec) {
      if (BFX) {
        const ushort4 u = *reinterpret_cast<const ushort4 *>((const unsigned short *)X + o);
        xv[0] = __uint_as_float((unsigned int)u.x << 16); xv[1] = __uint_as_float((unsigned int)u.y << 16);
        xv[2] = __uint_as_float((unsigned int)u.z << 16); xv[3] = __uint_as_float((unsigned int)u.w << 16);
      } else {
        const float4 f = *reinterpret_cast<const float4 *>((const float *)X + o);
        xv[0] = f.x; xv[1] = f.y; xv[2] = f.z; xv[3] = f.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[j] = (c0 + j < K) ? wn_load<BFX>(X, o + j) : 0.f;
    }
  };
  auto fma_row = [&](int r, const float (&xv)[4]) {
    const float4 *d4 = reinterpret_cast<const float4 *>(sdy + r * NMAX);
#pragma unroll
    for (int q = 0; q < NMAX / 4; ++q) {
      const float4 d = d4[q];
      const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[q * 4 + e][j] = fmaf(dd[e], xv[j], acc[q * 4 + e][j]);
    }
  };
  int r = rl;
  constexpr int UF = NMAX >= 16 ? 4 : 8;                     // rows in flight per thread (register budget: 4 NMAX accumulators)
  for (; r + (UF - 1) * RL < rows; r += UF * RL) {
    float xv[UF][4];
#pragma unroll
    for (int u = 0; u < UF; ++u) loadx(r + u * RL, xv[u]);
#pragma unroll
    for (int u = 0; u < UF; ++u) fma_row(r + u * RL, xv[u]);
  }
  for (; r < rows; r += RL) {
    float xv[4];
    loadx(r, xv);
    fma_row(r, xv);
  }
  // the row lanes meet: inside a wave by butterflies over the lane bits above the column group; each wave then stores
  // its tile (compile-time offsets, 16-byte stores) and the four tiles are added on the way out.  (ds_add_f32 retires
  // 0.33 lanes/clk; read-modify-write rounds with a run-time tile stride serialised into ~8 us per wave.)
  for (int off = cbg; off < 64; off <<= 1) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[n][j] += __shfl_xor(acc[n][j], off);
  }
  if ((lane_id() / cbg) == 0) {
    float *tw = sacc + wave_id() * (NMAX * 64) + cl;
#pragma unroll
    for (int n = 0; n < NMAX; ++n) *reinterpret_cast<float4 *>(tw + n * 64) = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
  }
  // column sums of dY (the bias gradient), by the workgroups of column block 0: thread t < NMAX walks column t of the slab
  // (256 / NMAX row lanes per column, eight independent LDS reads in flight, then one add per lane in lane order: a single
  // thread per column walking 256 rows was a serial ~8-us tail in every workgroup of column block 0)
  float *sbias = sacc + 4 * NMAX * 64;          // [256 / NMAX][NMAX] partial sums
  if (blockIdx.y == 0) {
    constexpr int BL = 256 / NMAX;
    const int n = threadIdx.x % NMAX, bl = threadIdx.x / NMAX;
    float sb = 0.f;
    int q = bl;
    for (; q + 7 * BL < WN_ROWS; q += 8 * BL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = sdy[(q + u * BL) * NMAX + n];     // rows >= `rows` are zero
#pragma unroll
      for (int u = 0; u < 8; ++u) sb += v[u];
    }
    for (; q < WN_ROWS; q += BL) sb += sdy[q * NMAX + n];
    sbias[bl * NMAX + n] = sb;
  }
  __syncthreads();
  float sbt = 0.f;
  if (blockIdx.y == 0 && (int)threadIdx.x < NMAX) {
    constexpr int BL = 256 / NMAX;
#pragma unroll
    for (int u = 0; u < BL; ++u) sbt += sbias[u * NMAX + threadIdx.x];
  }
  float *pw = part + (size_t)blockIdx.x * ((size_t)N * K + N);
  const int kb = min(Kq, K - (int)blockIdx.y * Kq);          // columns of this block that exist
  for (int i = threadIdx.x; i < N * kb; i += 256) {
    const int n = i / kb, c = i - n * kb;
    const float *t0 = sacc + n * 64 + c;
    pw[(long)n * K + blockIdx.y * Kq + c] = ((t0[0] + t0[NMAX * 64]) + t0[2 * NMAX * 64]) + t0[3 * NMAX * 64];
  }
  if (blockIdx.y == 0 && (int)threadIdx.x < N) pw[(size_t)N * K + threadIdx.x] = sbt;
}

// dW / db = the row slabs' partial results added in slab order.  Workgroup = 64 results x 4 slab groups (a thread adds
// every fourth slab, eight loads in flight), the four groups are combined through LDS in group order.
__global__ __launch_bounds__(256) void wgrad_narrow_fold_kernel(const float *__restrict__ part, int slices, long nw, long nb,
                                                                float *__restrict__ dW, float *__restrict__ db) {
  __shared__ float sg[4][64];
  const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + e;
  const long pitch = nw + nb;
  float s = 0.f;
  if (i < pitch) {
    int z = grp;
    for (; z + 28 < slices; z += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(long)(z + 4 * u) * pitch + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < slices; z += 4) s += part[(long)z * pitch + i];
  }
  sg[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < pitch) {
    const float t = ((sg[0][e] + sg[1][e]) + sg[2][e]) + sg[3][e];
    if (i < nw) dW[i] = t;
    else if (db) db[i - nw] = t;
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT long gcn_gemm_stats_ws_bytes(long M, int N) { return M < 0 || N < 1 ? -1 : (long)sizeof(double) * 2 * ((M + 31) / 32) * ((N + 31) / 32); }

GCN_EXPORT int gcn_gemm_bf16(const void *A, const void *W, const float *bias, void *out, int out_f32, long M, int N,
                             int Np, int K, double *gsum, void *stats_ws, int rows_per_cloud, int G, void *stream) {
  GCN_REQUIRE(A && W && out, "gcn_gemm_bf16: null pointer");
  GCN_REQUIRE(Np >= N, "gcn_gemm_bf16: W must hold Np >= N rows");
  GCN_REQUIRE(M >= 0 && N >= 1 && K >= 16 && K % 16 == 0, "gcn_gemm_bf16: need K %% 16 == 0 (pad with zeros), got M=%ld N=%d K=%d", M, N, K);
  GCN_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)out & 15) == 0, "gcn_gemm_bf16: 16-byte aligned operands");
  GCN_REQUIRE(out_f32 == 0 || out_f32 == 1, "gcn_gemm_bf16: out_f32 must be 0 or 1");
  if (gsum) {
    GCN_REQUIRE(stats_ws, "gcn_gemm_bf16: gsum needs stats_ws (gcn_gemm_stats_ws_bytes(M, N) bytes)");
    GCN_REQUIRE(G >= 1 && N % G == 0 && (N / G) % 32 == 0 && rows_per_cloud >= 128 && rows_per_cloud % 128 == 0 && M % rows_per_cloud == 0,
                "gcn_gemm_bf16: fused GroupNorm statistics need (N/G) %% 32 == 0 and rows_per_cloud %% 128 == 0");
  }
  if (M == 0) return GCN_OK;
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a{};
  a.A = (const unsigned short *)A; a.W = (const unsigned short *)W; a.bias = bias; a.out = out; a.gsum = gsum;
  a.M = M; a.N = N; a.Np = Np; a.K = K; a.out_f32 = out_f32; a.rows_per_cloud = rows_per_cloud; a.G = G;
  a.part = (double *)stats_ws;
#define GEMM_LAUNCH(BNV, BMV, NWVV)                                                                                \
  {                                                                                                                 \
    constexpr int RING = 4 * (BMV * 64 + BNV * 64);                                                                 \
    constexpr int WC = NWVV == 8 ? BNV / 4 : (BNV >= 128 ? BNV / 2 : BNV);                                          \
    constexpr int EPI = NWVV * 32 * (WC * 4 + 16);                                                                  \
    constexpr int LDSB = RING > EPI ? RING : EPI;                                                                   \
    const int mblocks = (int)((M + BMV - 1) / BMV);                                                                 \
    GCN_HIP(hipFuncSetAttribute((const void *)gemm_bf16_kernel<BNV, BMV, NWVV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
    gemm_bf16_kernel<BNV, BMV, NWVV><<<mblocks * ((N + BNV - 1) / BNV), 64 * NWVV, LDSB, st>>>(a);                 \
  }
  // N >= 256: 256 x 256 tiles on EIGHT waves (128 x 64 per wave, two waves per SIMD).  M = 65536, us, forward without /
  // with fused statistics (tools/gemm_bench.py; lib = hipBLASLt through torch, which needs a 16-us statistics pass on top):
  //     N x K        8 waves        4 waves 128x128/wave   4 waves 64x64/wave (round 2)    lib
  //   512 x 1280   122 / 125 (689 TF)   146 / 163              149 / 155                     70
  //   512 x 256     43 / 55              75 / 92                49 / 63                      31
  //   256 x 832     42 / 47              60 / 69                48 / 56                      30
  //   256 x 512     31 / 39              50 / 59                36 / 43                      24
  //   256 x 256     23 / 30              41 / 50                27 / 34                      20
  //  1024 x 256     81 / 97             142 / 170               87 / 107                     56
  // The one-wave-per-SIMD 128 x 128 variant (GCANET_GEMM_TILE=256) starves on its LDS-DMA ring (PMC: matrix pipe busy
  // 30 %, waves in s_waitcnt 40 % of their cycles, no LDS conflicts); with two waves per SIMD one wave's waits hide
  // under the other's MFMAs.  Still 0.6-0.85x of the library on the wide layers -- the policy in gcanet_amd/layers.py
  // keeps those on hipBLASLt and uses this kernel where its fused statistics pay (N * K <= 256 * 256).
  const char *tile_env = getenv("GCANET_GEMM_TILE");
  const int tile_sel = tile_env ? atoi(tile_env) : 2568;     // 256: 4 waves x 128x128; 128: the round-2 tile everywhere
  if (N >= 256 && M >= 256 * 128 && tile_sel == 256) GEMM_LAUNCH(256, 256, 4)
  else if (N >= 256 && M >= 256 * 128 && tile_sel == 2568) GEMM_LAUNCH(256, 256, 8)
  else if (N > 64) GEMM_LAUNCH(128, 128, 4) else if (N > 32) GEMM_LAUNCH(64, 128, 4) else GEMM_LAUNCH(32, 128, 4)
#undef GEMM_LAUNCH
  int rc = check_launch("gemm_bf16_kernel");
  if (rc || !gsum) return rc;
  gemm_stats_reduce_kernel<<<(int)(M / rows_per_cloud) * G, 256, 0, st>>>(a.part, gsum, rows_per_cloud / 32, N / 32, G);
  return check_launch("gemm_stats_reduce_kernel");
}

namespace gcn {
struct WgradPlan { int tiles, split, slices; long mper; };
static WgradPlan wgrad_plan(long M, int N, int K) {
  WgradPlan p{};
  p.tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int target = 512;                                          // two workgroups per CU
  if (const char *e = getenv("GCANET_WGRAD_WGS")) target = atoi(e);
  int split = (target + p.tiles - 1) / p.tiles;
  split = (split + 7) / 8 * 8;                               // one row slice per XCD and round
  if ((long)split * p.tiles > target + target / 4 && split > 8) split -= 8;
  const long maxsplit = (M + 255) / 256;                     // at least 256 rows each
  if (split > maxsplit) split = (int)maxsplit;
  if (split < 1) split = 1;
  p.split = split;
  p.mper = ((M + split - 1) / split + 31) / 32 * 32;
  p.slices = (int)((M + p.mper - 1) / p.mper);               // non-empty slices (the rounding of mper may drop the last)
  return p;
}
}  // namespace gcn

GCN_EXPORT long gcn_gemm_wgrad_ws_bytes(long M, int N, int K) {
  if (M < 1 || N < 8 || K < 8) return -1;
  const WgradPlan p = wgrad_plan(M, N, K);
  return (long)(sizeof(float) * (size_t)p.slices * ((size_t)N * K + N));
}

GCN_EXPORT int gcn_gemm_wgrad_bf16(const void *dY, const void *X, long M, int N, int K, float *dW, float *db, void *ws,
                                   void *stream) {
  GCN_REQUIRE(dY && X && dW && ws, "gcn_gemm_wgrad_bf16: null pointer");
  GCN_REQUIRE(M >= 1 && N >= 8 && K >= 8 && N % 8 == 0 && K % 8 == 0, "gcn_gemm_wgrad_bf16: need N %% 8 == 0 and K %% 8 == 0, got N=%d K=%d", N, K);
  GCN_REQUIRE(((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)dW & 15) == 0 &&
              (!db || ((uintptr_t)db & 15) == 0), "gcn_gemm_wgrad_bf16: 16-byte aligned operands");
  hipStream_t st = (hipStream_t)stream;
  const WgradPlan p = wgrad_plan(M, N, K);
  WgradArgs a{};
  a.dY = (const unsigned short *)dY; a.X = (const unsigned short *)X; a.dW = dW; a.db = db; a.M = M; a.N = N; a.K = K;
  a.part = (float *)ws; a.msplit = p.slices; a.mper = p.mper;
  const int LDSB = 4 * 2 * 32 * 256;
  GCN_HIP(hipFuncSetAttribute((const void *)gemm_wgrad_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  gemm_wgrad_bf16_kernel<<<p.tiles * ((p.slices + 7) / 8) * 8, 256, LDSB, st>>>(a);
  int rc = check_launch("gemm_wgrad_bf16_kernel");
  if (rc) return rc;
  const long nw = (long)N * K, nb = N, tot = nw + (db ? nb : 0);
  gemm_wgrad_fold_kernel<<<(int)((tot / 4 + 63) / 64), 256, 0, st>>>(a.part, p.slices, nw, nb, dW, db);
  return check_launch("gemm_wgrad_fold_kernel");
}

static int wn_kcp(int K) {                 // column groups of 4, rounded up to a power of two
  int kc = (K + 3) / 4, p = 1;
  while (p < kc) p <<= 1;
  return p;
}
static int wn_nmax(int N) { return N <= 4 ? 4 : (N <= 16 ? 16 : 32); }
static size_t wn_lds_bytes(int N, int K) {
  const int cbg = std::min(wn_kcp(K), WN_CBG);
  (void)cbg;
  return sizeof(float) * ((size_t)WN_ROWS * wn_nmax(N) + (size_t)4 * wn_nmax(N) * 64 + 256);
}

GCN_EXPORT int gcn_wgrad_narrow_supported(long M, int N, int K) {
  if (M < 1 || N < 1 || N > 32 || K < 1 || K > 1024) return 0;
  return wn_lds_bytes(N, K) <= 128 * 1024 ? 1 : 0;
}

GCN_EXPORT long gcn_wgrad_narrow_ws_bytes(long M, int N, int K) {
  if (!gcn_wgrad_narrow_supported(M, N, K)) return -1;
  return (long)(sizeof(float) * (size_t)((M + WN_ROWS - 1) / WN_ROWS) * ((size_t)N * K + N));
}

template <int NMAX>
static int wn_launch(const void *dY, int y_bf16, const void *X, int x_bf16, long M, int N, int K, float *part, hipStream_t st) {
  const int kcp = wn_kcp(K), cbg = std::min(kcp, WN_CBG);
  const size_t lds = wn_lds_bytes(N, K);
  const dim3 grid((unsigned int)((M + WN_ROWS - 1) / WN_ROWS), (unsigned int)(kcp / cbg));
#define WN_GO(BY, BX)                                                                                                   \
  do {                                                                                                                  \
    GCN_HIP(hipFuncSetAttribute((const void *)wgrad_narrow_kernel<NMAX, BY, BX>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds));                                                                             \
    wgrad_narrow_kernel<NMAX, BY, BX><<<grid, 256, lds, st>>>(dY, X, M, N, K, cbg, part);                                 \
  } while (0)
  if (y_bf16 && x_bf16) WN_GO(true, true);
  else if (y_bf16) WN_GO(true, false);
  else if (x_bf16) WN_GO(false, true);
  else WN_GO(false, false);
#undef WN_GO
  return check_launch("wgrad_narrow_kernel");
}

GCN_EXPORT int gcn_wgrad_narrow(const void *dY, int dy_bf16, const void *X, int x_bf16, long M, int N, int K, float *dW,
                                float *db, void *ws, void *stream) {
  GCN_REQUIRE(dY && X && dW && ws, "gcn_wgrad_narrow: null pointer");
  GCN_REQUIRE(gcn_wgrad_narrow_supported(M, N, K), "gcn_wgrad_narrow: need 1 <= N <= 32, 1 <= K <= 1024 within the LDS budget, got N=%d K=%d", N, K);
  GCN_REQUIRE(((uintptr_t)ws & 15) == 0, "gcn_wgrad_narrow: ws must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float *part = (float *)ws;
  int rc;
  const int nmax = wn_nmax(N);
  if (nmax == 4) rc = wn_launch<4>(dY, dy_bf16, X, x_bf16, M, N, K, part, st);
  else if (nmax == 16) rc = wn_launch<16>(dY, dy_bf16, X, x_bf16, M, N, K, part, st);
  else rc = wn_launch<32>(dY, dy_bf16, X, x_bf16, M, N, K, part, st);
  if (rc) return rc;
  const long nw = (long)N * K, nb = N;
  wgrad_narrow_fold_kernel<<<(int)((nw + nb + 63) / 64), 256, 0, st>>>(part, (int)((M + WN_ROWS - 1) / WN_ROWS), nw, nb, dW, db);
  return check_launch("wgrad_narrow_fold_kernel");
}
