// segdiam.hip -- per-segment feature DIAMETER of forward_grouping (models/dgcnn-hais-concat-direct-4.py:210-233: the global
// `max` of the (n,n) cdist matrix that compute_batch_adjacency_matrix normalises by), for gfx950, as
// SPLIT-bf16 BOUND PASSES on the matrix cores + EXACT RECHECK of the few blocks that can hold the maximum.
//
// softgroup.hip: seg_diameter_kernel evaluates every pair of a segment in the f32 arithmetic whose maximum is the result
// (expanded form xx_i + xx_j - 2 f_i.f_j, Gram blocks on v_mfma_f32_16x16x4_f32): O(n^2 C) at the f32 matrix rate, ~1.2 ms
// per call for 8 segments of 8192 rows at C = 64 -- the largest kernel of the literal forward_train step.  Only the
// MAXIMUM is needed and it is attained in a handful of 32 x 32 blocks.
//
// Every row is split f = h + l + r with h = bf16(f), l = bf16(f - h): |l| <= 2^-8 |f| (1 + 2^-8), |r| <= 2^-16 |f| (bf16
// keeps 8 significant bits).  g_ij = h_i.h_j + h_i.l_j + l_i.h_j on v_mfma_f32_32x32x16_bf16 (three products at 16x the
// f32 rate; plain bf16 products resolve 0.8 % of the diameter, which trained -- tightly clustered -- features do not
// survive: every block then holds a pair within the error of the maximum).  With K = 3 Cb accumulated terms
//     |f_i.f_j - g_ij| <= (3.03 * 2^-16 + K 2^-23) |f_i| |f_j|,    and the exact kernel's own sum errs by <= C 2^-23 |f_i||f_j|,
// so with |f_i||f_j| <= (xx_i + xx_j) / 2 (no square roots in the inner loop) its value v_ij = (xx_i + xx_j) - 2 acc_ij obeys
//     lo_ij = (1 - e)(xx_i + xx_j) - 2 g_ij  <=  v_ij  <=  (1 + e)(xx_i + xx_j) - 2 g_ij = hi_ij,     e = 5e-5 + 5e-7 Cb.
//   1. bound    every tile pair (a <= b): L = max lo_ij per segment (per-lane running maxima, one atomic per work item)
//   2. select   the same products again: a 32 x 32 block with some hi_ij >= L goes on a list (one ballot per block)
//   3. exact    the listed blocks in EXACTLY the arithmetic of seg_diameter_kernel (same operand order of the same
//               MFMAs), so the result is bit-identical to the exhaustive kernel (tests/test_grouping_gpu.py).
// Passes 1 and 2 recompute instead of storing per-tile bounds: the bf16 products cost less than the cross-lane
// reductions a stored per-tile maximum needs.  A workgroup owns a 256-row tile (A fragments stay in registers) and a
// strip of up to 16 column tiles staged through LDS; rows of 16 <= C <= 128 channels.
// Segments of <= SD_SMALL_T row tiles, and every segment when more than a sixth of the blocks was listed (features that
// f32 itself barely separates, e.g. identical rows), go through the exhaustive kernel instead -- decided on the device.
#include "common.h"

namespace gcn {

typedef __attribute__((ext_vector_type(8))) short sd_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sd_f32x16;
using sd_f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

constexpr int SD_STRIP = 16;                // column tiles per work item
constexpr int SD_SMALL_T = 32;              // segments of <= 32 row tiles (2048 rows) go to the exhaustive kernel: three
                                            // passes over 528 tile pairs cost more than evaluating them

struct SdArgs {
  const float *f;             // (n, C) f32 rows in segment order
  unsigned short *fb;         // (n, 2 Cb) bf16: row i = (h_i | l_i), Cb = C rounded up to 16, 32, 64, 128 or 256 (zeros)
  float *xx;                  // (n) squared norms in the exact kernel's order (ascending fmaf chain)
  const int32_t *seg_offsets; // (S+1)
  const int32_t *seg_cls;     // (S), < 0: inactive
  int32_t *work_prefix;       // (S+1) (row tile, strip) work items before segment s (segments of > SD_SMALL_T tiles)
  int32_t *tiles_small;       // (S+1) row tiles before segment s, small segments only   } the exhaustive kernel's
  int32_t *tiles_large;       // (S+1) the same for the large ones (the fall-back)       } tile_prefix argument
  unsigned int *L;            // (S) float bits of max lo
  int4 *cand;                 // listed blocks: (first row, first column, segment, -)
  unsigned int *ncand;        // [0] listed blocks, [1] the limit beyond which the large segments fall back
  unsigned int cand_cap;      // entries `cand` holds (every block of every large segment: sd_layout)
  unsigned int *dmax2;        // (S) float bits
  float eps;
  int n, C, Cb, S;
};

__device__ __forceinline__ unsigned short sd_f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float sd_bf2f(unsigned short h) { return __uint_as_float((unsigned int)h << 16); }

__global__ __launch_bounds__(256) void sd_prep_kernel(SdArgs a) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;        // one 8-channel chunk of one row
  const int cpr = a.Cb / 8;
  if (e < (long)a.n * cpr) {
    const long row = e / cpr;
    const int c0 = (int)(e % cpr) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c0 < a.C) {                                           // C is a multiple of 16: a chunk is inside or outside
      const float4 *src = reinterpret_cast<const float4 *>(a.f + row * a.C + c0);
      const float4 v0 = src[0], v1 = src[1];
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    sd_bf16x8 h, l;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned short hk = sd_f2bf(v[k]);
      h[k] = (short)hk;
      l[k] = (short)sd_f2bf(v[k] - sd_bf2f(hk));              // f - h is exact in f32
    }
    unsigned short *dst = a.fb + row * 2 * a.Cb + c0;
    *reinterpret_cast<sd_bf16x8 *>(dst) = h;
    *reinterpret_cast<sd_bf16x8 *>(dst + a.Cb) = l;
  }
  if (e < a.n) {                                              // the first n threads also take one row's norm each
    const float *row = a.f + e * a.C;
    float s = 0.f;
    for (int c = 0; c < a.C; ++c) s = fmaf(row[c], row[c], s);   // softgroup.hip: row_sqnorm_kernel's order
    a.xx[e] = s;
  }
}

// work items of a segment of T column tiles: a workgroup owns a 256-row tile A (column tiles 4 A ..) and one strip of
// SD_STRIP column tiles; the items are numbered as a rectangle (row tile, strip) and the strips left of the diagonal
// (no column tile >= 4 A in them) return at once -- a closed form of the triangle saves nothing at <= 2000 items
__device__ __forceinline__ int sd_strips(int T) { return (T + SD_STRIP - 1) / SD_STRIP; }
__device__ __forceinline__ int sd_work_items(int T) { return ((T + 3) / 4) * sd_strips(T); }

__global__ void sd_prefix_kernel(SdArgs a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int work = 0, small = 0, large = 0;
  long blocks = 0;
  for (int s = 0; s < a.S; ++s) {
    a.work_prefix[s] = work;
    a.tiles_small[s] = small;
    a.tiles_large[s] = large;
    if (a.seg_cls[s] < 0) continue;
    const int T = (a.seg_offsets[s + 1] - a.seg_offsets[s] + 63) / 64;
    if (T > SD_SMALL_T) { work += sd_work_items(T); large += T; blocks += 2L * T * (T + 1); } else small += T;
  }
  a.work_prefix[a.S] = work;
  a.tiles_small[a.S] = small;
  a.tiles_large[a.S] = large;
  a.ncand[1] = (unsigned int)(blocks / 6);
}

// after the select pass: the exhaustive kernel takes the large segments only when too many blocks were listed
__global__ void sd_fallback_kernel(SdArgs a) {
  if (a.ncand[0] <= a.ncand[1]) a.tiles_large[a.S] = 0;       // seg_diameter_kernel: `if (tile >= tile_prefix[S]) return`
}

// ------------------------------------------------------------------ 1. / 2. bound and select passes
// 512 threads: wave w owns rows 32 w .. 32 w + 31 of the 256-row tile (its h and l fragments stay in registers) and both
// 32-column halves of every column tile.  The (h | l) rows of a column tile are staged ONCE per workgroup in LDS (the
// 64 x 64-tile form of this kernel had each wave fetch its own operands: 2 GB of L2 reads per pass, 0.28 ms, L2-bound),
// double buffered: the global loads of tile tb + 1 are issued before the MFMAs of tile tb and stored after them, one
// barrier per tile.  Staged rows are 16 bytes longer than the data so that 16 consecutive lanes' ds_read_b128 cover all
// banks.
template <int KS, bool SELECT>
__global__ __launch_bounds__(512) void sd_pass_kernel(SdArgs a) {
  constexpr int Cb = KS * 16;
  constexpr int ROWB = 4 * Cb + 16;                        // staged bytes per column: h (2 Cb) | l (2 Cb) | pad
  constexpr int CPC = Cb / 4;                              // 16-byte chunks per column
  constexpr int NCH = (64 * CPC + 511) / 512;              // chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char sd_lds[];    // 2 x 64 x ROWB
  const int lane = lane_id(), wave = wave_id();
  const int lr = lane & 31, lh = lane >> 5;
  const int total = a.work_prefix[a.S];
  const float scale = SELECT ? 1.f + a.eps : 1.f - a.eps;
  for (int w = blockIdx.x; w < total; w += gridDim.x) {
    int slo = 0, shi = a.S;                  // last s with work_prefix[s] <= w: it owns item w (empty ones repeat the prefix)
    while (shi - slo > 1) {
      const int mid = (slo + shi) >> 1;
      if (a.work_prefix[mid] <= w) slo = mid; else shi = mid;
    }
    const int sg = slo;
    const int beg = a.seg_offsets[sg], end = a.seg_offsets[sg + 1];
    const int T = (end - beg + 63) / 64;
    const int nstrip = sd_strips(T);
    const int q = w - a.work_prefix[sg];
    const int A = q / nstrip;
    const int tb0 = max(4 * A, (q % nstrip) * SD_STRIP), tb1 = min(T, (q % nstrip + 1) * SD_STRIP);
    if (tb0 >= tb1) continue;                                  // strip left of the diagonal (uniform over the workgroup)

    const int i0 = beg + A * 256 + 32 * wave;                  // this wave's 32 rows (first operand: register index)
    const unsigned short *arow = a.fb + (long)min(i0 + lr, end - 1) * (2 * Cb) + lh * 8;
    sd_bf16x8 ah[KS], al[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ah[s] = *reinterpret_cast<const sd_bf16x8 *>(arow + s * 16);
      al[s] = *reinterpret_cast<const sd_bf16x8 *>(arow + Cb + s * 16);
    }
    float xi[16];                                              // acc[i] is row i0 + 4 lh + (i & 3) + 8 (i >> 2)
#pragma unroll
    for (int i = 0; i < 16; ++i) xi[i] = scale * a.xx[min(i0 + 4 * lh + (i & 3) + 8 * (i >> 2), end - 1)];
    const float Lseg = SELECT ? __uint_as_float(a.L[sg]) : 0.f;
    float run = 0.f;

    sd_bf16x8 stage[NCH];
    float xjn[2];
    auto fetch = [&](int tb) {                                 // global -> registers: tile tb's (h | l) rows and norms
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int c = (int)threadIdx.x + 512 * u;
        if (64 * CPC >= 512 || c < 64 * CPC) {
          const int jc = min(beg + tb * 64 + c / CPC, end - 1);
          stage[u] = *reinterpret_cast<const sd_bf16x8 *>(a.fb + (long)jc * (2 * Cb) + (c % CPC) * 8);
        }
      }
      xjn[0] = a.xx[min(beg + tb * 64 + lr, end - 1)];
      xjn[1] = a.xx[min(beg + tb * 64 + 32 + lr, end - 1)];
    };
    auto store = [&](int buf) {                                // registers -> LDS
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        const int c = (int)threadIdx.x + 512 * u;
        if (64 * CPC >= 512 || c < 64 * CPC)
          *reinterpret_cast<sd_bf16x8 *>(sd_lds + buf * (64 * ROWB) + (c / CPC) * ROWB + (c % CPC) * 16) = stage[u];
      }
    };
    fetch(tb0);
    store(0);
    __syncthreads();
    int cur = 0;
    for (int tb = tb0; tb < tb1; ++tb) {
      const float xj0 = scale * xjn[0], xj1 = scale * xjn[1];
      if (tb + 1 < tb1) fetch(tb + 1);
      const unsigned char *bt = sd_lds + cur * (64 * ROWB) + lh * 16;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const unsigned char *bcol = bt + (half * 32 + lr) * ROWB;
        sd_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const sd_bf16x8 bh = *reinterpret_cast<const sd_bf16x8 *>(bcol + s * 32);
          const sd_bf16x8 bl = *reinterpret_cast<const sd_bf16x8 *>(bcol + 2 * Cb + s * 32);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc, 0, 0, 0);       // small terms first
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc, 0, 0, 0);
        }
        // clamped duplicate rows are real rows; i == j gives lo <= ~0 (never a maximum) and at worst one more listed block
        const float xj = half ? xj1 : xj0;
        float m = SELECT ? -1.f : run;
#pragma unroll
        for (int i = 0; i < 16; i += 2)
          m = fmaxf(m, fmaxf(fmaf(-2.f, acc[i], xi[i] + xj), fmaf(-2.f, acc[i + 1], xi[i + 1] + xj)));
        if (SELECT) {
          if (__ballot(m >= Lseg) != 0ull && lane == 0) {
            const unsigned int slot = atomicAdd(a.ncand, 1u);
            if (slot < a.cand_cap) a.cand[slot] = make_int4(i0, beg + tb * 64 + half * 32, sg, 0);   // (a full list means fall-back)
          }
        } else {
          run = m;
        }
      }
      if (tb + 1 < tb1) store(cur ^ 1);
      __syncthreads();                                         // tile tb + 1 is staged; every wave is done with tile tb
      cur ^= 1;
    }
    if (!SELECT) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) run = fmaxf(run, __shfl_xor(run, o));
      // non-negative floats order as their bits; a (possibly stale) read first: tens of thousands of waves raising the
      // same word serialised into 0.4 ms
      if (lane == 0 && __float_as_uint(run) > __atomic_load_n(a.L + sg, __ATOMIC_RELAXED)) atomicMax(a.L + sg, __float_as_uint(run));
    }
  }
}

// ------------------------------------------------------------------ 3. exact values of the listed blocks
// One wave per listed 32 x 32 block, as 2 x 2 sub-blocks of 16 x 16: the arithmetic of softgroup.hip: seg_diameter_kernel
// (per 16-channel chunk four v_mfma_f32_16x16x4_f32 in x, y, z, w order; lane group g = lane / 16 supplies channels
// 4 g .. 4 g + 3 of the chunk on both operands).
__global__ __launch_bounds__(256) void sd_exact_kernel(SdArgs a) {
  const int lane = lane_id(), wave = wave_id();
  const int li = lane & 15, lk = lane >> 4;
  const unsigned int ncand = a.ncand[0];
  if (ncand > a.ncand[1]) return;                               // sd_fallback_kernel: the exhaustive kernel takes over
  for (unsigned int e = blockIdx.x * 4 + wave; e < ncand; e += gridDim.x * 4) {
    const int4 cd = a.cand[e];
    const int i0 = cd.x, j0 = cd.y, sg = cd.z;
    const int end = a.seg_offsets[sg + 1];
    const float *arow[2], *brow[2];
    float xi[2][4];
    sd_f32x4 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      arow[h] = a.f + (long)min(i0 + 16 * h + li, end - 1) * a.C + 4 * lk;
      brow[h] = a.f + (long)min(j0 + 16 * h + li, end - 1) * a.C + 4 * lk;
#pragma unroll
      for (int r = 0; r < 4; ++r) xi[h][r] = a.xx[min(i0 + 16 * h + 4 * lk + r, end - 1)];
      acc[h][0] = {0.f, 0.f, 0.f, 0.f};
      acc[h][1] = {0.f, 0.f, 0.f, 0.f};
    }
    for (int kc = 0; kc < a.C; kc += 16) {
      float4 av[2], bv[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        av[h] = *reinterpret_cast<const float4 *>(arow[h] + kc);
        bv[h] = *reinterpret_cast<const float4 *>(brow[h] + kc);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h].x, bv[t].x, acc[h][t], 0, 0, 0);
          acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h].y, bv[t].y, acc[h][t], 0, 0, 0);
          acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h].z, bv[t].z, acc[h][t], 0, 0, 0);
          acc[h][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h].w, bv[t].w, acc[h][t], 0, 0, 0);
        }
    }
    float best = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int j = j0 + 16 * t + li;
      const float xj = a.xx[min(j, end - 1)];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = i0 + 16 * h + 4 * lk + r;
          const float d2 = (xi[h][r] + xj) - 2.f * acc[h][t][r];
          if (i < end && j < end && i != j) best = fmaxf(best, d2);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    // a (possibly stale) read first: thousands of waves raising one word retire one every ~10 ns
    if (lane == 0 && __float_as_uint(best) > __atomic_load_n(a.dmax2 + sg, __ATOMIC_RELAXED))
      atomicMax(a.dmax2 + sg, __float_as_uint(best));
  }
}

static size_t sd_align(size_t v) { return (v + 255) & ~(size_t)255; }
static int sd_cb(int C) { int cb = 16; while (cb < C) cb *= 2; return cb; }
constexpr int SD_CMAX = 128;                // wider rows: registers and LDS of the pass kernel; the caller keeps the exhaustive kernel
struct SdWs { size_t L, ncand, work_prefix, tiles_small, tiles_large, fb, xx, cand, total; long blocks_max, work_max; };
static SdWs sd_layout(int n, int C, int S) {
  SdWs w{};
  const long Tmax = (long)n / 64 + S + 1;                     // 64-row tiles of all segments together
  // 32 x 32 blocks the select pass can list: a segment of T column tiles walks 16 (T - 4 A) blocks per 256-row tile A
  // (all 8 waves, clamped rows past the end included), 2 T^2 + 8 T + 8 <= 2 (T + 2)^2 in total, and
  // sum_s (T_s + 2)^2 <= (sum_s T_s + 2 S)^2 with sum_s T_s <= n / 64 + S
  w.blocks_max = 2 * (Tmax + 2 * S + 1) * (Tmax + 2 * S + 1);
  w.work_max = (Tmax / 4 + S + 1) * (Tmax / SD_STRIP + 1);
  size_t o = 0;
  w.L = o; o += sd_align(sizeof(unsigned int) * (size_t)S);          // L and ncand are zeroed together
  w.ncand = o; o += 256;
  w.work_prefix = o; o += sd_align(sizeof(int32_t) * (size_t)(S + 1));
  w.tiles_small = o; o += sd_align(sizeof(int32_t) * (size_t)(S + 1));
  w.tiles_large = o; o += sd_align(sizeof(int32_t) * (size_t)(S + 1));
  w.fb = o; o += sd_align(4 * (size_t)n * sd_cb(C));
  w.xx = o; o += sd_align(sizeof(float) * (size_t)n);
  w.cand = o; o += sd_align(sizeof(int4) * (size_t)w.blocks_max);
  w.total = o;
  return w;
}

}  // namespace gcn

using namespace gcn;

// softgroup.hip: seg_diameter_kernel over the tiles tile_prefix assigns
void launch_seg_diameter_tiles(int n, int S, int C, const float *feats, const float *xx, const int32_t *seg_offsets,
                               const int32_t *tile_prefix, float *dmax2, hipStream_t st);

GCN_EXPORT long gcn_segment_diameter2_ws_bytes(int n, int C, int S) {
  if (n < 0 || C < 16 || C % 16 != 0 || C > SD_CMAX || S < 0) return -1;
  const SdWs w = sd_layout(n, C, S);
  if (w.blocks_max > 0x7fffffffL) return -1;
  return (long)w.total;
}

GCN_EXPORT int gcn_segment_diameter2_filtered(int n, int C, const float *feats, const int32_t *seg_offsets,
                                              const int32_t *seg_cls, int S, void *ws, float *dmax2, void *stream) {
  GCN_REQUIRE(n >= 0 && S >= 0 && C >= 16 && C % 16 == 0 && C <= SD_CMAX, "gcn_segment_diameter2_filtered: C must be a multiple of 16 in [16, 128] (zero-pad the rows; gcn_segment_diameter2 beyond), got C=%d", C);
  if (n == 0 || S == 0) return GCN_OK;
  GCN_REQUIRE(feats && seg_offsets && seg_cls && ws && dmax2, "gcn_segment_diameter2_filtered: null pointer");
  GCN_REQUIRE(((uintptr_t)ws & 255) == 0 && ((uintptr_t)feats & 15) == 0, "gcn_segment_diameter2_filtered: ws must be 256-B aligned, feats 16-B aligned");
  const SdWs w = sd_layout(n, C, S);
  GCN_REQUIRE(w.blocks_max <= 0x7fffffffL, "gcn_segment_diameter2_filtered: too many tile pairs (see gcn_segment_diameter2_ws_bytes)");
  hipStream_t st = (hipStream_t)stream;
  char *base = (char *)ws;
  SdArgs a{};
  a.f = feats; a.fb = (unsigned short *)(base + w.fb); a.xx = (float *)(base + w.xx); a.seg_offsets = seg_offsets;
  a.seg_cls = seg_cls; a.work_prefix = (int32_t *)(base + w.work_prefix);
  a.tiles_small = (int32_t *)(base + w.tiles_small); a.tiles_large = (int32_t *)(base + w.tiles_large); a.L = (unsigned int *)(base + w.L);
  a.cand = (int4 *)(base + w.cand); a.ncand = (unsigned int *)(base + w.ncand); a.dmax2 = (unsigned int *)dmax2;
  a.n = n; a.C = C; a.Cb = sd_cb(C); a.S = S;
  a.eps = 5e-5f + 5e-7f * (float)a.Cb;
  a.cand_cap = (unsigned int)w.blocks_max;
  GCN_HIP(fill_dev(base + w.L, 0, w.work_prefix - w.L, st));
  GCN_HIP(fill_dev(dmax2, 0, sizeof(float) * (size_t)S, st));
  sd_prep_kernel<<<cdiv((long)n * (a.Cb / 8), 256), 256, 0, st>>>(a);
  sd_prefix_kernel<<<1, 1, 0, st>>>(a);
  const int grid = (int)(w.work_max < 2048 ? w.work_max : 2048);
#define SD_PASSES(KS)                                                                                               \
  {                                                                                                                 \
    constexpr int LDSB = 2 * 64 * (4 * KS * 16 + 16);                                                               \
    GCN_HIP(hipFuncSetAttribute((const void *)sd_pass_kernel<KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
    GCN_HIP(hipFuncSetAttribute((const void *)sd_pass_kernel<KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));  \
    sd_pass_kernel<KS, false><<<grid, 512, LDSB, st>>>(a);                                                          \
    sd_pass_kernel<KS, true><<<grid, 512, LDSB, st>>>(a);                                                           \
  }
  switch (a.Cb) {
    case 16: SD_PASSES(1); break;
    case 32: SD_PASSES(2); break;
    case 64: SD_PASSES(4); break;
    default: SD_PASSES(8); break;
  }
#undef SD_PASSES
  sd_fallback_kernel<<<1, 1, 0, st>>>(a);
  sd_exact_kernel<<<2048, 256, 0, st>>>(a);
  launch_seg_diameter_tiles(n, S, C, feats, a.xx, seg_offsets, a.tiles_small, dmax2, st);
  launch_seg_diameter_tiles(n, S, C, feats, a.xx, seg_offsets, a.tiles_large, dmax2, st);
  return check_launch("sd_exact_kernel");
}
