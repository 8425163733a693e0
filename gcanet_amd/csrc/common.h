// common.h -- shared helpers for the gfx950 kernels of libgcanet_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gcanet_hip.h"

#define GCN_EXPORT extern "C" __attribute__((visibility("default")))

namespace gcn {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return GCN_ELAUNCH;
  }
  return GCN_OK;
}

#define GCN_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      gcn::set_error(__VA_ARGS__);    \
      return GCN_EINVAL;              \
    }                                 \
  } while (0)

#define GCN_HIP(call)                                                   \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) {                                             \
      gcn::set_error("%s: %s", #call, hipGetErrorString(e_));           \
      return GCN_ELAUNCH;                                               \
    }                                                                   \
  } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// wave id within the block as a scalar (SGPR) value
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int readlane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }

// lane i <- lane i-1 across the whole wave (DPP wave_shr:1); lane 0 <- lane0_val
__device__ __forceinline__ int wave_shr1_i(int lane0_val, int v) {
  return __builtin_amdgcn_update_dpp(lane0_val, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ float wave_shr1_f(float lane0_val, float v) {
  return __builtin_bit_cast(float, wave_shr1_i(__builtin_bit_cast(int, lane0_val), __builtin_bit_cast(int, v)));
}

// (tile, cloud) of this workgroup in a (tiles, clouds) grid.  Workgroups are dealt to the 8 XCDs round robin by linear
// id, each XCD with its own 4 MB L2: cloud = id % clouds makes the workgroups of one XCD share clouds (one cloud per
// XCD at 8 clouds), so the rows a cloud's neighbour lists keep re-reading stay in ONE L2 instead of thrashing all eight.
__device__ __forceinline__ void xcd_tile_cloud(int &tile, int &cloud) {
  const int lin = blockIdx.x + gridDim.x * blockIdx.y;
  cloud = lin % (int)gridDim.y;
  tile = lin / (int)gridDim.y;
}

// Same idea when there are many more "clouds" (attention: batch x heads) than XCDs: XCD x = linear id % 8 works through
// clouds x, x+8, ... one after the other, all tiles of a cloud consecutively, so that one cloud's streamed operand
// (K/V: 1-2 MB) is what that XCD's L2 holds at a time.  Falls back to the plain (x, y) grid when clouds % 8 != 0.
__device__ __forceinline__ void xcd_major_tile_cloud(int &tile, int &cloud) {
  const int T = gridDim.x, Cn = gridDim.y;
  if (Cn % 8 != 0) { tile = blockIdx.x; cloud = blockIdx.y; return; }
  const int lin = blockIdx.x + T * blockIdx.y;
  const int xcd = lin & 7, slot = lin >> 3;                 // slot: this XCD's running workgroup number
  cloud = (slot / T) * 8 + xcd;
  tile = slot % T;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device-wide exclusive scan of int32 rows (three small launches; every access coalesced) ----
// rows arrays of length m, laid out v + row*m; bsum: rows * scan_blocks(m) ints of scratch.  copy (optional, same
// layout as v) receives the same result.  4096 elements per workgroup.
inline int scan_blocks(long m) { return (int)((m + 4095) / 4096); }

__device__ __forceinline__ int block_exscan_1024(int sum, int *wtot, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(inc, d); if (lane >= d) inc += y; }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  int run = inc - sum, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { const int t = wtot[w]; run += (w < wave) ? t : 0; tot += t; }
  total = tot;
  __syncthreads();
  return run;
}

static __global__ __launch_bounds__(1024) void scan_block_sums_kernel(int m, const int32_t *__restrict__ v, int32_t *__restrict__ bsum) {
  __shared__ int wtot[16];
  const int32_t *row = v + (long)blockIdx.y * m;
  const int base = blockIdx.x * 4096 + threadIdx.x * 4;
  int s = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) s += (base + e < m) ? row[base + e] : 0;
  int total;
  block_exscan_1024(s, wtot, total);
  if (threadIdx.x == 0) bsum[blockIdx.y * gridDim.x + blockIdx.x] = total;
}

static __global__ __launch_bounds__(1024) void scan_block_offsets_kernel(int nb, int32_t *__restrict__ bsum) {
  __shared__ int wtot[16];
  int32_t *row = bsum + (long)blockIdx.x * nb;
  int carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int c = i < nb ? row[i] : 0;
    int total;
    const int ex = block_exscan_1024(c, wtot, total);
    if (i < nb) row[i] = carry + ex;
    carry += total;
  }
}

static __global__ __launch_bounds__(1024) void scan_apply_kernel(int m, int32_t *__restrict__ v, const int32_t *__restrict__ bsum,
                                                                 int32_t *__restrict__ copy) {
  __shared__ int wtot[16];
  int32_t *row = v + (long)blockIdx.y * m;
  const int base = blockIdx.x * 4096 + threadIdx.x * 4;
  int c[4], s = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) { c[e] = (base + e < m) ? row[base + e] : 0; s += c[e]; }
  int total;
  int run = block_exscan_1024(s, wtot, total) + bsum[blockIdx.y * gridDim.x + blockIdx.x];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (base + e < m) {
      row[base + e] = run;
      if (copy) copy[(long)blockIdx.y * m + base + e] = run;
    }
    run += c[e];
  }
}

inline void exscan_rows(hipStream_t st, int rows, int m, int32_t *v, int32_t *bsum, int32_t *copy = nullptr) {
  const int nb = scan_blocks(m);
  scan_block_sums_kernel<<<dim3(nb, rows), 1024, 0, st>>>(m, v, bsum);
  scan_block_offsets_kernel<<<rows, 1024, 0, st>>>(nb, bsum);
  scan_apply_kernel<<<dim3(nb, rows), 1024, 0, st>>>(m, v, bsum, copy);
}

// Zero up to four device spans with as few hipMemsetAsync calls as possible: spans that are adjacent in memory
// (the Python side allocates the small per-call accumulators back to back) collapse into one fill -- a fill of a few
// hundred bytes costs ~4.5 us of GPU time like any other launch, and a training step had ~80 of them.
// Pre-zeroed scratch arena (gcn_zero_arena_register): a span that lies inside it was zeroed by the owner of the arena
// with ONE fill at the start of the step and has not been handed out since -- the fill is skipped.  Any other pointer
// is zeroed here as before.
// Device fills go through a KERNEL, never hipMemsetAsync: inside a captured HIP graph a memset NODE is what makes ROCm
// 7.2's packet-capture replay path fault when the graph is replayed after the queue went idle (tools/debug/
// graph_trigger5.py: a graph holding this library's kNN entry point faulted on its second replay 3 runs out of 4 -- the
// stale, un-zeroed counter behind it sent knnf_list_kernel to a wild address -- and never with the fill as a kernel
// node; graphs of plain kernels, 3000 nodes long, replay fine).  Word fills when pointer and size allow, bytes otherwise.
static __global__ __launch_bounds__(256) void fill_words_kernel(unsigned int *__restrict__ p, unsigned int v, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
    if (i + 4 <= nwords && (((uintptr_t)(p + i)) & 15) == 0) {
      *reinterpret_cast<uint4 *>(p + i) = make_uint4(v, v, v, v);
    } else {
      for (size_t e = i; e < nwords && e < i + 4; ++e) p[e] = v;
    }
  }
}
static __global__ __launch_bounds__(256) void fill_bytes_kernel(unsigned char *__restrict__ p, unsigned char v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
inline hipError_t fill_dev(void *p, int value, size_t n, hipStream_t st) {
  if (n == 0 || !p) return hipSuccess;
  const unsigned int b = (unsigned int)value & 0xffu;
  if ((((uintptr_t)p) & 3) == 0 && (n & 3) == 0) {
    const size_t nwords = n / 4;
    size_t blocks = (nwords + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    fill_words_kernel<<<(unsigned int)blocks, 256, 0, st>>>((unsigned int *)p, b * 0x01010101u, nwords);
  } else {
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    fill_bytes_kernel<<<(unsigned int)blocks, 256, 0, st>>>((unsigned char *)p, (unsigned char)b, n);
  }
  return hipGetLastError();
}

extern char *g_zero_lo, *g_zero_hi;
inline hipError_t zero_dev(void *p, size_t n, hipStream_t st) {
  if (n == 0 || !p) return hipSuccess;
  if ((char *)p >= g_zero_lo && (char *)p + n <= g_zero_hi) return hipSuccess;
  return fill_dev(p, 0, n, st);
}

struct ZeroSpan { void *p; size_t n; };
inline hipError_t zero_spans(hipStream_t st, ZeroSpan a, ZeroSpan b = {nullptr, 0}, ZeroSpan c = {nullptr, 0},
                             ZeroSpan d = {nullptr, 0}) {
  constexpr int NS = 4;
  ZeroSpan v[NS] = {a, b, c, d};
  // sort by address, skip empties
  for (int i = 0; i < NS; ++i)
    for (int j = i + 1; j < NS; ++j)
      if (v[j].p && (!v[i].p || (char *)v[j].p < (char *)v[i].p)) { ZeroSpan t = v[i]; v[i] = v[j]; v[j] = t; }
  int i = 0;
  while (i < NS && v[i].p) {
    char *lo = (char *)v[i].p, *hi = lo + v[i].n;
    int j = i + 1;
    while (j < NS && v[j].p && (char *)v[j].p <= hi) {            // exactly adjacent (or overlapping) spans only
      char *h2 = (char *)v[j].p + v[j].n;
      if (h2 > hi) hi = h2;
      ++j;
    }
    hipError_t e = zero_dev(lo, (size_t)(hi - lo), st);
    if (e != hipSuccess) return e;
    i = j;
  }
  return hipSuccess;
}


// Fold per-workgroup partial sums in a FIXED order (deterministic, no same-address atomics):
//   part_c (K, 2C) f32 -> out_a[0..C) = column sums of the first C columns, out_b[0..C) of the last C   (K = B * nblk)
//   part_s (B, nblk, 2G) f64 -> S (B, 2G)
// Workgroup = 64 outputs x 4 waves, each wave walks a quarter of the partials sixteen loads at a time; the four
// quarter sums meet in LDS.  grid = cdiv(2C, 64) + cdiv(B*2G, 64).
static __global__ __launch_bounds__(256) void fold_partials_kernel(const double *__restrict__ part_s, const float *__restrict__ part_c,
                                                                   int nblk, int B, int C, int G, double *__restrict__ S,
                                                                   float *__restrict__ out_a, float *__restrict__ out_b) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cblocks = (2 * C + 63) / 64;
  if ((int)blockIdx.x < cblocks) {
    const int t = blockIdx.x * 64 + lane;
    const long K = (long)B * nblk;
    const long k0 = K * wave / 4, k1 = K * (wave + 1) / 4;
    float acc = 0.f;
    if (t < 2 * C) {
      long k = k0;
      for (; k + 16 <= k1; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part_c[(k + u) * 2 * C + t];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
      for (; k < k1; ++k) acc += part_c[k * 2 * C + t];
    }
    red[wave][lane] = (double)acc;
    __syncthreads();
    if (wave == 0 && t < 2 * C) {
      const float r = (((float)red[0][lane] + (float)red[1][lane]) + (float)red[2][lane]) + (float)red[3][lane];
      if (t < C) out_a[t] = r; else out_b[t - C] = r;
    }
  } else {
    const int i = ((int)blockIdx.x - cblocks) * 64 + lane;        // (b, j) pair
    const int b = i / (2 * G), j = i % (2 * G);
    const int k0 = nblk * wave / 4, k1 = nblk * (wave + 1) / 4;
    double acc = 0.0;
    if (i < B * 2 * G)
      for (int k = k0; k < k1; ++k) acc += part_s[((long)b * nblk + k) * 2 * G + j];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < B * 2 * G) S[(long)b * 2 * G + j] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
  }
}

}  // namespace gcn
