// cast.hip -- ONE launch for the per-step f32 -> bf16 refresh of every parameter copy (gcanet_amd/layers.py:CastCache).
// torch._foreach_copy_ between different dtypes falls back to one copy kernel per tensor (57 launches, 0.2 ms per
// training step of the hot path); here a device-resident segment table describes all tensors and one grid walks it.
#include "common.h"

namespace gcn {

struct CastSeg {            // one workgroup's share: elements [first, first + count) of a row-major (rows, cols) f32 tensor
  const float *src;
  unsigned short *dst;      // bf16 image with row pitch `pitch` elements (>= cols; padded weight images)
  long cols, pitch, first, count;
};

__global__ __launch_bounds__(256) void multi_cast_bf16_kernel(const CastSeg *__restrict__ segs) {
  const CastSeg s = segs[blockIdx.x];
  const bool vec = (s.cols % 4 == 0) && (s.pitch % 4 == 0) && (s.first % 4 == 0) && (((uintptr_t)s.src & 15) == 0) &&
                   (((uintptr_t)s.dst & 7) == 0);
  if (vec) {
    for (long e = s.first + (long)threadIdx.x * 4; e < s.first + s.count; e += 256 * 4) {
      if (e + 4 <= s.first + s.count) {
        const float4 v = *reinterpret_cast<const float4 *>(s.src + e);
        const long row = e / s.cols, col = e - row * s.cols;
        const __bf16 h0 = (__bf16)v.x, h1 = (__bf16)v.y, h2 = (__bf16)v.z, h3 = (__bf16)v.w;
        uint2 o;
        o.x = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
        o.y = (unsigned int)__builtin_bit_cast(unsigned short, h2) | ((unsigned int)__builtin_bit_cast(unsigned short, h3) << 16);
        *reinterpret_cast<uint2 *>(s.dst + row * s.pitch + col) = o;
      } else {
        for (long t = e; t < s.first + s.count; ++t) {
          const long row = t / s.cols, col = t - row * s.cols;
          const __bf16 h = (__bf16)s.src[t];
          s.dst[row * s.pitch + col] = __builtin_bit_cast(unsigned short, h);
        }
      }
    }
  } else {
    for (long e = s.first + threadIdx.x; e < s.first + s.count; e += 256) {
      const long row = e / s.cols, col = e - row * s.cols;
      const __bf16 h = (__bf16)s.src[e];
      s.dst[row * s.pitch + col] = __builtin_bit_cast(unsigned short, h);
    }
  }
}

}  // namespace gcn

using namespace gcn;

GCN_EXPORT int gcn_multi_cast_bf16(const void *segs_dev, int nseg, void *stream) {
  GCN_REQUIRE(segs_dev || nseg == 0, "gcn_multi_cast_bf16: null segment table");
  GCN_REQUIRE(nseg >= 0, "gcn_multi_cast_bf16: bad segment count");
  if (nseg == 0) return GCN_OK;
  multi_cast_bf16_kernel<<<nseg, 256, 0, (hipStream_t)stream>>>((const CastSeg *)segs_dev);
  return check_launch("multi_cast_bf16_kernel");
}
