// Do f32 MFMA chains of some waves overlap with plain VALU work of OTHER waves on the same SIMD (gfx950)?
// Workgroup = 8 waves (2 per SIMD): waves 0-3 run only v_mfma_f32_16x16x4_f32, waves 4-7 only v_fma_f32 chains.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) short bf8;

template <int BF16>
__global__ __launch_bounds__(512) void k(int mfma_iters, int valu_iters, float *out) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  if (wave < 4) {
    if (BF16) {
      f16v acc;
      for (int j = 0; j < 16; ++j) acc[j] = 0.f;
      bf8 a, b;
      for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + (threadIdx.x & 3)); b[j] = 0x3f00; }
      for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      r = acc[0] + acc[15];
    } else {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float a = 1.f + (threadIdx.x & 3), b = 0.5f;
      for (int it = 0; it < mfma_iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
      r = acc[0] + acc[3];
    }
  } else {
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int it = 0; it < valu_iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 1.f);
      }
    r = x0 + x1 + x2 + x3;
  }
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int BF16>
static float run(int mi, int vi, float *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<BF16><<<256, 512>>>(mi ? 16 : 0, vi ? 16 : 0, out);
  hipEventRecord(e0);
  k<BF16><<<256, 512>>>(mi, vi, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float *out; hipMalloc(&out, 256 * 512 * 4);
  const int MI = 4096, VI = 2048;     // 65536 MFMAs per MFMA wave; 131072 VALU per VALU wave
  {
    const float tm = run<0>(MI, 0, out), tv = run<0>(0, VI, out), tb = run<0>(MI, VI, out);
    printf("f32 16x16x4  : MFMA waves only %.3f ms | VALU waves only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
  }
  {
    const float tm = run<1>(MI, 0, out), tv = run<1>(0, 4 * VI, out), tb = run<1>(MI, 4 * VI, out);
    printf("bf16 32x32x16: MFMA waves only %.3f ms | VALU waves only %.3f ms | both %.3f ms  (sum %.3f, max %.3f)\n", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
