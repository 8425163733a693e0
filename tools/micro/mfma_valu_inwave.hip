// Within ONE wave: does independent VALU work issued between dependent MFMAs hide behind them (gfx950)?
// Each wave runs a chain of MFMAs with F independent v_fma_f32 after every MFMA, F = 0, 2, 4, 6, 8; 1 and 2 waves/SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_valu_inwave.hip -o /tmp/iw && /tmp/iw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) short bf8;

template <int BF16, int F>
__global__ __launch_bounds__(256) void k(int iters, float *out) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  float r;
  if (BF16) {
    f16v acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    bf8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + (threadIdx.x & 3)); b[j] = 0x3f00; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < F; ++f) x[f] = fmaf(x[f], 1.0001f, 0.5f);
      }
    r = acc[0] + acc[15];
  } else {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const float a = 1.f + (threadIdx.x & 3), b = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < F; ++f) x[f] = fmaf(x[f], 1.0001f, 0.5f);
      }
    r = acc[0] + acc[3];
  }
  for (int i = 0; i < 8; ++i) r += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int BF16, int F>
static void run(int wps, float *out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<BF16, F><<<256 * wps, 256>>>(16, out);
  hipEventRecord(e0);
  k<BF16, F><<<256 * wps, 256>>>(8192, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s  waves/SIMD=%d  fillers/MFMA=%d : %.3f ms  (%.1f cycles per MFMA per SIMD @2.4GHz)\n", BF16 ? "bf16 32x32x16" : "f32 16x16x4  ", wps, F, ms,
         ms * 1e-3 * 2.4e9 / (8192.0 * 8 * wps));
}

int main() {
  float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int wps = 1; wps <= 2; ++wps) {
    run<0, 0>(wps, out); run<0, 2>(wps, out); run<0, 4>(wps, out); run<0, 6>(wps, out); run<0, 8>(wps, out);
  }
  for (int wps = 1; wps <= 2; ++wps) {
    run<1, 0>(wps, out); run<1, 2>(wps, out); run<1, 4>(wps, out); run<1, 6>(wps, out); run<1, 8>(wps, out);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
