// Issue rate of v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 in ONE dependent chain per wave, at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void k(int iters, float *out) {
  const float a = 1.0f + (threadIdx.x & 3), b = 0.5f;
  if (SHAPE == 16) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % NACC], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int SHAPE, int NACC>
void run(const char *name, int wgs_per_cu, float *out) {
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<SHAPE, NACC><<<256 * wgs_per_cu, 256>>>(16, out);
  hipEventRecord(e0);
  k<SHAPE, NACC><<<256 * wgs_per_cu, 256>>>(iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)wgs_per_cu * iters * 16;             // MFMAs per SIMD
  const double flop = (SHAPE == 16 ? 2048.0 : 4096.0) * mf * 1024;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f cycles/MFMA/SIMD @2.4GHz  %.1f TFLOP/s\n", name, wgs_per_cu, ms,
         ms * 1e-3 * 2.4e9 / mf, flop / ms / 1e9);
}

int main() {
  float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
  for (int w = 1; w <= 4; ++w) run<16, 1>("16x16x4 f32, 1 chain", w, out);
  for (int w = 1; w <= 4; w *= 2) run<16, 4>("16x16x4 f32, 4 accumulators", w, out);
  for (int w = 1; w <= 2; ++w) run<32, 1>("32x32x2 f32, 1 chain", w, out);
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
