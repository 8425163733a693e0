// LDS atomic throughput on gfx950: ds_add_f32 vs ds_add_u32 vs ds_add_u64 vs plain read-modify-write.
// hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_bench.hip -o /tmp/lds_atomic_bench && /tmp/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(int iters, int rows, const int *__restrict__ rowid, float *out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < rows * 64 * 2; i += 1024) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const int r = rowid[(it * 16 + wave) & 4095] % rows;   // wave-uniform pseudo-random row
    const float v = 1.0f + lane;
    if (MODE == 0) atomicAdd(&lds[r * 64 + lane], v);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned int *>(lds) + r * 64 + lane, (unsigned int)lane);
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long *>(lds) + r * 64 + lane, (unsigned long long)lane);
    if (MODE == 3) lds[r * 64 + lane] += v;                // racy plain RMW (rate reference only)
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x];
}

int main() {
  int *rowid; float *out;
  int h[4096];
  unsigned s = 12345;
  for (int i = 0; i < 4096; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 8) & 0xffff; }
  hipMalloc(&rowid, sizeof(h)); hipMemcpy(rowid, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&out, 256 * 64 * 4);
  const int iters = 4096, rows = 256;
  const size_t ldsb = rows * 64 * 2 * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
  for (int m = 0; m < 4; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m == 0) { hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); k<0><<<256, 1024, ldsb>>>(iters, rows, rowid, out); }
      if (m == 1) { hipFuncSetAttribute((const void *)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); k<1><<<256, 1024, ldsb>>>(iters, rows, rowid, out); }
      if (m == 2) { hipFuncSetAttribute((const void *)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); k<2><<<256, 1024, ldsb>>>(iters, rows, rowid, out); }
      if (m == 3) { hipFuncSetAttribute((const void *)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); k<3><<<256, 1024, ldsb>>>(iters, rows, rowid, out); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-12s %.3f ms  -> %.1f G lane-ops/s chip, %.2f lane-ops/clk/CU (2.4 GHz)\n", names[m], ms,
                      256.0 * 1024 * iters / ms / 1e6, 1024.0 * iters / (ms * 1e-3 * 2.4e9));
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
