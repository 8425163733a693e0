// What does __builtin_amdgcn_permlane32_swap return, and does hipcc keep both results?
// hipcc --offload-arch=gfx950 -O3 [-fno-honor-nans] -o pst permlane_swap_test.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ float both_max(float v) {
  const unsigned int u = __builtin_bit_cast(unsigned int, v);
  unsigned int w = u;
  asm volatile("" : "+v"(w));
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
#ifdef LAUNDER_RESULT
  unsigned int r1 = r[1];
  asm volatile("" : "+v"(r1));      // without this hipcc (ROCm 7.2) emits v_max_f32 v, r[0] only: 16 of 64 lanes wrong
  return __builtin_fmaxf(__builtin_fmaxf(v, __builtin_bit_cast(float, r[0])), __builtin_bit_cast(float, r1));
#else
  return __builtin_fmaxf(__builtin_fmaxf(v, __builtin_bit_cast(float, r[0])), __builtin_bit_cast(float, r[1]));
#endif
}
__device__ __forceinline__ int both_min_i(int v) {
  unsigned int w = (unsigned int)v;
  asm volatile("" : "+v"(w));
  const u32x2 r = __builtin_amdgcn_permlane32_swap((unsigned int)v, w, false, false);
  return min(min(v, (int)r[0]), (int)r[1]);
}
__global__ void k(unsigned int *out, const float *in) {
  const unsigned int l = threadIdx.x;
  const unsigned int v = 1000 + l;
  const u32x2 r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  out[l] = r[0];
  out[64 + l] = r[1];
  const float f = in[l];
  reinterpret_cast<float *>(out)[128 + l] = both_max(f);
  out[192 + l] = (unsigned int)both_min_i((int)in[64 + l]);
}
int main() {
  unsigned int *d, h[256];
  float *din, hin[128];
  for (int i = 0; i < 64; ++i) { hin[i] = (float)((i * 37) % 64); hin[64 + i] = (float)((i * 29 + 5) % 64); }
  hipMalloc(&d, sizeof(h));
  hipMalloc(&din, sizeof(hin));
  hipMemcpy(din, hin, sizeof(hin), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, din);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("swap(v,v)[0] lane0 %u lane32 %u ; [1] lane0 %u lane32 %u\n", h[0], h[32], h[64], h[96]);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const float want = hin[i] > hin[i ^ 32] ? hin[i] : hin[i ^ 32];
    const float got = reinterpret_cast<float *>(h)[128 + i];
    const int wmin = (int)(hin[64 + i] < hin[64 + (i ^ 32)] ? hin[64 + i] : hin[64 + (i ^ 32)]);
    if (got != want || (int)h[192 + i] != wmin) { ++bad; if (bad < 5) printf("lane %d: max got %g want %g ; min got %d want %d\n", i, got, want, (int)h[192 + i], wmin); }
  }
  printf("both_halves reductions: %d bad lanes of 64\n", bad);
  return bad != 0;
}
