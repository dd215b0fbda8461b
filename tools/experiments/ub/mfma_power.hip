// fp32 MFMA peak with constant vs random operand data (power / DVFS effect)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x + 256 * i) & 4095]; b[i] = in[(threadIdx.x * 7 + 131 * i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 7], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 3) & 7], b[j], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 5) & 7], b[(j + 2) & 7], acc[3], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 3 * 256 * 4);
  float h[4096];
  for (int it2 : {300, 1000, 3000, 10000, 30000, 300, 300}) {
    for (int i = 0; i < 4096; ++i) h[i] = (float)(rand() % 2000 - 1000) / 1000.f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(768), dim3(256), 0, 0, in, out, it2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("iters %5d: %.2f ms  %.1f TF/s\n", it2, ms, (double)768 * 4 * it2 * 32 * 4096.0 / ms / 1e9);
  }
  for (int mode = 0; mode < 1; ++mode) {
    for (int i = 0; i < 4096; ++i) h[i] = mode == 0 ? 1.0f : mode == 1 ? (float)(rand() % 2000 - 1000) / 1000.f : ((float)rand() / RAND_MAX - 0.5f) * 1e-3f * (1 + rand() % 1000);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
      const int iters = 20000, grid = 256 * 3;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s) rep %d: %.2f ms  %.1f TF/s\n", mode, mode == 0 ? "all ones" : mode == 1 ? "random 3 digits" : "random full mantissa", rep, ms,
             (double)grid * 4 * iters * 32 * 4096.0 / ms / 1e9);
    }
  }
  return 0;
}
