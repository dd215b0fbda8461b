// fp32 MFMA issue rate vs operand register pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, int iters, unsigned long long *clk) {
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x + 256 * i) & 4095]; b[i] = in[(threadIdx.x * 7 + 131 * i + 2048) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) {          // same A and B for every MFMA
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[3], 0, 0, 0);
      } else if (MODE == 1) {   // GEMM pattern: 2 A x 2 B per k-step, new registers every k-step
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 7], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[(j + 1) & 7], acc[3], 0, 0, 0);
      } else if (MODE == 2) {   // A changes, B fixed
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 7], b[0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 2) & 7], b[0], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 3) & 7], b[0], acc[3], 0, 0, 0);
      } else {                  // 16x16x4 form, GEMM pattern
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc[3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = wall_clock64() - w0; }
}
unsigned long long *clk;
template <int MODE> void run(const float *in, float *out, int wg) {
  for (int it2 : {4000}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, in, out, 10, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, in, out, it2, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    printf("mode %d wg/cu %d iters %5d: %.2f ms  %.1f TF/s | block 0: %.1f shader cycles per MFMA per SIMD-slot, shader clock %.2f GHz\n", MODE, wg, it2, ms,
           (double)256 * wg * 4 * it2 * 32 * 4096.0 / ms / 1e9, (double)hc[0] / (it2 * 32.0) / wg, (double)hc[0] / (hc[1] * 10.0));
  }
}
int main() {
  float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 3 * 256 * 4);
  hipMalloc(&clk, 16);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 2000 - 1000) / 1000.f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) for (int wg = 2; wg <= 3; ++wg) { run<0>(in, out, wg); run<1>(in, out, wg); run<2>(in, out, wg); run<3>(in, out, wg); }
  return 0;
}
