// microbenchmark: fp32 MFMA (32x32x2) interleaved with N independent v_fma_f32 per MFMA.
// Do the two fp32 pipes (matrix 64 FLOP/clk/SIMD, VALU 64 FLOP/clk/SIMD) run concurrently?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NF>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[m], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int idx = (m * NF + f) & 15;
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[idx]) : "v"(x), "v"(y));
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NF>
void run(int wg_per_cu, float *out) {
  const int iters = 4000, grid = 256 * wg_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NF>, dim3(grid), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NF>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)grid * 4;
  const double mf = waves * iters * 4 * 4096.0, vf = waves * iters * 4 * NF * 128.0;
  printf("wg/cu %d  NF %2d: %.3f ms  mfma %.1f TF/s  valu %.1f TF/s  total %.1f TF/s  (cycles per MFMA per SIMD @2.4GHz: %.1f)\n",
         wg_per_cu, NF, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9,
         ms * 1e-3 * 2.4e9 / (iters * 4.0 * wg_per_cu));
}
int main() {
  float *out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int w = 1; w <= 3; ++w) {
    run<0>(w, out); run<2>(w, out); run<4>(w, out); run<8>(w, out); run<12>(w, out); run<16>(w, out); run<24>(w, out); run<32>(w, out);
  }
  return 0;
}
