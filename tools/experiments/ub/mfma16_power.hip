// Sustained rate of a bare v_mfma_f32_32x32x16_f16 loop with random operands, by duration (the chip clocks the matrix pipe
// to its power budget): the ceiling the packed-operand GEMM's 1.2 PF/s (74 % busy at 1.68 GHz) is to be read against.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_power tools/experiments/ub/mfma16_power.hip && /tmp/mfma16_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(const _Float16 *in, float *out, int iters, unsigned long long *clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) { a[i][e] = in[(threadIdx.x * 8 + e + 2048 * i) & 16383]; b[i][e] = in[(threadIdx.x * 56 + 131 * i + e + 8192) & 16383]; }
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(j + i) & 3], b[(j + 2 * i + 1) & 3], acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  _Float16 *in; float *out; unsigned long long *clk;
  hipMalloc(&in, 16384 * 2); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&clk, 16);
  _Float16 h[16384];
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 16384; ++i) h[i] = mode ? (_Float16)((float)(rand() % 2000 - 1000) / 1000.f) : (_Float16)1.0f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int wg : {256, 512}) {
      for (int iters : {2000, 20000, 100000, 2000}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<4>, dim3(wg), dim3(256), 0, 0, in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)wg * 4 /*waves*/ * iters * 16 /*mfma*/ * 2.0 * 32 * 32 * 16;
        printf("%s operands, %d workgroups x 256 threads, %7d iterations: %8.3f ms  %7.1f TF/s   shader clock %.3f GHz\n",
               mode ? "random  " : "constant", wg, iters, ms, flops / ms / 1e9, (double)c[0] / (c[1] * 10.0));
      }
    }
  }
  return 0;
}
