#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  float w[32]; for (int j = 0; j < 32; ++j) w[j] = in[j + threadIdx.x % 7];
  float h[4]; for (int j = 0; j < 4; ++j) h[j] = in[40 + j + threadIdx.x % 3];
  f32x4 acc[NACC]; for (int a = 0; a < NACC; ++a) acc[a] = (f32x4){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      acc[j % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(h[j & 3], w[j], acc[j % NACC], 0, 0, 0);
    asm volatile("" :: "v"(acc[0].x));
  }
  float s = 0; for (int a = 0; a < NACC; ++a) s += acc[a].x + acc[a].y + acc[a].z + acc[a].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// correctness: D_b[i][j] = sum_k A_b[i][k] B_b[k][j]; lane = 4b + (i for A | j for B, D), D regs = i
__global__ void chk(float* out) {
  const int l = threadIdx.x, b = l >> 2, r = l & 3;
  f32x4 acc = {0, 0, 0, 0};
  float a = 10.f * b + r + 1;        // A_b[i=r]
  float bb = 100.f * b + 2 * r + 1;  // B_b[j=r]
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bb, acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}
int main() {
  float *out, *in; hipMalloc(&out, 256*512*4*2); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipLaunchKernelGGL(chk, dim3(1), dim3(64), 0, 0, out); hipDeviceSynchronize();
  float h[256]; hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    int b = l >> 2, j = l & 3; float want = (10.f * b + i + 1) * (100.f * b + 2 * j + 1);
    if (h[l * 4 + i] != want) { if (bad < 4) printf("mismatch lane %d reg %d got %g want %g\n", l, i, h[l*4+i], want); ++bad; }
  }
  printf("layout check (D_b[i][j]: lane=4b+j, reg=i; A lane=4b+i; B lane=4b+j): %s\n", bad ? "FAILED" : "ok");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int thr : {256, 512}) for (int nacc : {1, 2, 4}) {
    int iters = 4000;
    auto run = [&]() { if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, out, in, iters);
                       else if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(thr), 0, 0, out, in, iters);
                       else hipLaunchKernelGGL(k<4>, dim3(256), dim3(thr), 0, 0, out, in, iters); };
    run(); hipDeviceSynchronize();
    hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_per_simd = (double)iters * 32 * (thr / 256);
    printf("threads %d accumulators %d: %.3f ms, %.2f ns per MFMA per SIMD, %.1f TFLOP/s\n", thr, nacc, ms, ms * 1e6 / n_per_simd, 256.0 * thr / 64 * iters * 32 * 512 / ms / 1e9);
  }
  return 0;
}
