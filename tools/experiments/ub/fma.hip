#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
  f32x2 acc[8][2];
  f32x2 W[16][2];
  for (int j = 0; j < 16; ++j) { W[j][0] = (f32x2){in[j], in[j+16]}; W[j][1] = (f32x2){in[j+32], in[j+48]}; }
  for (int b = 0; b < 8; ++b) acc[b][0] = acc[b][1] = (f32x2){0.f, 0.f};
  float hb[8]; for (int b = 0; b < 8; ++b) hb[b] = in[64 + b + threadIdx.x % 3];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (MODE == 0) {
          const f32x2 hh = {hb[b], hb[b]};
          acc[b][0] = __builtin_elementwise_fma(hh, W[j][0], acc[b][0]);
          acc[b][1] = __builtin_elementwise_fma(hh, W[j][1], acc[b][1]);
        } else if (MODE == 1) {   // scalar fmas
          acc[b][0].x = fmaf(hb[b], W[j][0].x, acc[b][0].x); acc[b][0].y = fmaf(hb[b], W[j][0].y, acc[b][0].y);
          acc[b][1].x = fmaf(hb[b], W[j][1].x, acc[b][1].x); acc[b][1].y = fmaf(hb[b], W[j][1].y, acc[b][1].y);
        } else {                  // packed with pair operands (no broadcast): rows packed
          const f32x2 hh = {hb[b], hb[(b+1)&7]};
          acc[b][0] = __builtin_elementwise_fma(hh, W[j][0], acc[b][0]);
          acc[b][1] = __builtin_elementwise_fma(hh, W[j][1], acc[b][1]);
        }
      }
    }
    asm volatile("" :: "v"(acc[0][0].x));
  }
  float s = 0; for (int b = 0; b < 8; ++b) s += acc[b][0].x + acc[b][0].y + acc[b][1].x + acc[b][1].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float *out, *in; hipMalloc(&out, 256*512*4*2); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int thr : {256, 512}) for (int mode = 0; mode < 3; ++mode) {
    int iters = 2000;
    auto run = [&](){ if (mode==0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(thr), 0, 0, out, in, iters);
                      else if (mode==1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(thr), 0, 0, out, in, iters);
                      else hipLaunchKernelGGL(k<2>, dim3(256), dim3(thr), 0, 0, out, in, iters); };
    run(); hipDeviceSynchronize();
    hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fmas = 256.0 * thr * iters * 16 * 8 * 4;   // lane-FMAs
    double per_simd_waves = thr / 256.0;               // waves per SIMD
    double ns_per_inst = ms * 1e6 / (iters * 16.0 * 8 * (mode==1 ? 4 : 2) * per_simd_waves);
    printf("threads %d mode %d: %.3f ms  %.1f TFLOP/s  %.2f ns per wave-instruction per SIMD\n", thr, mode, ms, 2*fmas/ms/1e9, ns_per_inst);
  }
  return 0;
}
