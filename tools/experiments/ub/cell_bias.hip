// one LSTM cell update from identical inputs: the persistent kernels' formulas (fast_sigmoid / fast_tanh: v_rcp_f32) and the
// step kernels' (true division), against float64 — mean RELATIVE error of c and h.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -w tools/experiments/ub/cell_bias.hip -o /tmp/cell_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
__device__ __forceinline__ float rcp_nr(float d) {     // v_rcp_f32 + one Newton step: r + r (1 - d r)
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float nsig(float x) { return rcp_nr(1.0f + __expf(-x)); }
__device__ __forceinline__ float n2tanh(float x) {
  const float e = __expf(-2.0f * __builtin_amdgcn_fmed3f(x, -30.0f, 30.0f));
  return (1.0f - e) * rcp_nr(1.0f + e);
}
__device__ __forceinline__ float ntanh(float x) {      // (1 - e) / (1 + e), e = exp(-2x): no cancellation behind the reciprocal
  const float e = __expf(-2.0f * __builtin_amdgcn_fmed3f(x, -30.0f, 30.0f));
  return (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float dsig(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dtanh(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }
__global__ void k(const float *z, const float *cp, float *o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float zi = z[4 * i], zj = z[4 * i + 1], zf = z[4 * i + 2], zo = z[4 * i + 3], c0 = cp[i];
  { const float gi = fsig(zi), gj = ftanh(zj), gf = fsig(zf + 1.0f), go = fsig(zo);
    const float c = c0 * gf + gi * gj; o[i] = c; o[n + i] = ftanh(c) * go; }
  { const float gi = dsig(zi), gj = dtanh(zj), gf = dsig(zf + 1.0f), go = dsig(zo);
    const float c = c0 * gf + gi * gj; o[2 * n + i] = c; o[3 * n + i] = dtanh(c) * go; }
  { const float gi = nsig(zi), gj = n2tanh(zj), gf = nsig(zf + 1.0f), go = nsig(zo);
    const float c = c0 * gf + gi * gj; o[4 * n + i] = c; o[5 * n + i] = n2tanh(c) * go; }
}
int main() {
  const int n = 1 << 20;
  std::vector<float> z(4 * n), cp(n), o(6 * n);
  srand(3);
  auto g = []() { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
  for (auto &v : z) v = (float)(1.5 * g());
  for (auto &v : cp) v = (float)(0.7 * g());
  float *dz, *dc, *dob; hipMalloc(&dz, 4 * n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dob, 6 * n * 4);
  hipMemcpy(dz, z.data(), 4 * n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, cp.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dz, dc, dob, n);
  hipMemcpy(o.data(), dob, 6 * n * 4, hipMemcpyDeviceToHost);
  double s[6] = {0, 0, 0, 0, 0, 0}, q[6] = {0, 0, 0, 0, 0, 0}; long cnt = 0;
  for (int i = 0; i < n; ++i) {
    const double zi = z[4 * i], zj = z[4 * i + 1], zf = z[4 * i + 2], zo = z[4 * i + 3], c0 = cp[i];
    const double c = c0 / (1 + exp(-(zf + 1.0))) + tanh(zj) / (1 + exp(-zi)), h = tanh(c) / (1 + exp(-zo));
    if (fabs(h) < 1e-3 || fabs(c) < 1e-3) continue;
    const double r[6] = {(o[i] - c) / c, (o[n + i] - h) / h, (o[2 * n + i] - c) / c, (o[3 * n + i] - h) / h,
                         (o[4 * n + i] - c) / c, (o[5 * n + i] - h) / h};
    for (int j = 0; j < 6; ++j) { s[j] += r[j]; q[j] += r[j] * r[j]; }
    ++cnt;
  }
  printf("persistent formulas: c mean rel %+.3e (rms %.2e), h mean rel %+.3e (rms %.2e)\n", s[0] / cnt, sqrt(q[0] / cnt), s[1] / cnt, sqrt(q[1] / cnt));
  printf("step-kernel formulas: c mean rel %+.3e (rms %.2e), h mean rel %+.3e (rms %.2e)\n", s[2] / cnt, sqrt(q[2] / cnt), s[3] / cnt, sqrt(q[3] / cnt));
  printf("rcp + Newton step:   c mean rel %+.3e (rms %.2e), h mean rel %+.3e (rms %.2e)\n", s[4] / cnt, sqrt(q[4] / cnt), s[5] / cnt, sqrt(q[5] / cnt));
  return 0;
}
