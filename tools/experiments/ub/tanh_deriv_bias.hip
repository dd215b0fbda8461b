// bias of 1 - g^2 computed from the fp32 value g = tanh(z) each formula stores (the backward pass's tanh'), against float64.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -w tools/experiments/ub/tanh_deriv_bias.hip -o /tmp/tdb
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__device__ __forceinline__ float rcp_nr(float d) { const float r = __builtin_amdgcn_rcpf(d); return fmaf(fmaf(-d, r, 1.0f), r, r); }
__global__ void k(const float *z, float *o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = z[i];
  o[i] = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f;              // round-4 fast_tanh
  o[n + i] = 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f;                               // step kernels
  { const float e = __expf(-2.0f * x); o[2 * n + i] = (1.0f - e) * rcp_nr(1.0f + e); }   // round-5 fast_tanh
  { const float a = fabsf(x), e = __expf(-2.0f * a); o[3 * n + i] = copysignf((1.0f - e) * rcp_nr(1.0f + e), x); }   // odd
}
int main() {
  const int n = 1 << 22;
  std::vector<float> z(n), o(4 * (size_t)n);
  srand(5);
  auto g = []() { double s = 0; for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX; return s - 6.0; };
  for (auto &v : z) v = (float)(2.0 * g());
  float *dz, *dob; hipMalloc(&dz, n * 4); hipMalloc(&dob, 4 * (size_t)n * 4);
  hipMemcpy(dz, z.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dz, dob, n);
  hipMemcpy(o.data(), dob, 4 * (size_t)n * 4, hipMemcpyDeviceToHost);
  const char *nm[4] = {"2 rcp - 1 (round 4)", "2 / (1 + e) - 1 (step)", "(1-e) rcp_nr(1+e)", "odd: on |x|, copysign"};
  for (int f = 0; f < 4; ++f) {
    double sv = 0, sd = 0, sdd = 0, ssym = 0;
    for (int i = 0; i < n; ++i) {
      const double t = tanh((double)z[i]), gv = o[(size_t)f * n + i];
      sv += (gv - t) * (t > 0 ? 1 : -1);                 // error of the value, signed away from zero
      const double d = (1.0 - gv * gv) - (1.0 - t * t);  // error of the derivative
      sd += d; sdd += d * d;
    }
    // odd symmetry: g(-x) == -g(x)?
    printf("%-24s value error away-from-zero mean %+.3e | derivative error mean %+.3e rms %.3e\n", nm[f], sv / n, sd / n, sqrt(sdd / n));
  }
  return 0;
}
