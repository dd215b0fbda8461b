// mean signed error (in ulp) of v_rcp_f32 / v_exp_f32 based sigmoid and tanh as the persistent kernels compute them
// (fast_sigmoid, fast_tanh of lstm_persist_dev.h) against the step kernels' forms (true division) and float64.
// build: hipcc --offload-arch=gfx950 -O2 -w tools/experiments/ub/rcp_bias.hip -o /tmp/rcp_bias && /tmp/rcp_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__global__ void k(const float *x, float *o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  o[i] = __builtin_amdgcn_rcpf(v);
  o[n + i] = __builtin_amdgcn_rcpf(1.0f + __expf(-v));                       // fast_sigmoid
  o[2 * n + i] = 1.0f / (1.0f + __expf(-v));                                 // sigmoidf_
  o[3 * n + i] = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * v)) - 1.0f;   // fast_tanh
  o[4 * n + i] = 2.0f / (1.0f + __expf(-2.0f * v)) - 1.0f;                   // tanhf_
}
int main() {
  const int n = 1 << 20;
  {   // v_rcp_f32 by input range
    std::vector<float> x(n), o(5 * n);
    float *dx, *dob; hipMalloc(&dx, n * 4); hipMalloc(&dob, 5 * n * 4);
    const double lo[5] = {1.0, 1.0, 1.1, 1.5, 1.0}, hi[5] = {1.01, 1.1, 1.5, 2.0, 1.0001};
    for (int r = 0; r < 5; ++r) {
      for (int i = 0; i < n; ++i) x[i] = (float)(lo[r] + (hi[r] - lo[r]) * i / n);
      hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dob, n);
      hipMemcpy(o.data(), dob, 5 * n * 4, hipMemcpyDeviceToHost);
      double s = 0, sq = 0;
      for (int i = 0; i < n; ++i) {
        const double ref = 1.0 / (double)x[i];
        int ex; frexp(ref, &ex);
        const double e = (o[i] - ref) / ldexp(1.0, ex - 24);
        s += e; sq += e * e;
      }
      printf("v_rcp_f32 on [%g, %g): mean signed error %+.4f ulp, rms %.4f ulp\n", lo[r], hi[r], s / n, sqrt(sq / n));
    }
    hipFree(dx); hipFree(dob);
  }
  std::vector<float> x(n), o(5 * n);
  for (int i = 0; i < n; ++i) x[i] = 1.0f + (float)i / n;     // rcp on [1, 2); the others on the same values shifted to [-3, 3)
  float *dx, *dob; hipMalloc(&dx, n * 4); hipMalloc(&dob, 5 * n * 4);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) for (int i = 0; i < n; ++i) x[i] = 1.0f + 4.0f * (float)i / n;      // the saturated side: |result| in (0.73, 1)
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dob, n);
    hipMemcpy(o.data(), dob, 5 * n * 4, hipMemcpyDeviceToHost);
    const char *names[5] = {"v_rcp_f32", "fast_sigmoid", "sigmoidf_ (division)", "fast_tanh", "tanhf_ (division)"};
    for (int f = pass ? 1 : 0; f < (pass ? 5 : 1); ++f) {
      double s = 0, sq = 0;
      for (int i = 0; i < n; ++i) {
        const double v = x[i];
        const double ref = f == 0 ? 1.0 / v : f <= 2 ? 1.0 / (1.0 + exp(-v)) : tanh(v);
        int ex; frexp(ref, &ex);
        const double e = (o[(size_t)f * n + i] - ref) / ldexp(1.0, ex - 24);
        s += e; sq += e * e;
      }
      printf("%-22s mean signed error %+.4f ulp, rms %.4f ulp\n", names[f], s / n, sqrt(sq / n));
    }
  }
  return 0;
}
