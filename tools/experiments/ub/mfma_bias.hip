// Is the rounding of v_mfma_f32_16x16x32_f16 (and of the fp32 v_mfma_f32_4x4x1) biased?  One wave, many random tiles:
// D = A.B + C against a float64 reference; prints the mean SIGNED error in units of the result's last place, once as
// (D - ref) and once multiplied by sign(ref) (towards / away from zero), and the rms.  Round 5: the persistent backward
// recurrence's bias gradients (sums of 32000 dz) showed 4 x the step-wise kernels' error at T = 1000 — a systematic
// component no fp32 FMA chain has.
// build: hipcc --offload-arch=gfx950 -O2 tools/experiments/ub/mfma_bias.hip -o /tmp/mfma_bias && /tmp/mfma_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k16(const _Float16 *A, const _Float16 *B, const float *C, float *D, int tiles) {
  const int lane = threadIdx.x, n = lane & 15, q = lane >> 4;
  for (int t = 0; t < tiles; ++t) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = A[((size_t)t * 16 + n) * 32 + 8 * q + e];      // A[m = n][k]
      b[e] = B[((size_t)t * 16 + n) * 32 + 8 * q + e];      // B[n][k]
    }
    f4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[((size_t)t * 16 + 4 * q + i) * 16 + n];
    f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((size_t)t * 16 + 4 * q + i) * 16 + n] = d[i];
  }
}

int main() {
  const int tiles = 4096;
  std::vector<_Float16> A((size_t)tiles * 16 * 32), B(A.size());
  std::vector<float> C((size_t)tiles * 256), D(C.size());
  srand(7);
  auto rnd = []() { return (rand() / (double)RAND_MAX) * 2.0 - 1.0; };
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: operands ~ U(-1, 1) * 2^14 (the scaled planes), C = 0; mode 1: C of the magnitude of the sum;
    // mode 2: C large (a long accumulation: |C| ~ 30 x the 32-term sum)
    // (every row of a tile's A is the same vector, every row of its B too, C one constant: all 256 results of a tile are
    // the same dot product whatever the instruction's register layout is)
    for (int t = 0; t < tiles; ++t)
      for (int k = 0; k < 32; ++k) {
        const _Float16 av = (_Float16)(rnd() * 16384.0), bv = (_Float16)(rnd() * 16384.0);
        for (int m = 0; m < 16; ++m) { A[((size_t)t * 16 + m) * 32 + k] = av; B[((size_t)t * 16 + m) * 32 + k] = bv; }
      }
    const double cs = mode == 0 ? 0.0 : mode == 1 ? 16384.0 * 16384.0 * 3.0 : 16384.0 * 16384.0 * 100.0;
    for (int t = 0; t < tiles; ++t) { const float cv = (float)(rnd() * cs); for (int i = 0; i < 256; ++i) C[(size_t)t * 256 + i] = cv; }
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, tiles);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double s = 0, ss = 0, sq = 0; size_t cnt = 0;
    for (int t = 0; t < tiles; ++t)
      for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
          double ref = C[((size_t)t * 16 + m) * 16 + n], mag = fabs(ref);
          for (int k = 0; k < 32; ++k) {
            const double pr = (double)A[((size_t)t * 16 + m) * 32 + k] * (double)B[((size_t)t * 16 + n) * 32 + k];
            ref += pr; mag += fabs(pr);
          }
          const double got = D[((size_t)t * 16 + m) * 16 + n];
          int ex; frexp(mag, &ex);          // unit: the last place of the sum of MAGNITUDES (what an fp32 chain rounds at)
          const double ulp = ldexp(1.0, ex - 24);
          const double e = (got - ref) / ulp;
          s += e; ss += e * (ref > 0 ? 1 : -1); sq += e * e; ++cnt;
        }
    printf("16x16x32_f16 mode %d: mean signed error %+.4f ulp, toward(+)/away(-) from zero %+.4f ulp, rms %.4f ulp (RNE of the exact sum: 0, 0, 0.289)\n",
           mode, s / cnt, -ss / cnt, sqrt(sq / cnt));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
  }
  return 0;
}
