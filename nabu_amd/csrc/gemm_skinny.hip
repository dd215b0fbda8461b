// gemm_skinny.hip — C[M,N] = A[M,K] · B[K,N] for M <= 64 (the per-step products of the Speller
// decoder: 32..64 batch rows against a [K,N] weight matrix).  A 128x128 tile kernel wastes 3/4 of
// its MFMAs on such a product and occupies 16 of 256 CUs; it is bound by streaming B once.
//
// Grid = (N/32 column slices) x (K/KC k-chunks): every workgroup streams a [KC x 32] block of B
// exactly once with 128-byte coalesced rows, against A^T[KC x M] staged in LDS (k-major, so an
// MFMA operand is 32 consecutive floats).  The 4 waves split the k-chunk, their partial 32x32
// tiles meet in LDS; k-chunks are summed by the deterministic split-K reduce kernel of gemm.hip.
// Exact fp32 (v_mfma_f32_32x32x2_f32).
#include "gemm_args.h"

namespace nabu {

template <int MT>   // row tiles of 32
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int MR = 32 * MT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 32;
  const int KC = a.ksplit;
  const int k0 = blockIdx.y * KC;
  // A^T chunk -> LDS xT[k][m]
  for (int idx = tid; idx < (KC / 4) * MR; idx += 256) {
    const int m = idx % MR, kq = idx / MR;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < a.M) v = *reinterpret_cast<const float4 *>(a.A + (size_t)m * a.lda + k0 + 4 * kq);
    float *d = sm + (size_t)(4 * kq) * MR + m;
    d[0] = v.x; d[MR] = v.y; d[2 * MR] = v.z; d[3 * MR] = v.w;
  }
  __syncthreads();
  f32x16 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int kw = KC / 4;                              // k values of this wave
  const int li = lane & 31, lk = lane >> 5;
  const float *bp = a.B + (size_t)(k0 + w * kw + lk) * a.ldb + n0 + li;
  const float *xp = sm + (size_t)(w * kw + lk) * MR + li;
  const size_t bstep = 2 * (size_t)a.ldb;
  for (int kk = 0; kk < kw; kk += 16) {               // 8 MFMA k-steps per batch of loads
    float b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = bp[(size_t)j * bstep];
    bp += 8 * bstep;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[(size_t)(kk + 2 * j) * MR + 32 * t], b[j], acc[t], 0, 0, 0);
    }
  }
  __syncthreads();                                    // xT is dead: reuse LDS for the wave partials
  float *red = sm;                                    // [4 waves][MT][16][64]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((w * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < MT * 16 * 64; e += 256) {
    const float s = red[e] + red[MT * 1024 + e] + red[2 * MT * 1024 + e] + red[3 * MT * 1024 + e];
    const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
    const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = n0 + (l & 31);
    if (m >= a.M) continue;
    if (a.nsplit == 1) {
      float *c = a.C + (size_t)m * a.ldc + n;
      float v = a.alpha * s + (a.bias ? a.bias[n] : 0.f);
      if (a.beta != 0.f) v += a.beta * *c;
      *c = v;
    } else {
      a.partial[((size_t)blockIdx.y * a.M + m) * a.N + n] = s;
    }
  }
}

// k-chunk of the skinny kernel: the largest of 256/128/64 that divides K (0 = not eligible)
int gemm_skinny_chunk(int M, int N, int K) {
  if (M > 64 || N % 32 != 0 || K < 64) return 0;
  for (int kc = 256; kc >= 64; kc >>= 1)
    if (K % kc == 0) return kc;
  return 0;
}

int gemm_skinny_launch(const GemmArgs &a, hipStream_t s) {
  const int MT = a.M > 32 ? 2 : 1;
  const size_t xt = (size_t)a.ksplit * 32 * MT * sizeof(float), red = (size_t)4 * MT * 1024 * sizeof(float);
  const size_t lds = xt > red ? xt : red;
  dim3 grid(a.N / 32, a.nsplit);
  if (MT == 1) {
    hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), lds, s, a);
  } else {
    static bool configured = false;
    if (!configured) {
      NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
      configured = true;
    }
    hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), lds, s, a);
  }
  NABU_LAUNCH_CHECK();
  return 0;
}

}  // namespace nabu
