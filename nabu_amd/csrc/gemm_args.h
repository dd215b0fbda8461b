// gemm_args.h — argument block shared by the fp32-MFMA and the bf16-split-MFMA GEMM kernels.
#pragma once
#include "common.h"

namespace nabu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;
constexpr int FBK = 32;   // k-tile of the fast kernels

struct GemmArgs {
  const float *A, *B, *bias;
  float *C;
  float *partial;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
  int kseg;
  long long a_seg, b_seg;
  int ksplit;  // k-range per z-slice (multiple of FBK)
  int nsplit;
  int vecA, vecB;  // 16-byte vector loads allowed
};

// skinny products (gemm_skinny.hip): M <= 64, no transposes
int gemm_skinny_chunk(int M, int N, int K);
int gemm_skinny_launch(const GemmArgs &a, hipStream_t stream);

// bf16-split kernels (gemm_bf16.hip); planes = 1 (bf16), 2 (bf16x3) or 3 (bf16x6)
int gemm_bf16_launch(const GemmArgs &a, bool transA, bool transB, int planes, dim3 grid, hipStream_t stream);

}  // namespace nabu
