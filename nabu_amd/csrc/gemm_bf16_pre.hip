// gemm_bf16_pre.hip — bf16 matrix-pipe GEMM for operands that are ALREADY bf16 in memory, both with
// the reduction index contiguous:  C[M,N] (fp32) = alpha * sum_k A[m,k] * B[n,k] + beta*C + bias.
//
// Why a second bf16 kernel: gemm_bf16.hip reads fp32 operands and rounds them while it stages a tile —
// 4 bytes per element from L2 and a v_cvt_pk per pair on the critical path; at 128 x 128 tiles that is
// 64 flop per operand byte, L2-bandwidth bound at 250-380 TF/s.  The "bf16 input-to-hidden
// GEMMs" of BASELINE.json configs[4] use every operand several times (x: forward + weight gradient;
// dz: input gradient + weight gradient; the weights: every tile row), so the operands are converted ONCE
// (cvt kernels below: plain and transposed copies, 6 bytes of traffic per element) and the products run
// on 2-byte operands: 128 flop per byte, no conversion in the loop, staging is a 16-byte copy.
//
// Tile 128 x 128 x 64, 256 threads = 4 waves in 2 x 2, each wave 2 x 2 v_mfma_f32_32x32x16_bf16 tiles;
// LDS rows of 64 k + 8 pad bf16 (144 bytes: the 16 lanes of a ds_read_b128 phase hit 16 x 4 distinct
// banks), double-buffered (72 KB: two workgroups per CU), one barrier per k-tile, the registers of tile
// i+1 stored and the loads of tile i+2 issued between the MFMAs of tile i (the fp32 kernel's pipeline).
// Rounding is the same RNE conversion gemm_bf16.hip applies in its loop, accumulation is fp32 in the MFMA.
#include "gemm_args.h"

namespace nabu {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

constexpr int PKT = 64;                   // k-tile (elements)
constexpr int PROW = 2 * PKT + 16;        // bytes per LDS row
constexpr int PTILE = 128 * PROW;         // one operand tile

struct PreArgs {
  const unsigned short *A, *B;            // bf16 [M, lda], [N, ldb]; k contiguous
  int lda, ldb;
};

__global__ __launch_bounds__(256) void gemm_bf16_pre_kernel(GemmArgs a, PreArgs q) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: piece idx = tid + 256 j (j < 4): row idx >> 3, 16-byte piece idx & 7 of the row's 128 bytes
  pu32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const unsigned short *pa0, *pa1, *pa2, *pa3, *pb0, *pb1, *pb2, *pb3;
#define P_ADDR(j)                                                                                        \
  {                                                                                                      \
    const int idx = tid + 256 * j;                                                                       \
    pa##j = q.A + (size_t)min(m0 + (idx >> 3), a.M - 1) * q.lda + kbeg + 8 * (idx & 7);                  \
    pb##j = q.B + (size_t)min(n0 + (idx >> 3), a.N - 1) * q.ldb + kbeg + 8 * (idx & 7);                  \
  }
#define P_LOAD(j)                                                                                        \
  {                                                                                                      \
    ra##j = *reinterpret_cast<const pu32x4 *>(pa##j);                                                    \
    rb##j = *reinterpret_cast<const pu32x4 *>(pb##j);                                                    \
    pa##j += PKT;                                                                                        \
    pb##j += PKT;                                                                                        \
  }
#define P_STORE(j, buf)                                                                                  \
  {                                                                                                      \
    const int idx = tid + 256 * j;                                                                       \
    char *d_ = psm + (buf) * 2 * PTILE + (idx >> 3) * PROW + (idx & 7) * 16;                             \
    *reinterpret_cast<pu32x4 *>(d_) = ra##j;                                                             \
    *reinterpret_cast<pu32x4 *>(d_ + PTILE) = rb##j;                                                     \
  }
  P_ADDR(0) P_ADDR(1) P_ADDR(2) P_ADDR(3)
  P_LOAD(0) P_LOAD(1) P_LOAD(2) P_LOAD(3)
  P_STORE(0, 0) P_STORE(1, 0) P_STORE(2, 0) P_STORE(3, 0)
  if (kbeg + PKT < kend) { P_LOAD(0) P_LOAD(1) P_LOAD(2) P_LOAD(3) }
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += PKT) {
    const bool have_next = k0 + PKT < kend, load_next2 = k0 + 2 * PKT < kend;
    // operand fragment: row (lane & 31), 8 consecutive k at 16-byte chunk (lane >> 5) + 2 ks
    const char *ap = psm + cur * 2 * PTILE + (wm * 64 + (lane & 31)) * PROW + (lane >> 5) * 16;
    const char *bp = psm + cur * 2 * PTILE + PTILE + (wn * 64 + (lane & 31)) * PROW + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < PKT / 16; ++ks) {
      pbf16x8 fa[2], fb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t] = *reinterpret_cast<const pbf16x8 *>(ap + t * 32 * PROW + ks * 32);
        fb[t] = *reinterpret_cast<const pbf16x8 *>(bp + t * 32 * PROW + ks * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (ks == 0) {   // MFMAs are queued: stage the next tile behind them
        if (have_next) { P_STORE(0, cur ^ 1) P_STORE(1, cur ^ 1) P_STORE(2, cur ^ 1) P_STORE(3, cur ^ 1) }
        if (load_next2) { P_LOAD(0) P_LOAD(1) P_LOAD(2) P_LOAD(3) }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#undef P_ADDR
#undef P_LOAD
#undef P_STORE

  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + col;
      if (n >= a.N) continue;    // edge tiles: clamped (duplicate) loads, results dropped here
      const float bv = (a.nsplit == 1 && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + rbase;
        if (m >= a.M) continue;
        if (a.nsplit == 1) {
          float *c = a.C + (size_t)m * a.ldc + n;
          float v = a.alpha * acc[mi][ni][r] + bv;
          if (a.beta != 0.f) v += a.beta * *c;
          *c = v;
        } else {
          a.partial[((size_t)blockIdx.z * a.M + m) * a.N + n] = acc[mi][ni][r];
        }
      }
    }
}

// dst[r][c] = bf16(src[r][c]), 8 elements (16 bytes out) per thread; C % 8 == 0
__global__ __launch_bounds__(256) void cvt_bf16_kernel(size_t R, int C, const float *__restrict__ src, int lds_,
                                                      unsigned short *__restrict__ dst, int ldd) {
  const size_t n8 = R * (size_t)(C / 8);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const size_t r = i / (C / 8);
    const int c = (int)(i % (C / 8)) * 8;
    const float4 v0 = *reinterpret_cast<const float4 *>(src + r * lds_ + c);
    const float4 v1 = *reinterpret_cast<const float4 *>(src + r * lds_ + c + 4);
    pu32x4 o;
    o.x = __builtin_bit_cast(unsigned, __builtin_convertvector((pf32x2){v0.x, v0.y}, pbf16x2));
    o.y = __builtin_bit_cast(unsigned, __builtin_convertvector((pf32x2){v0.z, v0.w}, pbf16x2));
    o.z = __builtin_bit_cast(unsigned, __builtin_convertvector((pf32x2){v1.x, v1.y}, pbf16x2));
    o.w = __builtin_bit_cast(unsigned, __builtin_convertvector((pf32x2){v1.z, v1.w}, pbf16x2));
    *reinterpret_cast<pu32x4 *>(dst + r * ldd + c) = o;
  }
}

// dst[c][r] = bf16(src[r][c]): 64 x 64 tiles through LDS; grid (ceil(C/64), ceil(R/64)); R, C % 8 == 0 not
// required (edges guarded), ldd % 2 == 0
__global__ __launch_bounds__(256) void cvt_bf16_t_kernel(int R, int C, const float *__restrict__ src, int lds_,
                                                        unsigned short *__restrict__ dst, int ldd) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
  for (int i = tid; i < 64 * 16; i += 256) {          // 64 rows x 16 float4
    const int r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < R) {
      const float *p = src + (size_t)(r0 + r) * lds_ + c0 + c4;
      if (c0 + c4 + 3 < C) v = *reinterpret_cast<const float4 *>(p);
      else {
        if (c0 + c4 < C) v.x = p[0];
        if (c0 + c4 + 1 < C) v.y = p[1];
        if (c0 + c4 + 2 < C) v.z = p[2];
      }
    }
    tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
  }
  __syncthreads();
  for (int i = tid; i < 64 * 32; i += 256) {          // 64 output rows (c) x 32 pairs of r
    const int c = i >> 5, r2 = (i & 31) * 2;
    if (c0 + c < C && r0 + r2 < R) {
      const unsigned o = __builtin_bit_cast(unsigned, __builtin_convertvector((pf32x2){tile[r2][c], tile[r2 + 1][c]}, pbf16x2));
      unsigned short *d = dst + (size_t)(c0 + c) * ldd + r0 + r2;
      if (r0 + r2 + 1 < R) *reinterpret_cast<unsigned *>(d) = o;
      else *d = (unsigned short)(o & 0xffffu);
    }
  }
}

int cvt_bf16(size_t R, int C, const float *src, int ld, unsigned short *dst, int ldd, hipStream_t s) {
  if (R == 0 || C == 0) return 0;
  if (C % 8 || ld % 4 || ldd % 8) return fail(NABU_EUNSUP, "cvt_bf16: C %% 8, ld %% 4, ldd %% 8 required");
  size_t blocks = (R * (size_t)(C / 8) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cvt_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, R, C, src, ld, dst, ldd);
  NABU_LAUNCH_CHECK();
  return 0;
}
int cvt_bf16_t(int R, int C, const float *src, int ld, unsigned short *dst, int ldd, hipStream_t s) {
  if (R == 0 || C == 0) return 0;
  if (ld % 4 || ldd % 2) return fail(NABU_EUNSUP, "cvt_bf16_t: ld %% 4, ldd %% 2 required");
  hipLaunchKernelGGL(cvt_bf16_t_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, s, R, C, src, ld, dst, ldd);
  NABU_LAUNCH_CHECK();
  return 0;
}

bool gemm_bf16_pre_ok(int M, int N, int K, int lda, int ldb) {
  return M > 0 && N > 0 && K >= PKT && K % PKT == 0 && lda % 8 == 0 && ldb % 8 == 0;
}

size_t gemm_bf16_pre_ws_bytes(int M, int N, int K) {
  int ks;
  const int ns = gemm_split_for(M, N, K, PKT, 512, &ks);
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

// C = alpha * A·B^T + beta*C + bias with bf16 A [M,lda], B [N,ldb] (k contiguous), fp32 C
int gemm_bf16_pre(int M, int N, int K, float alpha, const unsigned short *A, int lda, const unsigned short *B, int ldb,
                  float beta, float *C, int ldc, const float *bias, void *ws, size_t ws_bytes, hipStream_t s) {
  if (!gemm_bf16_pre_ok(M, N, K, lda, ldb)) return fail(NABU_EUNSUP, "bf16 GEMM: unsupported shape M=%d N=%d K=%d", M, N, K);
  GemmArgs a;
  a.A = nullptr; a.B = nullptr; a.C = C; a.bias = bias; a.partial = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = alpha; a.beta = beta; a.kseg = 0; a.a_seg = a.b_seg = 0;
  a.vecA = a.vecB = 1; a.swz = 0; a.nbatch = 1; a.a_bs = a.b_bs = a.c_bs = 0;
  a.nsplit = gemm_split_for(M, N, K, PKT, 512, &a.ksplit);
  if (a.nsplit > 1) {
    const size_t need = (size_t)a.nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(NABU_EWS, "bf16 GEMM: workspace %zu < %zu", ws_bytes, need);
    a.partial = static_cast<float *>(ws);
  }
  PreArgs q = {A, B, lda, ldb};
  const size_t lds = 4 * (size_t)PTILE;
  NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16_pre_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, a.nsplit);
  hipLaunchKernelGGL(gemm_bf16_pre_kernel, grid, dim3(256), lds, s, a, q);
  NABU_LAUNCH_CHECK();
  if (a.nsplit > 1) return gemm_splitk_reduce(a, s);
  return 0;
}

}  // namespace nabu
