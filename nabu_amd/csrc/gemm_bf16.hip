// gemm_bf16.hip — GEMM on the gfx950 bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the fp32
// MFMA rate) for fp32 operands in HBM, fp32 accumulation, fp32 result.
//
//   planes = 1  "bf16"   : operands rounded to bf16 (RNE) — the bf16 input-to-hidden GEMMs of
//                          BASELINE.json configs[4] (cfg5);
//   planes = 2  "bf16x3" : x = hi + mid (two bf16), products hi·hi + hi·mid + mid·hi,
//                          relative error of a product ~2^-16;
//   planes = 3  "bf16x6" : x = hi + mid + lo — the three bf16 pieces hold all 24 significand
//                          bits of an fp32 value — products hi·hi, hi·mid, mid·hi, hi·lo, mid·mid,
//                          lo·hi (the dropped terms are <= 2^-24 of the product): an
//                          fp32-accurate GEMM at 6/16 of the fp32-MFMA instruction time.
// Every bf16 x bf16 product is exact in fp32; sums are accumulated in fp32 by the MFMA.
//
// The split happens while a tile is staged: global fp32 -> registers (prefetched under the
// previous tile's MFMAs) -> v_cvt_pk_bf16_f32 / subtract -> bf16 planes in LDS, row-major with
// k contiguous ([row][32 k + 8 pad] bf16 = 80-byte rows) so that an MFMA operand (8 consecutive
// k of one row) is ONE conflict-free ds_read_b128.  Operands that are contiguous along the row
// index in memory (A^T, B) are transposed 4x4 in registers on the way.
// Tiling, split-K, segmented K, edge handling and the epilogue are those of gemm.hip.
#include "gemm_args.h"

namespace nabu {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int ROWB = 80;                 // bytes per LDS row: 32 bf16 + 8 pad
constexpr int PLANE = 128 * ROWB;        // one plane of one operand tile

// two fp32 -> NP packed bf16 pairs (low half = first element)
template <int NP>
__device__ __forceinline__ void split2(float x0, float x1, unsigned (&out)[NP]) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const f32x2 v = {x0, x1};
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    out[p] = h;
    if (p + 1 < NP) {   // exact remainders
      x0 -= __builtin_bit_cast(float, h << 16);
      x1 -= __builtin_bit_cast(float, h & 0xffff0000u);
    }
  }
}

template <bool TA, bool TB, int NP>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *As = smem, *Bs = smem + NP * PLANE;
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

  // staging identities
  //   k-contiguous operand (A with !TA, B with TB): thread = (row tid>>1, 16 k at (tid&1)*16)
  //   row-contiguous operand (A with TA, B with !TB): thread = (k-quad tid&7, row-quad tid>>3)
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // named: arrays ended up in scratch (gemm.hip)
#define NABU_BLOAD1(j, k0_)                                                                        \
  {                                                                                                \
    if (TA) {                                                                                      \
      const int k = (k0_) + 4 * (tid & 7) + j;                                                     \
      const size_t off = a.kseg > 0 ? (size_t)(k / a.kseg) * (size_t)a.a_seg +                     \
                                          (size_t)(k % a.kseg) * a.lda                             \
                                    : (size_t)k * a.lda;                                           \
      ra##j = *reinterpret_cast<const float4 *>(a.A + off + min(m0 + 4 * (tid >> 3), a.M - 4));    \
    } else {                                                                                       \
      ra##j = *reinterpret_cast<const float4 *>(a.A + (size_t)min(m0 + (tid >> 1), a.M - 1) * a.lda + \
                                                (k0_) + 16 * (tid & 1) + 4 * j);                   \
    }                                                                                              \
    if (TB) {                                                                                      \
      rb##j = *reinterpret_cast<const float4 *>(a.B + (size_t)min(n0 + (tid >> 1), a.N - 1) * a.ldb + \
                                                (k0_) + 16 * (tid & 1) + 4 * j);                   \
    } else {                                                                                       \
      const int k = (k0_) + 4 * (tid & 7) + j;                                                     \
      const size_t off = a.kseg > 0 ? (size_t)(k / a.kseg) * (size_t)a.b_seg +                     \
                                          (size_t)(k % a.kseg) * a.ldb                             \
                                    : (size_t)k * a.ldb;                                           \
      rb##j = *reinterpret_cast<const float4 *>(a.B + off + min(n0 + 4 * (tid >> 3), a.N - 4));    \
    }                                                                                              \
  }
#define NABU_BLOAD(k0_) { NABU_BLOAD1(0, k0_) NABU_BLOAD1(1, k0_) NABU_BLOAD1(2, k0_) NABU_BLOAD1(3, k0_) }

  // registers -> bf16 planes in LDS
  auto store_kc = [&](char *S, const float4 &v0, const float4 &v1, const float4 &v2, const float4 &v3) {
    // 16 consecutive k of one row: 2 x 16 bytes per plane
    unsigned q[8][NP];
    split2<NP>(v0.x, v0.y, q[0]); split2<NP>(v0.z, v0.w, q[1]);
    split2<NP>(v1.x, v1.y, q[2]); split2<NP>(v1.z, v1.w, q[3]);
    split2<NP>(v2.x, v2.y, q[4]); split2<NP>(v2.z, v2.w, q[5]);
    split2<NP>(v3.x, v3.y, q[6]); split2<NP>(v3.z, v3.w, q[7]);
    char *dst = S + (tid >> 1) * ROWB + (tid & 1) * 32;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      *reinterpret_cast<u32x4 *>(dst + p * PLANE) = (u32x4){q[0][p], q[1][p], q[2][p], q[3][p]};
      *reinterpret_cast<u32x4 *>(dst + p * PLANE + 16) = (u32x4){q[4][p], q[5][p], q[6][p], q[7][p]};
    }
  };
  auto store_rc = [&](char *S, const float4 &v0, const float4 &v1, const float4 &v2, const float4 &v3) {
    // v_j = 4 consecutive rows at k-quad element j: row i gets (v0[i], v1[i], v2[i], v3[i]) = 4 k
    const float c0[4] = {v0.x, v0.y, v0.z, v0.w}, c1[4] = {v1.x, v1.y, v1.z, v1.w};
    const float c2[4] = {v2.x, v2.y, v2.z, v2.w}, c3[4] = {v3.x, v3.y, v3.z, v3.w};
    char *dst = S + 4 * (tid >> 3) * ROWB + (tid & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned lo[NP], hi[NP];
      split2<NP>(c0[i], c1[i], lo);
      split2<NP>(c2[i], c3[i], hi);
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x2 *>(dst + p * PLANE + i * ROWB) = (u32x2){lo[p], hi[p]};
    }
  };
#define NABU_BSTORE()                                                  \
  {                                                                    \
    if (TA) store_rc(As, ra0, ra1, ra2, ra3); else store_kc(As, ra0, ra1, ra2, ra3); \
    if (TB) store_kc(Bs, rb0, rb1, rb2, rb3); else store_rc(Bs, rb0, rb1, rb2, rb3); \
  }

  NABU_BLOAD(kbeg);
  NABU_BSTORE();
  __syncthreads();
  // operand fragment: row (lane & 31), 8 consecutive k at chunk (lane >> 5) + 2*ks
  const char *ap = As + (wm * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const char *bp = Bs + (wn * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
  for (int k0 = kbeg; k0 < kend; k0 += FBK) {
    const bool more = k0 + FBK < kend;
    if (more) NABU_BLOAD(k0 + FBK);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          fa[t][p] = *reinterpret_cast<const bf16x8 *>(ap + p * PLANE + t * 32 * ROWB + ks * 32);
          fb[t][p] = *reinterpret_cast<const bf16x8 *>(bp + p * PLANE + t * 32 * ROWB + ks * 32);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest terms first
          if (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][2], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][1], acc[i][j], 0, 0, 0);
          }
          if (NP >= 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][1], fb[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][0], fb[j][0], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      NABU_BSTORE();
      __syncthreads();
    }
  }
#undef NABU_BLOAD
#undef NABU_BLOAD1
#undef NABU_BSTORE

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

template <int NP>
static int launch_np(const GemmArgs &a, bool ta, bool tb, dim3 grid, hipStream_t s) {
  const size_t lds = 2 * (size_t)NP * PLANE;
  auto go = [&](auto kernel) -> int {
    static thread_local bool configured = false;
    if (!configured && lds > 48 * 1024) {
      NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
    NABU_LAUNCH_CHECK();
    return 0;
  };
  if (ta && tb) return go(gemm_bf16_kernel<true, true, NP>);
  if (ta) return go(gemm_bf16_kernel<true, false, NP>);
  if (tb) return go(gemm_bf16_kernel<false, true, NP>);
  return go(gemm_bf16_kernel<false, false, NP>);
}

int gemm_bf16_launch(const GemmArgs &a, bool transA, bool transB, int planes, dim3 grid, hipStream_t stream) {
  switch (planes) {
    case 1: return launch_np<1>(a, transA, transB, grid, stream);
    case 2: return launch_np<2>(a, transA, transB, grid, stream);
    case 3: return launch_np<3>(a, transA, transB, grid, stream);
  }
  return fail(NABU_EINVAL, "gemm: bad number of bf16 planes %d", planes);
}

}  // namespace nabu
