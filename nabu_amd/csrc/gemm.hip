// gemm.hip — fp32 GEMM on the gfx950 f32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// C[M,N] = alpha * op(A) * op(B) + beta * C + bias, row-major, exact f32
// (the f32-input MFMA is bitwise a k-ordered fmaf chain).  Used for the
// time-batched input projections X·Wx of every (B)LSTM layer, their gradients
// dX = dZ·Wx^T, dWx = X^T·dZ, dWh = H_{t-1}^T·dZ (segmented K), the DNNDecoder
// output layer and the attention Dense layers.
//
// Tiling: 128x128x16 block tile, 256 threads = 4 wave64 in a 2x2 grid, each
// wave owns 64x64 = 2x2 MFMA tiles (64 accumulator VGPRs).  Operands are staged
// k-major in LDS (As[k][m], Bs[k][n]) so that the 32 lanes of an MFMA operand
// read 32 consecutive floats (conflict-free ds_read_b32); the next k-tile is
// prefetched into registers while the current one is multiplied.
// Small-MN / long-K products (weight gradients) use deterministic split-K:
// partial tiles go to a workspace and are summed in fixed order by a second
// kernel (no float atomics).
#include "gemm_args.h"

#include <stdlib.h>
#include <string>

namespace nabu {

// k-tile of gemm_f32_fast_kernel: 16 (2 x 2 x 8.25 KiB of LDS) lets three workgroups share a CU;
// measured 2-3 % faster than 32 on the NT / TN products
constexpr int FKT = 16;

// ---- "k-contiguous" operand: element (r,k) at base[r*ld + k] ----------------
__device__ __forceinline__ void load_kc(float4 (&v)[2], const float *base, int ld, int r0,
                                        int rmax, int k0, int kend, int vec, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int idx = tid + 256 * j;
    int r = r0 + (idx >> 2), k = k0 + 4 * (idx & 3);
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rmax) {
      const float *p = base + (size_t)r * ld + k;
      if (vec && k + 3 < kend) {
        x = *reinterpret_cast<const float4 *>(p);
      } else {
        if (k < kend) x.x = p[0];
        if (k + 1 < kend) x.y = p[1];
        if (k + 2 < kend) x.z = p[2];
        if (k + 3 < kend) x.w = p[3];
      }
    }
    v[j] = x;
  }
}
__device__ __forceinline__ void store_kc(float (*S)[LDT], const float4 (&v)[2], int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int idx = tid + 256 * j;
    int r = idx >> 2, k = 4 * (idx & 3);
    S[k + 0][r] = v[j].x;
    S[k + 1][r] = v[j].y;
    S[k + 2][r] = v[j].z;
    S[k + 3][r] = v[j].w;
  }
}
// ---- "r-contiguous" operand: element (r,k) at base[row(k) + r] ---------------
__device__ __forceinline__ void load_rc(float4 (&v)[2], const float *base, int ld, int kseg,
                                        long long seg, int r0, int rmax, int k0, int kend,
                                        int vec, int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int idx = tid + 256 * j;
    int k = k0 + (idx >> 5), r = r0 + 4 * (idx & 31);
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < kend) {
      size_t off = kseg > 0 ? (size_t)(k / kseg) * (size_t)seg + (size_t)(k % kseg) * ld
                            : (size_t)k * ld;
      const float *p = base + off + r;
      if (vec && r + 3 < rmax) {
        x = *reinterpret_cast<const float4 *>(p);
      } else {
        if (r < rmax) x.x = p[0];
        if (r + 1 < rmax) x.y = p[1];
        if (r + 2 < rmax) x.z = p[2];
        if (r + 3 < rmax) x.w = p[3];
      }
    }
    v[j] = x;
  }
}
__device__ __forceinline__ void store_rc(float (*S)[LDT], const float4 (&v)[2], int tid) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int idx = tid + 256 * j;
    *reinterpret_cast<float4 *>(&S[idx >> 5][4 * (idx & 31)]) = v[j];
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BK][LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BK][LDT];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (a.nbatch > 1) {   // batched: blockIdx.z selects the product, the whole K is one slice
    a.A += (size_t)blockIdx.z * a.a_bs;
    a.B += (size_t)blockIdx.z * a.b_bs;
    a.C += (size_t)blockIdx.z * a.c_bs;
  }
  const int kbeg = a.nbatch > 1 ? 0 : blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2], rb[2];
  auto gload = [&](int k0) {
    if (TA) load_rc(ra, a.A, a.lda, a.kseg, a.a_seg, m0, a.M, k0, kend, a.vecA, tid);
    else    load_kc(ra, a.A, a.lda, m0, a.M, k0, kend, a.vecA, tid);
    if (TB) load_kc(rb, a.B, a.ldb, n0, a.N, k0, kend, a.vecB, tid);
    else    load_rc(rb, a.B, a.ldb, a.kseg, a.b_seg, n0, a.N, k0, kend, a.vecB, tid);
  };
  auto sstore = [&]() {
    if (TA) store_rc(As, ra, tid); else store_kc(As, ra, tid);
    if (TB) store_kc(Bs, rb, tid); else store_rc(Bs, rb, tid);
  };

  if (kbeg < kend) {
    gload(kbeg);
    sstore();
    __syncthreads();
    const int li = lane & 31, lk = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = k0 + BK < kend;
      if (more) gload(k0 + BK);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a0 = As[kk + lk][wm * 64 + li];
        float a1 = As[kk + lk][wm * 64 + 32 + li];
        float b0 = Bs[kk + lk][wn * 64 + li];
        float b1 = Bs[kk + lk][wn * 64 + 32 + li];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        sstore();
        __syncthreads();
      }
    }
  }

  // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + col;
      if (n >= a.N) continue;
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


// ---------------------------------------------------------------------------
// Fast path: K-range % 32 == 0, M % 4 == N % 4 == 0 and 16-byte aligned operands ->
// no guards and no divergence in the loads (edge tiles clamp their addresses and drop
// the surplus rows/columns at the store), so the next k-tile's global loads really
// stay in flight under the current tile's 64 MFMAs per wave.

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_fast_kernel(GemmArgs a) {
  constexpr int KT = FKT, KQ = KT / 4;   // k-tile, 16-byte pieces per row
  // double-buffered operand tiles.  An operand whose contiguous dimension is the row/column index
  // (A with TA, B with !TB) is staged k-major [KT][LDT] with 16-byte stores; an operand whose contiguous
  // dimension is k (A with !TA, B with TB) is staged row-major [128][LDK = KT + 2] with two 8-byte
  // stores per 16-byte piece (the k-major layout needs four scalar stores with bank conflicts for it:
  // 8-16 ds_write_b32 per thread and k-tile made the NN / NT variants LDS-store bound).  Both layouts
  // give conflict-free ds_read_b32 MFMA operands (32 consecutive floats, resp. stride 18 = 16 even banks
  // for the 32 rows of one k, the odd banks for the other k of the pair).
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  constexpr int LDK = KT + 2;
  constexpr int TILE_F = KT * LDT > 128 * LDK ? KT * LDT : 128 * LDK;     // floats per operand tile
#define NABU_ATILE(buf) (fsm + (buf) * (2 * TILE_F))
#define NABU_BTILE(buf) (fsm + (buf) * (2 * TILE_F) + TILE_F)
#define NABU_AEL(t, r, kx) (TA ? (t)[(kx) * LDT + (r)] : (t)[(r) * LDK + (kx)])
#define NABU_BEL(t, c, kx) (TB ? (t)[(c) * LDK + (kx)] : (t)[(kx) * LDT + (c)])
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  int tile_m, tile_n;
  tile_of_block(a.swz, &tile_m, &tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Per-thread pieces of the two operand tiles (4 x 16 bytes each), addressed with
  // compile-time indices only (anything fancier made hipcc put them in scratch).
  //   k-contiguous operand (A with !TA, B with TB): element (r,k) at base[r*ld + k]
  //   r-contiguous operand (A with TA, B with !TB): element (r,k) at base[row(k) + r]
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // named scalars: arrays ended up in scratch
  // Load addresses are kept as running pointers (one per 16-byte piece) that advance by a constant
  // per k-tile; recomputing them cost ~150 integer instructions per tile.  Segmented K (kseg > 0)
  // has irregular row offsets and recomputes.
  const float *pa0, *pa1, *pa2, *pa3, *pb0, *pb1, *pb2, *pb3;
#define NABU_GADDR1(j, k0_)                                                                      \
  {                                                                                              \
    const int idx = tid + 256 * j;                                                               \
    if (TA) {                                                                                    \
      const int k = (k0_) + (idx >> 5);                                                          \
      const size_t off = a.kseg > 0 ? (size_t)(k / a.kseg) * (size_t)a.a_seg +                   \
                                          (size_t)(k % a.kseg) * a.lda                           \
                                    : (size_t)k * a.lda;                                         \
      pa##j = a.A + off + min(m0 + 4 * (idx & 31), a.M - 4);                                     \
    } else {                                                                                     \
      pa##j = a.A + (size_t)min(m0 + (idx / KQ), a.M - 1) * a.lda + (k0_) + 4 * (idx % KQ);       \
    }                                                                                            \
    if (TB) {                                                                                    \
      pb##j = a.B + (size_t)min(n0 + (idx / KQ), a.N - 1) * a.ldb + (k0_) + 4 * (idx % KQ);       \
    } else {                                                                                     \
      const int k = (k0_) + (idx >> 5);                                                          \
      const size_t off = a.kseg > 0 ? (size_t)(k / a.kseg) * (size_t)a.b_seg +                   \
                                          (size_t)(k % a.kseg) * a.ldb                           \
                                    : (size_t)k * a.ldb;                                         \
      pb##j = a.B + off + min(n0 + 4 * (idx & 31), a.N - 4);                                     \
    }                                                                                            \
  }
#define NABU_GADDR(k0_) { NABU_GADDR1(0, k0_) NABU_GADDR1(1, k0_) if (KT > 16) { NABU_GADDR1(2, k0_) NABU_GADDR1(3, k0_) } }
  const size_t stepA = TA ? (size_t)KT * a.lda : (size_t)KT;
  const size_t stepB = TB ? (size_t)KT : (size_t)KT * a.ldb;
#define NABU_GLOAD1(j, k0_)                                                                      \
  {                                                                                              \
    ra##j = *reinterpret_cast<const float4 *>(pa##j);                                            \
    rb##j = *reinterpret_cast<const float4 *>(pb##j);                                            \
    pa##j += stepA;                                                                              \
    pb##j += stepB;                                                                              \
  }
#define NABU_GLOAD(k0_)                                                                          \
  {                                                                                              \
    if (a.kseg > 0) NABU_GADDR(k0_)                                                              \
    NABU_GLOAD1(0, k0_) NABU_GLOAD1(1, k0_) if (KT > 16) { NABU_GLOAD1(2, k0_) NABU_GLOAD1(3, k0_) }  \
  }
#define NABU_SSTORE1(j)                                                                          \
  {                                                                                              \
    float *As = NABU_ATILE(sbuf), *Bs = NABU_BTILE(sbuf);                                        \
    const int idx = tid + 256 * j;                                                               \
    if (TA) {                                                                                    \
      *reinterpret_cast<float4 *>(&As[(idx >> 5) * LDT + 4 * (idx & 31)]) = ra##j;               \
    } else {                                                                                     \
      float *d_ = &As[(idx / KQ) * LDK + 4 * (idx % KQ)];                                         \
      *reinterpret_cast<float2 *>(d_) = make_float2(ra##j.x, ra##j.y);                           \
      *reinterpret_cast<float2 *>(d_ + 2) = make_float2(ra##j.z, ra##j.w);                       \
    }                                                                                            \
    if (TB) {                                                                                    \
      float *d_ = &Bs[(idx / KQ) * LDK + 4 * (idx % KQ)];                                         \
      *reinterpret_cast<float2 *>(d_) = make_float2(rb##j.x, rb##j.y);                           \
      *reinterpret_cast<float2 *>(d_ + 2) = make_float2(rb##j.z, rb##j.w);                       \
    } else {                                                                                     \
      *reinterpret_cast<float4 *>(&Bs[(idx >> 5) * LDT + 4 * (idx & 31)]) = rb##j;               \
    }                                                                                            \
  }
#define NABU_SSTORE() { NABU_SSTORE1(0) NABU_SSTORE1(1) if (KT > 16) { NABU_SSTORE1(2) NABU_SSTORE1(3) } }

  // Pipeline, ONE barrier per k-tile: while tile i is multiplied out of LDS buffer i%2, the
  // registers holding tile i+1 (loaded during the previous iteration) are written to the other
  // buffer and the global loads of tile i+2 are issued — LDS stores and global loads sit between
  // the MFMAs of the same wave (the matrix pipe runs them asynchronously).
  int sbuf = 0;
  NABU_GADDR(kbeg);
  NABU_GLOAD(kbeg);
  NABU_SSTORE();
  const bool two = kbeg + KT < kend && !(a.vecA & 2);   // vecA bit 1: timing experiment (no staging)
  if (two) NABU_GLOAD(kbeg + KT);
  __syncthreads();
  const int li = lane & 31, lk = lane >> 5;
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += KT) {
    const bool have_next = k0 + KT < kend && !(a.vecA & 2);       // registers hold tile k0+KT
    const bool load_next2 = k0 + 2 * KT < kend && !(a.vecA & 2);
    const float *As = NABU_ATILE(cur), *Bs = NABU_BTILE(cur);
    // Operand reads as explicit ds_read_b32 with IMMEDIATE offsets off one base register per operand and
    // tile: hipcc pairs the reads into ds_read2(st64)_b32, whose 8-bit offsets cannot hold the k-step
    // offset, and re-computes a base with a VALU add for every pair — ~1 VALU instruction per MFMA, paid
    // out of the matrix pipe's own cycles (fp32 MFMA and VALU share it on gfx950).  The loads are
    // invisible to the compiler's wait-count bookkeeping, so every k-step claims the previous step's four
    // values with a counted wait (LDS operations complete in order: "at most the four newest outstanding"
    // means everything older, compiler-issued stores included, is done) whose in/out operands tie the
    // MFMAs below to it.  Operands of k-step kk+2 are read before the MFMAs of k-step kk are issued.
    const unsigned a_base = (unsigned)(reinterpret_cast<uintptr_t>(TA ? As + lk * LDT + wm * 64 + li : As + (wm * 64 + li) * LDK + lk));
    const unsigned b_base = (unsigned)(reinterpret_cast<uintptr_t>(TB ? Bs + (wn * 64 + li) * LDK + lk : Bs + lk * LDT + wn * 64 + li));
    constexpr int A_KSTEP = TA ? 2 * LDT * 4 : 2 * 4, A_ROW32 = TA ? 32 * 4 : 32 * LDK * 4;   // bytes per k-step / per 32 rows
    constexpr int B_KSTEP = TB ? 2 * 4 : 2 * LDT * 4, B_ROW32 = TB ? 32 * LDK * 4 : 32 * 4;
#define NABU_LDSR(dst, base, off) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(base), "n"(off))
    float a0n, a1n, b0n, b1n;
    NABU_LDSR(a0n, a_base, 0); NABU_LDSR(a1n, a_base, A_ROW32);
    NABU_LDSR(b0n, b_base, 0); NABU_LDSR(b1n, b_base, B_ROW32);
#pragma unroll
    for (int kk = 0; kk < KT; kk += 2) {
      float a0 = a0n, a1 = a1n, b0 = b0n, b1 = b1n;
      if (kk + 2 < KT) {
        NABU_LDSR(a0n, a_base, (kk / 2 + 1) * A_KSTEP); NABU_LDSR(a1n, a_base, (kk / 2 + 1) * A_KSTEP + A_ROW32);
        NABU_LDSR(b0n, b_base, (kk / 2 + 1) * B_KSTEP); NABU_LDSR(b1n, b_base, (kk / 2 + 1) * B_KSTEP + B_ROW32);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      if (kk == 2) {   // a few MFMAs are queued: stage the next tile behind them
        if (have_next && !(a.vecA & 8)) {
          sbuf = cur ^ 1;
          NABU_SSTORE();
        }
        if (load_next2 && !(a.vecA & 4)) NABU_GLOAD(k0 + 2 * KT);
      }
    }
    __syncthreads();
    cur ^= 1;
  }
#undef NABU_LDSR
#undef NABU_ATILE
#undef NABU_BTILE
#undef NABU_AEL
#undef NABU_BEL
#undef NABU_GADDR
#undef NABU_GADDR1
#undef NABU_GLOAD
#undef NABU_SSTORE
#undef NABU_GLOAD1
#undef NABU_SSTORE1

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


// ---------------------------------------------------------------------------
// Short reduction, tall output (the first layer's input projection: [B*T, D] · [D, 4H], D = 40 or 80): the
// product is bound by WRITING C (262 MB at cfg2), not by the matrix pipe.  One load phase for the whole
// K (A tile [128 x K] row-major with an odd row stride, B tile [K x 128] k-major), one barrier, K/2 MFMA
// steps, stores.  No k-loop pipeline to fill and drain: the generic kernel spent two barriers and a
// load round trip per 16 reduction indices on it (150 us against ~60 us for the write alone).
constexpr int SMALLK_MAX = 96;
__global__ __launch_bounds__(256) void gemm_smallk_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float ssm[];
  // K > 48 is taken in chunks of <= 48 (K = 80: 40 + 40) so that the operand tiles never need more LDS than
  // the 128 x 128 output tile of the epilogue does: two workgroups per CU at every K
  const int nchunk = (a.K + 47) / 48, K = ((a.K + nchunk - 1) / nchunk + 3) & ~3, LDA_S = K + 1;   // odd stride: 32 rows hit 32 banks
  float *As = ssm, *Bs = ssm + 128 * LDA_S + 3;     // Bs 16-byte aligned below
  Bs = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(Bs) + 15) & ~(uintptr_t)15);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, lk = lane >> 5;
  const float *ap0 = As + (wm * 64 + li) * LDA_S + lk, *ap1 = ap0 + 32 * LDA_S;
  const float *bp0 = Bs + lk * LDT + wn * 64 + li;
  for (int kc = 0; kc < a.K; kc += K) {
    const int kn = min(K, a.K - kc);                  // a multiple of 4 (K % 4 == 0); odd halves are zero-filled below
    const int KQ = kn / 4;
    if (kc) __syncthreads();                          // the previous chunk's operands have been read
    for (int idx = tid; idx < 128 * KQ; idx += 256) {          // A: rows clamped, dropped at the store
      const int r = idx / KQ, kq = idx % KQ;
      const float4 v = *reinterpret_cast<const float4 *>(a.A + (size_t)min(m0 + r, a.M - 1) * a.lda + kc + 4 * kq);
      float *d = As + r * LDA_S + 4 * kq;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int idx = tid; idx < kn * 32; idx += 256) {           // B: 16-byte pieces along n
      const int k = idx >> 5, c = 4 * (idx & 31);
      const float4 v = *reinterpret_cast<const float4 *>(a.B + (size_t)(kc + k) * a.ldb + min(n0 + c, a.N - 4));
      *reinterpret_cast<float4 *>(Bs + k * LDT + c) = v;
    }
    __syncthreads();
    for (int kk = 0; kk < kn; kk += 2) {
      const float a0 = ap0[kk], a1 = ap1[kk], b0 = bp0[kk * LDT], b1 = bp0[kk * LDT + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // epilogue through LDS: the accumulators hold one column per lane; written back as they are that is a
  // 4-byte store per lane and element (2.2 TB/s on this write-bound product).  Transposed through LDS a
  // lane stores 16 bytes of a row and a wave a full 512-byte row segment at a time.
  __syncthreads();                                   // operand tiles are dead
  float *Cs = ssm;                                   // [128][LDT]
  const int col = lane & 31, rbase = 4 * (lane >> 5);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + rbase) * LDT + wn * 64 + ni * 32 + col] = acc[mi][ni][r];
  __syncthreads();
  const int c4 = 4 * (tid & 31);
  const int n = n0 + c4;
  if (n < a.N) {                                     // N % 4 == 0: a 16-byte piece is inside or outside
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bv = *reinterpret_cast<const float4 *>(a.bias + n);
    for (int r = tid >> 5; r < 128; r += 8) {
      const int m = m0 + r;
      if (m >= a.M) break;
      const float4 v = *reinterpret_cast<const float4 *>(Cs + r * LDT + c4);
      float4 o = make_float4(a.alpha * v.x + bv.x, a.alpha * v.y + bv.y, a.alpha * v.z + bv.z, a.alpha * v.w + bv.w);
      float4 *c = reinterpret_cast<float4 *>(a.C + (size_t)m * a.ldc + n);
      if (a.beta != 0.f) {
        const float4 p = *c;
        o.x += a.beta * p.x; o.y += a.beta * p.y; o.z += a.beta * p.z; o.w += a.beta * p.w;
      }
      *c = o;
    }
  }
}
static bool smallk_ok(bool ta, bool tb, int M, int N, int K, const void *A, int lda, const void *B, int ldb, int kseg,
                      const void *C, int ldc, const void *bias) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return !ta && !tb && kseg == 0 && K >= 8 && K <= SMALLK_MAX && K % 4 == 0 && M >= 2048 && N >= 4 && N % 4 == 0 &&
         al16(A) && lda % 4 == 0 && al16(B) && ldb % 4 == 0 && al16(C) && ldc % 4 == 0 && al16(bias);
}
static size_t smallk_lds(int Kfull) {
  const int nchunk = (Kfull + 47) / 48, K = ((Kfull + nchunk - 1) / nchunk + 3) & ~3;
  const size_t ops = ((size_t)128 * (K + 1) + 8 + (size_t)K * LDT) * sizeof(float), out = (size_t)128 * LDT * sizeof(float);
  return ops > out ? ops : out;
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs a) {
  const size_t total = (size_t)a.M * a.N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int m = (int)(i / a.N), n = (int)(i % a.N);
    float s = 0.f;
    for (int z = 0; z < a.nsplit; ++z) s += a.partial[(size_t)z * total + i];
    float v = a.alpha * s + (a.bias ? a.bias[n] : 0.f);
    float *c = a.C + (size_t)m * a.ldc + n;
    if (a.beta != 0.f) v += a.beta * *c;
    *c = v;
  }
}

// split-K policy: only for products with few output tiles and a long reduction.
static int choose_split(int M, int N, int K, int *ksplit) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int ns = 1;
  if (tiles < 768 && K >= 1024) {   // aim at 3 workgroups per CU (256 CUs; the 16-wide k-tile fits three)
    ns = (768 + tiles - 1) / tiles;
    const int maxs = K / (K >= 4096 ? 512 : 256);
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
  }
  int ks = (K + ns - 1) / ns;
  ks = (ks + FBK - 1) / FBK * FBK;
  ns = (K + ks - 1) / ks;
  *ksplit = ks;
  return ns;
}

int gemm_split_for(int M, int N, int K, int kmult, int target, int *ksplit) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int ns = 1;
  if (tiles < target && K >= 1024) {
    ns = (target + tiles - 1) / tiles;
    const int maxs = K / (K >= 4096 ? 512 : 256);
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
  }
  int ks = (K + ns - 1) / ns;
  ks = (ks + kmult - 1) / kmult * kmult;
  ns = (K + ks - 1) / ks;
  *ksplit = ks;
  return ns;
}

int gemm_splitk_reduce(const GemmArgs &a, hipStream_t s) {
  const size_t total = (size_t)a.M * a.N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

int gemm_batched_f32(bool transA, bool transB, int M, int N, int K, const float *A, int lda, long long a_bs,
                     const float *B, int ldb, long long b_bs, float beta, float *C, int ldc, long long c_bs, int nbatch,
                     hipStream_t s) {
  if (M <= 0 || N <= 0 || nbatch <= 0) return 0;
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = nullptr; a.partial = nullptr;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = 1.f; a.beta = beta; a.kseg = 0; a.a_seg = a.b_seg = 0;
  a.ksplit = (K + BK - 1) / BK * BK; if (a.ksplit < BK) a.ksplit = BK;
  a.nsplit = 1; a.swz = 0;
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vecA = al16(A) && lda % 4 == 0 && a_bs % 4 == 0;
  a.vecB = al16(B) && ldb % 4 == 0 && b_bs % 4 == 0;
  a.nbatch = nbatch > 1 ? nbatch : 2;   // > 1 selects the batched addressing (a single product: z = 0 only)
  a.a_bs = a_bs; a.b_bs = b_bs; a.c_bs = c_bs;
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, nbatch), block(256);
  if (transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, a);
  else if (transA) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, a);
  else if (transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

}  // namespace nabu

using namespace nabu;

extern "C" size_t nabu_gemm_ws_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int ks;
  int ns = choose_split(M, N, K, &ks);
  if (K % FKT && K > FKT) {             // the K-tail split runs the fast kernel on K - K % 16
    const int nm = choose_split(M, N, K - K % FKT, &ks);
    if (nm > ns) ns = nm;
  }
  const int kc = gemm_skinny_chunk(M, N, K);
  if (kc && K / kc > ns) ns = K / kc;
  return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 0;
}

// [A | A2]·[B ; B2] with the split-K reduction inside the launch (gemm_skinny.hip)
extern "C" size_t nabu_gemm2_ws_bytes(int M, int N, int K1, int K2) {
  if (M <= 0 || N <= 0 || K1 <= 0 || K2 < 0) return 0;
  return 4096 + ((size_t)(K1 + K2) / 64 + 1) * M * N * sizeof(float);
}
extern "C" int nabu_gemm2_f32(int M, int N, int K1, const float *A, int lda, const float *B, int ldb, int K2,
                              const float *A2, int lda2, const float *B2, int ldb2, float beta, float *C, int ldc,
                              const float *bias, void *ws, size_t ws_bytes, nabu_stream_t stream) {
  NABU_CHECK_ARG(M > 0 && N > 0 && K1 > 0 && K2 >= 0, "gemm2: bad dimensions");
  NABU_CHECK_ARG(A && B && C && ws && (K2 == 0 || (A2 && B2)), "gemm2: null pointer");
  NABU_CHECK_ARG(N / 32 * sizeof(unsigned) <= 4096, "gemm2: N too large");
  if (ws_bytes < nabu_gemm2_ws_bytes(M, N, K1, K2)) return fail(NABU_EWS, "gemm2: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  NABU_HIP(hipMemsetAsync(ws, 0, 4096, s));
  return gemm_skinny_fused(M, N, K1, A, lda, B, ldb, K2, A2, lda2, B2, ldb2, beta, C, ldc, bias,
                           reinterpret_cast<float *>(static_cast<char *>(ws) + 4096), static_cast<unsigned *>(ws), s);
}

// bf16-resident operands (gemm_bf16_pre.hip)
extern "C" int nabu_cvt_bf16(size_t R, int C, const float *src, int ld, void *dst_bf16, int ldd, int transpose,
                             nabu_stream_t stream) {
  NABU_CHECK_ARG(src && dst_bf16 && C > 0, "cvt_bf16: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (transpose) return cvt_bf16_t((int)R, C, src, ld, static_cast<unsigned short *>(dst_bf16), ldd, s);
  return cvt_bf16(R, C, src, ld, static_cast<unsigned short *>(dst_bf16), ldd, s);
}
extern "C" size_t nabu_gemm_bf16_nt_ws_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return gemm_bf16_pre_ws_bytes(M, N, K);
}
extern "C" int nabu_gemm_bf16_nt(int M, int N, int K, float alpha, const void *A_bf16, int lda, const void *B_bf16,
                                 int ldb, float beta, float *C, int ldc, const float *bias, void *ws, size_t ws_bytes,
                                 nabu_stream_t stream) {
  NABU_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "gemm_bf16_nt: bad dimensions");
  if (M == 0 || N == 0) return 0;
  NABU_CHECK_ARG(A_bf16 && B_bf16 && C, "gemm_bf16_nt: null pointer");
  return gemm_bf16_pre(M, N, K, alpha, static_cast<const unsigned short *>(A_bf16), lda,
                       static_cast<const unsigned short *>(B_bf16), ldb, beta, C, ldc, bias, ws, ws_bytes,
                       static_cast<hipStream_t>(stream));
}

static int g_default_precision = 0;   // 0 = not initialised yet

extern "C" int nabu_gemm_get_default_precision(void) {
  if (g_default_precision == 0) {
    g_default_precision = NABU_GEMM_F32;
    if (const char *e = getenv("NABU_GEMM_PRECISION")) {
      const std::string v(e);
      if (v == "bf16") g_default_precision = NABU_GEMM_BF16;
      else if (v == "bf16x3") g_default_precision = NABU_GEMM_BF16X3;
      else if (v == "bf16x6") g_default_precision = NABU_GEMM_BF16X6;
      else if (v == "f16x3") g_default_precision = NABU_GEMM_F16X3;
    }
  }
  return g_default_precision;
}

extern "C" int nabu_gemm_set_default_precision(int precision) {
  NABU_CHECK_ARG(precision >= NABU_GEMM_F32 && precision <= NABU_GEMM_F16X3, "gemm: unknown precision");
  g_default_precision = precision;
  return 0;
}

extern "C" int nabu_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                             const float *A, int lda, const float *B, int ldb, float beta,
                             float *C, int ldc, const float *bias, int kseg,
                             long long a_seg_stride, long long b_seg_stride, void *ws,
                             size_t ws_bytes, nabu_stream_t stream) {
  return nabu_gemm_ex(NABU_GEMM_DEFAULT, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, kseg,
                      a_seg_stride, b_seg_stride, ws, ws_bytes, stream);
}

static int gemm_run(int precision, int transA, int transB, int M, int N, int K, float alpha,
                    const float *A, int lda, const float *B, int ldb, float beta,
                    float *C, int ldc, const float *bias, int kseg,
                    long long a_seg_stride, long long b_seg_stride, void *ws,
                    size_t ws_bytes, nabu_stream_t stream);

// Shapes the fast fp32 kernel takes: reduction length a multiple of its 16-wide k-tile, 16-byte
// vector loads along the contiguous dimension of each operand (so only THAT dimension must be a
// multiple of 4; the other one is clamped at the edges).
static bool fast_f32_ok(bool ta, bool tb, int M, int N, int K, const void *A, int lda, const void *B, int ldb,
                        long long a_seg, long long b_seg) {
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return K > 0 && K % FKT == 0 && al16(A) && lda % 4 == 0 && a_seg % 4 == 0 && al16(B) && ldb % 4 == 0 &&
         b_seg % 4 == 0 && (!ta || M % 4 == 0) && (tb || N % 4 == 0);
}

// hipFuncSetAttribute is a per-device setting: remember, per calling thread, the devices a kernel family was
// configured on (a process-wide flag would skip the second GPU of a multi-device process, and an unsynchronised one
// races between threads)
static bool configured_on_current_device(unsigned long long &mask, int *dev_out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  *dev_out = dev;
  return dev < 64 && ((mask >> dev) & 1ull);
}

extern "C" int nabu_gemm_ex(int precision, int transA, int transB, int M, int N, int K, float alpha,
                            const float *A, int lda, const float *B, int ldb, float beta,
                            float *C, int ldc, const float *bias, int kseg,
                            long long a_seg_stride, long long b_seg_stride, void *ws,
                            size_t ws_bytes, nabu_stream_t stream) {
  NABU_CHECK_ARG(precision >= NABU_GEMM_DEFAULT && precision <= NABU_GEMM_F16X3, "gemm: unknown precision");
  if (precision == NABU_GEMM_DEFAULT) precision = nabu_gemm_get_default_precision();
  // f16x3 exists on packed operands only (gemm_pk.hip): products on row-major operands take the other
  // fp32-equivalent arithmetic
  if (precision == NABU_GEMM_F16X3) precision = NABU_GEMM_BF16X6;
  NABU_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return 0;
  NABU_CHECK_ARG(A && B && C, "gemm: null pointer");
  NABU_CHECK_ARG(kseg == 0 || (transA && !transB && K % kseg == 0),
                 "gemm: segmented K needs transA=1, transB=0 and K %% kseg == 0");
  // Reduction lengths that are not a multiple of the fast kernel's k-tile (B*T of real batches): the
  // fast kernel takes the first K - K % 16 indices, the generic kernel adds the rest.  With segmented K
  // the rest must lie inside the last segment (it does unless a segment is shorter than 16).
  const int Kt = K % FKT, Km = K - Kt;
  if (precision == NABU_GEMM_F32 && Kt != 0 && Km >= 4 * FKT && !(M <= 64 && !transA && !transB) &&
      (kseg == 0 || Km / kseg == (K - 1) / kseg) &&
      fast_f32_ok(transA != 0, transB != 0, M, N, Km, A, lda, B, ldb, a_seg_stride, b_seg_stride)) {
    if (int e = gemm_run(precision, transA, transB, M, N, Km, alpha, A, lda, B, ldb, beta, C, ldc, bias, kseg,
                         a_seg_stride, b_seg_stride, ws, ws_bytes, stream))
      return e;
    auto off = [&](bool kmajor, int ld, long long seg) -> size_t {     // offset of reduction index Km
      if (!kmajor) return (size_t)Km;
      return kseg > 0 ? (size_t)(Km / kseg) * (size_t)seg + (size_t)(Km % kseg) * ld : (size_t)Km * ld;
    };
    return gemm_run(precision, transA, transB, M, N, Kt, alpha, A + off(transA != 0, lda, a_seg_stride), lda,
                    B + off(transB == 0, ldb, b_seg_stride), ldb, 1.f, C, ldc, nullptr, 0, 0, 0, ws, ws_bytes, stream);
  }
  return gemm_run(precision, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, kseg,
                  a_seg_stride, b_seg_stride, ws, ws_bytes, stream);
}

static int gemm_run(int precision, int transA, int transB, int M, int N, int K, float alpha,
                    const float *A, int lda, const float *B, int ldb, float beta,
                    float *C, int ldc, const float *bias, int kseg,
                    long long a_seg_stride, long long b_seg_stride, void *ws,
                    size_t ws_bytes, nabu_stream_t stream) {
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = alpha; a.beta = beta;
  a.kseg = kseg; a.a_seg = a_seg_stride; a.b_seg = b_seg_stride;
  a.nbatch = 1; a.a_bs = a.b_bs = a.c_bs = 0;
  a.nsplit = K > 0 ? choose_split(M, N, K, &a.ksplit) : 1;
  if (K == 0) a.ksplit = BK;
  auto al16p = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // skinny products (decoder steps): exact fp32 whatever the requested precision
  const int skinny_kc = (!transA && !transB && kseg == 0 && al16p(A) && lda % 4 == 0) ? gemm_skinny_chunk(M, N, K) : 0;
  if (skinny_kc) {
    a.ksplit = skinny_kc;
    a.nsplit = K / skinny_kc;
  }
  a.partial = nullptr;
  if (a.nsplit > 1) {
    const size_t need = (size_t)a.nsplit * M * N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(NABU_EWS, "gemm: workspace %zu < %zu", ws_bytes, need);
    a.partial = static_cast<float *>(ws);
  }
  auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  a.vecA = al16(A) && lda % 4 == 0 && (a_seg_stride % 4 == 0);
  a.vecB = al16(B) && ldb % 4 == 0 && (b_seg_stride % 4 == 0);
  static int swz_env = -2;
  if (swz_env == -2) { const char *e = getenv("NABU_GEMM_SWIZZLE"); swz_env = e ? atoi(e) : -1; }
  a.swz = swz_env >= 0 ? swz_env : 0;
  if (const char *e = getenv("NABU_GEMM_NOSTAGE")) a.vecA |= 2 * atoi(e);   // timing experiments: 1 nothing, 2 no loads, 4 no LDS stores
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, a.nsplit), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool fast = M % 4 == 0 && N % 4 == 0 && K > 0 && K % FBK == 0 && a.vecA && a.vecB;      // bf16 kernels
  const bool fast32 = fast_f32_ok(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, a_seg_stride, b_seg_stride);
  if ((precision == NABU_GEMM_F32 || K % FBK != 0) && smallk_ok(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, kseg, C, ldc, bias)) {
    a.nsplit = 1; a.partial = nullptr;
    const size_t lds = smallk_lds(K);
    static thread_local unsigned long long configured = 0;
    int dev;
    if (!configured_on_current_device(configured, &dev)) {
      NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_smallk_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)smallk_lds(SMALLK_MAX)));
      if (dev < 64) configured |= 1ull << dev;
    }
    hipLaunchKernelGGL(gemm_smallk_kernel, dim3((N + BN - 1) / BN, (M + BM - 1) / BM), block, lds, s, a);
    NABU_LAUNCH_CHECK();
    return 0;
  }
  if (skinny_kc) {
    if (int e = gemm_skinny_launch(a, s)) return e;
  } else
  if (fast && precision != NABU_GEMM_F32) {
    if (int e = gemm_bf16_launch(a, transA != 0, transB != 0, precision - NABU_GEMM_BF16 + 1, grid, s)) return e;
  } else
  if (fast32) {
    const size_t lds = 4 * (size_t)(FKT * LDT > 128 * (FKT + 2) ? FKT * LDT : 128 * (FKT + 2)) * sizeof(float);
    static thread_local unsigned long long configured = 0;
    int dev;
    if (!configured_on_current_device(configured, &dev)) {
      const void *fns[4] = {reinterpret_cast<const void *>(gemm_f32_fast_kernel<true, true>),
                            reinterpret_cast<const void *>(gemm_f32_fast_kernel<true, false>),
                            reinterpret_cast<const void *>(gemm_f32_fast_kernel<false, true>),
                            reinterpret_cast<const void *>(gemm_f32_fast_kernel<false, false>)};
      for (const void *f : fns) NABU_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (dev < 64) configured |= 1ull << dev;
    }
    if (transA && transB) hipLaunchKernelGGL((gemm_f32_fast_kernel<true, true>), grid, block, lds, s, a);
    else if (transA) hipLaunchKernelGGL((gemm_f32_fast_kernel<true, false>), grid, block, lds, s, a);
    else if (transB) hipLaunchKernelGGL((gemm_f32_fast_kernel<false, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gemm_f32_fast_kernel<false, false>), grid, block, lds, s, a);
  } else
  if (transA && transB) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, a);
  else if (transA) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, a);
  else if (transB) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, a);
  NABU_LAUNCH_CHECK();
  if (a.nsplit > 1) {
    const size_t total = (size_t)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), block, 0, s, a);
    NABU_LAUNCH_CHECK();
  }
  return 0;
}
