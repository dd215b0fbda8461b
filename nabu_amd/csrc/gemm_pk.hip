// gemm_pk.hip — dense products on the bf16 matrix pipe over PACKED bf16-plane operands.
//
//   C[M,N] (fp32) = alpha * sum_k A[m,k] * B[n,k] + beta*C + bias[n]
//
// Two arithmetics on the same kernel (template parameter NP):
//   NP = 3  "bf16x6": every fp32 operand value x is held as three bf16 planes x = h + m + l (h = rne(x),
//           m = rne(x - h), l = x - h - m: the three significands hold all 24 bits of x, the split is exact),
//           and a product is the six plane products h·h, h·m, m·h, m·m, h·l, l·h (what is dropped is below
//           2^-26 |a||b| per term) on v_mfma_f32_32x32x16_bf16, the 16-k partial sums promoted to the fp32
//           accumulators by the VALU (see PROMOTED ACCUMULATION in the kernel): fp32-equivalent results from
//           the pipe that is 16 times faster than v_mfma_f32_32x32x2_f32;
//   NP = 1  plain bf16 operands (plane h only), fp32 accumulation: BASELINE.json configs[4]'s "bf16 MFMA
//           input-to-hidden GEMMs".
//   NP = 2  "f16x3": every operand ROW (an M resp. N index: all its k) carries a power-of-two scale s that
//           puts the row's largest magnitude into [2^14, 2^15); the scaled value is held as two fp16 planes
//           x·s = h + l (h = rne(x·s), l = rne(x·s - h): 22-24 significant bits, absolute error below
//           2^-40 of the row's maximum) and a product is the three plane products l·h, h·l, h·h on
//           v_mfma_f32_32x32x16_f16, promoted to the fp32 accumulators per 16 k as for NP = 3; the epilogue
//           multiplies by 1 / (s_a[m] s_b[n]).  Half the matrix instructions of bf16x6 at the same error
//           level against float64 (tests/test_hip_gemm_pk.py); a stage is 4 pieces (32 KiB), ring of three.
//           The row maxima come from pk_amax_kernel (one read of the source) or are known a priori
//           (LSTM outputs: |h| < 1).
//
// Why packed operands.  Round 2's kernels either split fp32 operands inside the k-loop (gemm_bf16.hip:
// 4-byte operand traffic and ~300 VALU instructions per tile on the critical path, 140-160 TF/s effective)
// or used 128 x 128 tiles on row-major bf16 copies (gemm_bf16_pre.hip: 64 flop per operand byte, matrix pipe
// 27 % busy, every operand panel fetched ~5 times).  Here every operand is converted ONCE per use-site into
// the layout the tile loop wants (pack kernels below, HBM-streaming, ~10 bytes of traffic per element):
//
//   packed[kb][plane][row][16 k]   bf16, row = the operand's M (resp. N) index padded to a multiple of 256,
//                                  kb = k / 16; inside a row's 32 bytes the two 16-byte halves are swapped
//                                  when bit 3 of the row index is set (LDS bank swizzle baked into memory).
//
// A 256-row tile of one (kb, plane) is then ONE contiguous 8 KiB piece of memory, and it is exactly the LDS
// image the MFMA operand reads want: staging is a linear LDS-DMA copy (buffer_load_dwordx4 ... lds, 1 KiB per
// wave instruction, no registers, no address arithmetic beyond a scalar add per stage, every byte of every
// 128-byte line used), and the operand reads are conflict-free ds_read_b128 (row stride 32 B + the half swap:
// the 16 lanes of a read group cover all 64 banks).
//
// Tile loop: 256 x 256 output tile per 512-thread workgroup (8 waves: 2 along M x 4 along N, each
// 128 x 64 = 4 x 2 MFMA tiles, 128 accumulator registers), one workgroup per CU.  A STAGE is 6 pieces of
// 8 KiB (NP = 3: one k-block of 16 x 3 planes x {A, B}; NP = 1: three k-blocks x {A, B}) = 48 KiB, ring of
// three stages = 144 KiB of LDS.  The two waves of a SIMD run half a stage apart (waves 4-7 pass one extra
// barrier at the start): while one is in its COMPUTE phase (48 / 24 back-to-back MFMAs = 1536 / 768 cycles,
// no memory instruction at all) its partner is in its LOAD phase (issue the 6 LDS-DMA pieces of stage t+2,
// 18 operand reads of stage t) — matrix beside memory on every SIMD, two barriers per stage.
// LDS-DMA completion is counted by hand (s_waitcnt vmcnt(6): everything but the pieces just issued), the
// loads are invisible to hipcc (inline asm, no destination registers), so no compiler wait ever drains them.
// Ring discipline (t = stage, barrier numbering in the kernel):
//   RAW  pieces of stage t+2 are issued in LOAD(t), waited for at the end of LOAD(t+1) by the issuing wave,
//        which then passes a barrier before anybody reads them in LOAD(t+2);
//   WAR  stage t+2 lands in the buffer of stage t-1, whose last reads (LOAD(t-1) of the late half) are
//        retired (lgkmcnt(0)) before the barrier that precedes LOAD(t) of the early half.
//
// XCD-aware order: workgroup id -> XCD is id % 8; every XCD walks a contiguous range of the tile sequence,
// which goes row-fastest through bands of 4 row tiles, so the 32 tiles in flight on an XCD are ~4 x 8 tiles
// that share 12 operand panels in that XCD's L2 and move along k together.
//
// Replaces: the tf MatMul of the LSTM cell's input part and its autodiff, time-batched
// (nabu/neuralnetworks/components/layer.py:35-47, nabu/neuralnetworks/trainers/trainer.py:556-558).
#include "gemm_args.h"
#include "gemm_pk_asm.inc"

#include <stdlib.h>

namespace nabu {

typedef __bf16 kbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 kf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 kf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 kbf16x2 __attribute__((ext_vector_type(2)));
typedef float kf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ku32x4 __attribute__((ext_vector_type(4)));
typedef int ki32x4 __attribute__((ext_vector_type(4)));

constexpr int PK_T = 256;               // tile edge (rows of A = M index, rows of B = N index)
constexpr int PK_CH = PK_T * 32;        // one piece: 256 rows x 16 k x 2 bytes
constexpr int PK_STAGE = 6 * PK_CH;     // 48 KiB
constexpr int PK_LDS = 3 * PK_STAGE;    // 144 KiB
#ifndef PK_RING2
#define PK_RING2 3                      // f16x3: stages in the LDS ring (3 or 4)
#endif
constexpr int PK_STAGE2 = 4 * PK_CH;    // f16x3: 32 KiB
constexpr int PK_LDS2 = PK_RING2 * PK_STAGE2;
constexpr int PK_GM = 4;                // row tiles per band of the tile order

struct PkOp {
  const char *base[2];        // per batch entry
  unsigned long long kb_stride;   // bytes from k-block kb to kb+1 (= planes stored * rows_pad * 32)
  unsigned plane_stride;          // rows_pad * 32
};

struct PkArgs {
  PkOp A, B;
  int M, N;                   // logical output size per batch entry
  int tiles_m, tiles_n, nbatch;
  int nkb;                    // k-blocks to reduce over (multiple of the stage's k-blocks)
  int nsplit, kb_per_split;
  float *C[2], *C2[2];        // per batch entry; columns >= n_split go to C2 (column n - n_split)
  int ldc, n_split;
  const float *bias, *bias2;  // bias2: columns >= n_split
  float alpha, beta;
  float *partial;             // [nsplit][nbatch][M][N] when nsplit > 1
  const unsigned *a_amax[2], *b_amax[2];   // f16x3: bit patterns of the packed rows' largest magnitudes (per batch entry)
  int throttle;               // experiment (NABU_PK_THROTTLE): s_sleep units per stage
};

// f16x3 row scales, derived from the bit pattern of the row's largest magnitude wherever they are needed:
// scale = 2^(14 - floor(log2 amax)) puts amax into [2^14, 2^15) (fp16 overflows at 65504); exponent field clamped
// so that scale and inverse are normal numbers; an all-zero row takes amax = 1; inf / NaN rows keep a finite scale
// and propagate through the planes
__device__ __forceinline__ unsigned pk_amax_exp(unsigned bits) {
  unsigned e = (bits >> 23) & 0xFFu;
  if ((bits & 0x7FFFFFFFu) == 0) e = 127;
  return e < 15 ? 15 : (e > 253 ? 253 : e);
}
__device__ __forceinline__ float pk_scale_of(unsigned bits) { return __builtin_bit_cast(float, (268u - pk_amax_exp(bits)) << 23); }
__device__ __forceinline__ float pk_inv_scale_of(unsigned bits) { return __builtin_bit_cast(float, (pk_amax_exp(bits) - 14u) << 23); }

__device__ __forceinline__ ki32x4 pk_rsrc(unsigned long long a) {
  return (ki32x4){(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xFFFFu), -1, 0x00020000};
}

// six LDS-DMA pieces of one stage: 1 KiB per wave and piece.  m0 = LDS byte address of this wave's part of
// piece 0 (pieces are PK_CH apart); voff = tid * 16; A pieces at soffsets 0, sa1, sa2 of ra, B pieces of rb.
__device__ __forceinline__ void pk_stage_issue(unsigned m0, unsigned voff, ki32x4 ra, ki32x4 rb, unsigned sa1,
                                               unsigned sa2, unsigned sb1, unsigned sb2) {
  const unsigned m1 = m0 + PK_CH, m2 = m0 + 2 * PK_CH, m3 = m0 + 3 * PK_CH, m4 = m0 + 4 * PK_CH, m5 = m0 + 5 * PK_CH;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %7, 0 offen lds\n\t"
      "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %7, %9 offen lds\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %7, %10 offen lds\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, 0 offen lds\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %11 offen lds\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tbuffer_load_dwordx4 %6, %8, %12 offen lds"
      :
      : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "v"(voff), "s"(ra), "s"(rb), "s"(sa1), "s"(sa2),
        "s"(sb1), "s"(sb2)
      : "memory");
}

// f16x3: four pieces (A h, A l, B h, B l)
__device__ __forceinline__ void pk_stage_issue4(unsigned m0, unsigned voff, ki32x4 ra, ki32x4 rb, unsigned sa1,
                                                unsigned sb1) {
  const unsigned m1 = m0 + PK_CH, m2 = m0 + 2 * PK_CH, m3 = m0 + 3 * PK_CH;
  asm volatile(
      "s_nop 4\n\t"
      "s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, 0 offen lds\n\t"
      "s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %7 offen lds\n\t"
      "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, 0 offen lds\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %6, %8 offen lds"
      :
      : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "v"(voff), "s"(ra), "s"(rb), "s"(sa1), "s"(sb1)
      : "memory");
}

template <int N>
__device__ __forceinline__ void pk_wait() {   // LDS-DMA pieces but the last N landed; operand reads retired
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pk_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int NP, int VAR>
__global__ __launch_bounds__(512) void gemm_pk_kernel(PkArgs p) {
  extern __shared__ __attribute__((aligned(16))) char pk_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = w >> 2, wn = w & 3;

  // ---- work item of this workgroup: XCD-contiguous ranges of (split, batch, band, column, row) ----
  const int total = gridDim.x, wg = blockIdx.x;
  int t;
  {
    const int per = total / 8, rem = total % 8, xcd = wg % 8, local = wg / 8;
    t = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + local;
  }
  const int per_batch = p.tiles_m * p.tiles_n, ntiles = per_batch * p.nbatch;
  const int split = t / ntiles;
  int tile = t - split * ntiles;
  const int batch = tile / per_batch;
  tile -= batch * per_batch;
  const int band = PK_GM * p.tiles_n, g = tile / band, first = g * PK_GM;
  const int rows = min(p.tiles_m - first, PK_GM), r = tile - g * band;
  const int tm = first + r % rows, tn = r / rows;

  const int kb0 = split * p.kb_per_split;
  const int kbn = min(p.nkb, kb0 + p.kb_per_split) - kb0;
  constexpr int KBS = NP == 1 ? 3 : 1;          // k-blocks per stage
  constexpr int NPIECE = NP == 2 ? 4 : 6;       // 8 KiB pieces per stage
  constexpr unsigned STAGE = NPIECE * PK_CH;
  const int nst = kbn / KBS;

  unsigned long long pa = reinterpret_cast<unsigned long long>(p.A.base[batch]) + (unsigned long long)tm * PK_CH +
                          (unsigned long long)kb0 * p.A.kb_stride;
  unsigned long long pb = reinterpret_cast<unsigned long long>(p.B.base[batch]) + (unsigned long long)tn * PK_CH +
                          (unsigned long long)kb0 * p.B.kb_stride;
  const unsigned long long step_a = (unsigned long long)KBS * p.A.kb_stride, step_b = (unsigned long long)KBS * p.B.kb_stride;
  const unsigned sa1 = NP != 1 ? p.A.plane_stride : (unsigned)p.A.kb_stride, sa2 = 2 * sa1;
  const unsigned sb1 = NP != 1 ? p.B.plane_stride : (unsigned)p.B.kb_stride, sb2 = 2 * sb1;
  const unsigned voff = (unsigned)tid * 16u;
  const unsigned wbase = (unsigned)w * 1024u;     // this wave's part of a piece
  auto issue = [&](unsigned m0) {
    if constexpr (NP == 2) pk_stage_issue4(m0, voff, pk_rsrc(pa), pk_rsrc(pb), sa1, sb1);
    else pk_stage_issue(m0, voff, pk_rsrc(pa), pk_rsrc(pb), sa1, sa2, sb1, sb2);
    pa += step_a; pb += step_b;
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  f32x16 tpend = acc[0][0], t0 = acc[0][0];   // promoted accumulation: scratch tiles; tpend = the partial sum whose add is still owed
  f32x16 ts[4] = {acc[0][0], acc[0][0], acc[0][0], acc[0][0]};   // f16x3: four scratch tiles (gemm_pk_asm.inc)

  // operand read offsets inside a stage: row (lane & 31) of a 32-row MFMA tile, half (lane >> 5) swapped by
  // bit 3 of the row (tile bases are multiples of 32 rows, so bit 3 of the row is bit 3 of the lane)
  const unsigned laneoff = (unsigned)(lane & 31) * 32u + (unsigned)(((lane >> 5) ^ ((lane >> 3) & 1)) * 16);
  const unsigned offA = laneoff + (unsigned)grp * (128u * 32u);
  const unsigned offB = laneoff + (unsigned)(NPIECE / 2) * PK_CH + (unsigned)wn * (64u * 32u);

  // ---- prologue: stages 0 .. RING-2 ----
  constexpr int RING = NP == 2 ? PK_RING2 : 3;
  if (nst > 0) issue(wbase);
  if (nst > 1) issue(STAGE + wbase);
  if (RING == 4 && nst > 2) issue(2 * STAGE + wbase);
  if (RING == 4 && nst > 2) pk_wait<2 * NPIECE>();
  else if (nst > 1) pk_wait<NPIECE>();
  else pk_wait<0>();
  pk_barrier();                 // barrier 0: stage 0 visible
  if (grp == 1) pk_barrier();   // the late half runs one barrier interval behind

  unsigned so_rd = 0, so_wr = (RING - 1) * STAGE;
  for (int st = 0; st < nst; ++st) {
    // -------- LOAD(st) --------
    const bool more = st + 2 < nst;               // a stage beyond st+1 is (or is being) fetched
    const bool more2 = RING == 4 && st + 3 < nst;   // ... two of them
    if (st + RING - 1 < nst) issue(so_wr + wbase);
    kbf16x8 fa[4][3], fb[2][3];
    kf16x8 ha[4][2], hb[2][2];
    if constexpr (NP == 3 && VAR == 1) {
      // hand-scheduled path: the operand reads land in the physical registers the COMPUTE stream names
      // (gemm_pk_asm.inc); reads and their wait are ONE statement, so no compiler copy can see a register
      // before its data
      const unsigned va = so_rd + offA, vb = so_rd + offB;
      if (more) asm volatile(PK_STREAM_LOAD "s_waitcnt vmcnt(6) lgkmcnt(0)" : PK_ASM_LOAD_OUTPUTS : "v"(va), "v"(vb) : "memory");
      else asm volatile(PK_STREAM_LOAD "s_waitcnt vmcnt(0) lgkmcnt(0)" : PK_ASM_LOAD_OUTPUTS : "v"(va), "v"(vb) : "memory");
    } else if constexpr (NP == 2) {
      const unsigned va = so_rd + offA, vb = so_rd + offB;
      // all but the stages beyond st+1 have landed when the reads of this one retire
      if (more2) asm volatile(PK2_STREAM_LOAD "s_waitcnt vmcnt(8) lgkmcnt(0)" : PK2_ASM_LOAD_OUTPUTS : "v"(va), "v"(vb) : "memory");
      else if (more) asm volatile(PK2_STREAM_LOAD "s_waitcnt vmcnt(4) lgkmcnt(0)" : PK2_ASM_LOAD_OUTPUTS : "v"(va), "v"(vb) : "memory");
      else asm volatile(PK2_STREAM_LOAD "s_waitcnt vmcnt(0) lgkmcnt(0)" : PK2_ASM_LOAD_OUTPUTS : "v"(va), "v"(vb) : "memory");
    } else {
      const char *sA = pk_smem + so_rd + offA, *sB = pk_smem + so_rd + offB;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][q] = *reinterpret_cast<const kbf16x8 *>(sA + q * PK_CH + i * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j][q] = *reinterpret_cast<const kbf16x8 *>(sB + q * PK_CH + j * 1024);
      }
      if (more) pk_wait<6>(); else pk_wait<0>();
    }
    pk_barrier();
    // -------- COMPUTE(st) --------
    constexpr bool hand = (NP == 3 && VAR == 1) || (NP == 2 && VAR == 1);
    if (!hand) __builtin_amdgcn_s_setprio(1);
    if constexpr (NP == 2 && VAR == 0) {
      // direct accumulation (A/B measurements: NABU_PK_VAR=0)
#define PK_PROD(qa, qb)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb[j][qb], ha[i][qa], acc[i][j], 0, 0, 0);
      PK_PROD(1, 0) PK_PROD(0, 1) PK_PROD(0, 0)
#undef PK_PROD
    } else if constexpr (NP == 2) {
      // f16x3, promoted accumulation: chains of three MFMAs (l.h, h.l, h.h) from 0 in four rotating scratch tiles,
      // the 16 adds of a tile spread over the next three MFMA gaps (tools/gen_pk_asm.py, stream_f16)
      asm volatile("s_setprio 1\n\t" PK2_STREAM "s_setprio 0" : PK2_ASM_COMPUTE_INOUT : PK2_ASM_COMPUTE_INPUTS);
    } else if (NP == 3 && VAR == 0) {
      // direct accumulation (kept for A/B measurements: NABU_PK_VAR=0)
#define PK_PROD(qa, qb)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][qb], fa[i][qa], acc[i][j], 0, 0, 0);
      PK_PROD(2, 0) PK_PROD(0, 2) PK_PROD(1, 1) PK_PROD(1, 0) PK_PROD(0, 1) PK_PROD(0, 0)
#undef PK_PROD
    } else if (NP == 3) {
      // PROMOTED ACCUMULATION.  The six plane products of one (tile, k-block) are chained in a scratch
      // accumulator that starts at 0 (smallest terms first), and the finished 16-k partial sum is added to the
      // tile's accumulator by the VALU: one RNE rounding at the accumulator's magnitude per 16 k, where the
      // exact-fp32 kernel's fma chain has 16 and a direct MFMA chain 6 (each with the matrix pipe's own
      // rounding).  The instruction stream is hand-scheduled (tools/gen_pk_asm.py, schedule S1: the 16 adds of
      // tile n-1 in the gaps of tile n's MFMA chain; hipcc clusters them and loses 14 % of the matrix pipe).
      asm volatile("s_setprio 1\n\t" PK_STREAM_S1 "s_setprio 0" : PK_ASM_COMPUTE_INOUT : PK_ASM_COMPUTE_INPUTS);
    } else {
#define PK_PROD(qa, qb)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)               \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][qb], fa[i][qa], acc[i][j], 0, 0, 0);
      PK_PROD(0, 0) PK_PROD(1, 1) PK_PROD(2, 2)
#undef PK_PROD
    }
    if (!hand) __builtin_amdgcn_s_setprio(0);
    for (int z = 0; z < p.throttle; ++z) __builtin_amdgcn_s_sleep(1);
    pk_barrier();
    so_rd = so_rd == (RING - 1) * STAGE ? 0 : so_rd + STAGE;
    so_wr = so_wr == (RING - 1) * STAGE ? 0 : so_wr + STAGE;
  }
  if (grp == 0) pk_barrier();
  if (NP == 3 && VAR == 1) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(tpend));   // the last chain's matrix result -> VALU read: not padded by hipcc
    acc[3][1] += tpend;
  }
  if (NP == 2 && VAR == 1) {
    // the adds the stream still owes: tile 6 = (3, 0) its last elements out of T2, tile 7 = (3, 1) out of T3
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(ts[2]), "+v"(ts[3]));
#pragma unroll
    for (int q = PK2_PEND_FIRST; q < 16; ++q) acc[3][0][q] += ts[2][q];
    acc[3][1] += ts[3];
  }

  // ---- epilogue.  The products were issued with the B fragment as the matrix instruction's first operand, so a
  // 32 x 32 result tile is held transposed: lane -> row m = lane & 31, register quad g -> the four consecutive
  // columns n = 8 g + 4 (lane >> 5) + {0..3}: one 16-byte store per quad (32 rows x 32 contiguous bytes per
  // instruction) instead of four 4-byte stores.
  const int mrow = lane & 31, nquad = 4 * (lane >> 5);
  const int m_t = tm * PK_T + grp * 128, n_t = tn * PK_T + wn * 64;
  float *Cb;
  const float *bias = nullptr;
  int n_off = 0, ldc;
  if (p.nsplit == 1) {
    Cb = p.C[batch]; bias = p.bias; ldc = p.ldc;
    if (p.n_split > 0 && n_t >= p.n_split) { Cb = p.C2[batch]; bias = p.bias2; n_off = p.n_split; }
  } else {
    Cb = p.partial + ((size_t)split * p.nbatch + batch) * (size_t)p.M * p.N;
    ldc = p.N;
  }
  const bool plain = p.nsplit > 1 || (p.alpha == 1.f && p.beta == 0.f);
  const bool scaled = NP == 2 && p.nsplit == 1;   // f16x3: undo the row scales here (split-K: in the reduce pass)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_t + i * 32 + mrow;
    if (m >= p.M) continue;
    float *crow = Cb + (size_t)m * ldc - n_off;
    const float ra = scaled ? pk_inv_scale_of(p.a_amax[batch][m]) : 1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n_t + j * 32 + 8 * g + nquad;
        if (n >= p.N) continue;                      // N % 4 == 0: a quad is inside or outside as a whole
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        if (scaled) {
          const ku32x4 bb = *reinterpret_cast<const ku32x4 *>(p.b_amax[batch] + n);
          v.x *= ra * pk_inv_scale_of(bb.x); v.y *= ra * pk_inv_scale_of(bb.y);
          v.z *= ra * pk_inv_scale_of(bb.z); v.w *= ra * pk_inv_scale_of(bb.w);
        }
        if (p.nsplit == 1) {
          if (!plain) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
          if (bias) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias + n - n_off);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          if (p.beta != 0.f) {
            const float4 o = *reinterpret_cast<const float4 *>(crow + n);
            v.x += p.beta * o.x; v.y += p.beta * o.y; v.z += p.beta * o.z; v.w += p.beta * o.w;
          }
        }
        *reinterpret_cast<float4 *>(crow + n) = v;
      }
  }
}

// split-K: fixed-order sum of the partial slabs, then the same epilogue as above
__global__ __launch_bounds__(256) void gemm_pk_reduce_kernel(PkArgs p) {
  const size_t per = (size_t)p.M * p.N, total = per * p.nbatch, n4 = total / 4;   // N % 4 == 0 (checked by the host)
  for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (size_t)gridDim.x * 256) {
    const size_t i = i4 * 4;
    const int batch = (int)(i / per);
    const size_t e = i - (size_t)batch * per;
    const int m = (int)(e / p.N), n = (int)(e % p.N);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < p.nsplit; ++z) {
      const float4 v = *reinterpret_cast<const float4 *>(p.partial + (size_t)z * total + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (p.a_amax[0]) {   // f16x3 operands: undo the row scales
      const float ra = pk_inv_scale_of(p.a_amax[batch][m]);
      const ku32x4 bb = *reinterpret_cast<const ku32x4 *>(p.b_amax[batch] + n);
      s.x *= ra * pk_inv_scale_of(bb.x); s.y *= ra * pk_inv_scale_of(bb.y);
      s.z *= ra * pk_inv_scale_of(bb.z); s.w *= ra * pk_inv_scale_of(bb.w);
    }
    float *Cb = p.C[batch];
    const float *bias = p.bias;
    int n_off = 0;
    if (p.n_split > 0 && n >= p.n_split) { Cb = p.C2[batch]; bias = p.bias2; n_off = p.n_split; }
    float *c = Cb + (size_t)m * p.ldc + (n - n_off);
    float4 o = make_float4(p.alpha * s.x, p.alpha * s.y, p.alpha * s.z, p.alpha * s.w);
    if (bias) {
      const float4 b = *reinterpret_cast<const float4 *>(bias + n - n_off);
      o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w;
    }
    if (p.beta != 0.f) {
      const float4 old = *reinterpret_cast<const float4 *>(c);
      o.x += p.beta * old.x; o.y += p.beta * old.y; o.z += p.beta * old.z; o.w += p.beta * old.w;
    }
    *reinterpret_cast<float4 *>(c) = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// pack kernels: fp32 -> bf16 planes in the layout above.
__device__ __forceinline__ unsigned pk_cvt2(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((kf32x2){a, b}, kbf16x2));
}
__device__ __forceinline__ float pk_hi(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float pk_lo(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }

// f16x3: 16 consecutive k of one packed row, scaled -> planes h, l (fp16, round to nearest even)
__device__ __forceinline__ unsigned pk_cvt2h(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((kf32x2){a, b}, kf16x2));
}
__device__ __forceinline__ void pk_split_store_f16(const float *x, char *dst, unsigned plane_stride, int row, float scale) {
  unsigned h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = x[2 * i] * scale, b = x[2 * i + 1] * scale;       // exact (power of two)
    h[i] = pk_cvt2h(a, b);
    const kf16x2 hv = __builtin_bit_cast(kf16x2, h[i]);
    l[i] = pk_cvt2h(a - (float)hv[0], b - (float)hv[1]);               // the difference is exact in fp32
  }
  const int sw = (row >> 3) & 1;
  ku32x4 *d = reinterpret_cast<ku32x4 *>(dst), *d1 = reinterpret_cast<ku32x4 *>(dst + plane_stride);
  d[sw] = (ku32x4){h[0], h[1], h[2], h[3]};
  d[sw ^ 1] = (ku32x4){h[4], h[5], h[6], h[7]};
  d1[sw] = (ku32x4){l[0], l[1], l[2], l[3]};
  d1[sw ^ 1] = (ku32x4){l[4], l[5], l[6], l[7]};
}

// 16 consecutive k of one packed row -> NP planes of 32 bytes (halves swapped when bit 3 of the row is set)
template <int NP>
__device__ __forceinline__ void pk_split_store(const float *x, char *dst, unsigned plane_stride, int row,
                                               const unsigned *amax = nullptr) {
  if constexpr (NP == 2) {
    pk_split_store_f16(x, dst, plane_stride, row, pk_scale_of(amax[row]));
    return;
  }
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = x[2 * i], b = x[2 * i + 1];
    h[i] = pk_cvt2(a, b);
    if (NP == 3) {
      const float ra = a - pk_hi(h[i]), rb = b - pk_lo(h[i]);     // exact
      m[i] = pk_cvt2(ra, rb);
      l[i] = pk_cvt2(ra - pk_hi(m[i]), rb - pk_lo(m[i]));          // exact, fits 8 bits
    }
  }
  const int sw = (row >> 3) & 1;
  ku32x4 *d = reinterpret_cast<ku32x4 *>(dst);
  d[sw] = (ku32x4){h[0], h[1], h[2], h[3]};
  d[sw ^ 1] = (ku32x4){h[4], h[5], h[6], h[7]};
  if (NP == 3) {
    ku32x4 *d1 = reinterpret_cast<ku32x4 *>(dst + plane_stride), *d2 = reinterpret_cast<ku32x4 *>(dst + 2 * (size_t)plane_stride);
    d1[sw] = (ku32x4){m[0], m[1], m[2], m[3]};
    d1[sw ^ 1] = (ku32x4){m[4], m[5], m[6], m[7]};
    d2[sw] = (ku32x4){l[0], l[1], l[2], l[3]};
    d2[sw ^ 1] = (ku32x4){l[4], l[5], l[6], l[7]};
  }
}

struct PackArgs {
  const unsigned *amax;       // f16x3: largest magnitude (bit pattern) of every packed row of the destination
  const float *src;
  long long ld;
  int R, C;                   // valid source rows / columns
  int fill_rows, fill_kb;     // packed rows / k-blocks written (zeros beyond the source)
  char *dst;                  // packed buffer
  unsigned long long kb_stride;
  unsigned plane_stride;
  int row_off, kb_off;        // where this matrix sits in the packed operand
  int period, shift;          // transposed form: source row of reduction index r is r + shift when
                              // 0 <= r % period + shift < period, else the value is 0 (period 0: no shift)
};

// reduction index contiguous in the source: packed row = source row, k = source column.
// grid (ceil(fill_kb / 4), ceil(fill_rows / 64)), 256 threads: a 64 x 64 tile through LDS
template <int NP>
__device__ __forceinline__ void pk_pack_rows_body(const PackArgs &a, float (*tile)[68]) {
  const int tid = threadIdx.x, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (r0 >= a.fill_rows || (int)blockIdx.x * 4 >= a.fill_kb) return;      // (uniform: a batched launch's spare blocks)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < a.R) {
      const float *s = a.src + (size_t)(r0 + r) * a.ld + c0 + c4;
      if (c0 + c4 + 3 < a.C) v = *reinterpret_cast<const float4 *>(s);
      else {
        if (c0 + c4 < a.C) v.x = s[0];
        if (c0 + c4 + 1 < a.C) v.y = s[1];
        if (c0 + c4 + 2 < a.C) v.z = s[2];
      }
    }
    *reinterpret_cast<float4 *>(&tile[r][c4]) = v;
  }
  __syncthreads();
  const int r = tid & 63, kbl = tid >> 6;
  const int row = r0 + r, kb = blockIdx.x * 4 + kbl;
  if (row >= a.fill_rows || kb >= a.fill_kb) return;
  float x[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 v = *reinterpret_cast<const float4 *>(&tile[r][kbl * 16 + 4 * i]);
    x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
  }
  const int prow = a.row_off + row;
  pk_split_store<NP>(x, a.dst + (size_t)(a.kb_off + kb) * a.kb_stride + (size_t)prow * 32, a.plane_stride, prow, a.amax);
}
template <int NP>
__global__ __launch_bounds__(256) void pk_pack_rows_kernel(PackArgs a) {
  __shared__ __attribute__((aligned(16))) float tile[64][68];
  pk_pack_rows_body<NP>(a, tile);
}
// up to PK_NREQ packs of one kind in ONE launch (blockIdx.z = request; the grid covers the largest): the two cells'
// weights, the two directions' h^T — each a few microseconds of work behind its own dispatch otherwise
constexpr int PK_NREQ = 4;
struct PackArgsN { PackArgs r[PK_NREQ]; };
template <int NP>
__global__ __launch_bounds__(256) void pk_pack_rows_multi_kernel(PackArgsN a) {
  __shared__ __attribute__((aligned(16))) float tile[64][68];
  pk_pack_rows_body<NP>(a.r[blockIdx.z], tile);
}

// reduction index = source ROW: packed row = source column, k = source row (transposed copy).
// grid (ceil(fill_rows / 64) over source columns, ceil(fill_kb / 4) over source rows)
template <int NP>
__device__ __forceinline__ void pk_pack_cols_body(const PackArgs &a, float (*tile)[65]) {
  const int tid = threadIdx.x, c0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  if (c0 >= a.fill_rows || (int)blockIdx.y * 4 >= a.fill_kb) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, kk = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    long long sr = (long long)k0 + kk;
    bool ok = sr < a.R;
    if (a.period > 0) {
      const int ph = (int)(sr % a.period) + a.shift;
      ok = ok && ph >= 0 && ph < a.period;
      sr += a.shift;
    }
    if (ok) {
      const float *s = a.src + (size_t)sr * a.ld + c0 + c4;
      if (c0 + c4 + 3 < a.C) v = *reinterpret_cast<const float4 *>(s);
      else {
        if (c0 + c4 < a.C) v.x = s[0];
        if (c0 + c4 + 1 < a.C) v.y = s[1];
        if (c0 + c4 + 2 < a.C) v.z = s[2];
      }
    }
    tile[kk][c4] = v.x; tile[kk][c4 + 1] = v.y; tile[kk][c4 + 2] = v.z; tile[kk][c4 + 3] = v.w;
  }
  __syncthreads();
  const int c = tid & 63, kbl = tid >> 6;
  const int row = c0 + c, kb = blockIdx.y * 4 + kbl;
  if (row >= a.fill_rows || kb >= a.fill_kb) return;
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = tile[kbl * 16 + i][c];
  const int prow = a.row_off + row;
  pk_split_store<NP>(x, a.dst + (size_t)(a.kb_off + kb) * a.kb_stride + (size_t)prow * 32, a.plane_stride, prow, a.amax);
}
template <int NP>
__global__ __launch_bounds__(256) void pk_pack_cols_kernel(PackArgs a) {
  __shared__ float tile[64][65];
  pk_pack_cols_body<NP>(a, tile);
}
template <int NP>
__global__ __launch_bounds__(256) void pk_pack_cols_multi_kernel(PackArgsN a) {
  __shared__ float tile[64][65];
  pk_pack_cols_body<NP>(a.r[blockIdx.z], tile);
}

// BOTH layouts of one source matrix from ONE read (the backward pass needs dz as [BT rows, k = gate column] for the
// input gradient and as [gate-column rows, k = frame] for the weight gradients): a 64 x 64 tile through LDS, then the
// natural pack (row = source row) and the transposed pack (row = source column) of it.
// grid (ceil(C / 64), ceil(R / 64)); a = natural destination, b = transposed destination (their src/ld/R/C are equal)
template <int NP>
__global__ __launch_bounds__(256) void pk_pack_both_kernel(PackArgs a, PackArgs b) {
  __shared__ __attribute__((aligned(16))) float tile[64][68];
  const int tid = threadIdx.x, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 256 * j, r = i >> 4, c4 = (i & 15) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < a.R) {
      const float *s = a.src + (size_t)(r0 + r) * a.ld + c0 + c4;
      if (c0 + c4 + 3 < a.C) v = *reinterpret_cast<const float4 *>(s);
      else {
        if (c0 + c4 < a.C) v.x = s[0];
        if (c0 + c4 + 1 < a.C) v.y = s[1];
        if (c0 + c4 + 2 < a.C) v.z = s[2];
      }
    }
    *reinterpret_cast<float4 *>(&tile[r][c4]) = v;
  }
  __syncthreads();
  const int l = tid & 63, kbl = tid >> 6;
  float x[16];
  {   // natural: packed row = source row r0 + l, k-block = source columns c0 + 16 kbl ...
    const int row = r0 + l, kb = blockIdx.x * 4 + kbl;
    if (row < a.fill_rows && kb < a.fill_kb) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4 *>(&tile[l][kbl * 16 + 4 * i]);
        x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
      }
      const int prow = a.row_off + row;
      pk_split_store<NP>(x, a.dst + (size_t)(a.kb_off + kb) * a.kb_stride + (size_t)prow * 32, a.plane_stride, prow, a.amax);
    }
  }
  {   // transposed: packed row = source column c0 + l, k-block = source rows r0 + 16 kbl ...
    const int row = c0 + l, kb = blockIdx.y * 4 + kbl;
    if (row < b.fill_rows && kb < b.fill_kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = tile[kbl * 16 + i][l];
      const int prow = b.row_off + row;
      pk_split_store<NP>(x, b.dst + (size_t)(b.kb_off + kb) * b.kb_stride + (size_t)prow * 32, b.plane_stride, prow, b.amax);
    }
  }
}

// largest magnitudes of the rows and / or the columns of a source matrix, as bit patterns (|x| compares like its
// bits; NaN compares largest, so a NaN row keeps its NaN): atomicMax into zero-initialised arrays.
// grid (ceil(C / 64), ceil(R / 256)), 256 threads: 64 columns x 256 rows per workgroup
// (blockIdx.z = 1: the second source / column array of nabu::pk_amax_pair — the two cells' kernels in one launch, the
// row maxima shared; cols_b: a second array that receives the same column maxima)
__global__ __launch_bounds__(256) void pk_amax_kernel(const float *src, long long ld, int R, int C, unsigned *rows,
                                                      unsigned *cols, const float *src1 = nullptr, unsigned *cols1 = nullptr,
                                                      unsigned *cols_b = nullptr) {
  __shared__ unsigned cm[16][64];
  if (blockIdx.z) { src = src1; cols = cols1; }
  const int tid = threadIdx.x, c0 = blockIdx.x * 64, r0 = blockIdx.y * 256;
  const int c4 = (tid & 15) * 4, rl = tid >> 4;
  unsigned cmax[4] = {0u, 0u, 0u, 0u};
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int r = r0 + rl + 16 * j;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) {
      const float *s = src + (size_t)r * ld + c0 + c4;
      if (c0 + c4 + 3 < C) v = *reinterpret_cast<const float4 *>(s);
      else {
        if (c0 + c4 < C) v.x = s[0];
        if (c0 + c4 + 1 < C) v.y = s[1];
        if (c0 + c4 + 2 < C) v.z = s[2];
      }
    }
    const unsigned b0 = __builtin_bit_cast(unsigned, v.x) & 0x7FFFFFFFu, b1 = __builtin_bit_cast(unsigned, v.y) & 0x7FFFFFFFu;
    const unsigned b2 = __builtin_bit_cast(unsigned, v.z) & 0x7FFFFFFFu, b3 = __builtin_bit_cast(unsigned, v.w) & 0x7FFFFFFFu;
    cmax[0] = max(cmax[0], b0); cmax[1] = max(cmax[1], b1); cmax[2] = max(cmax[2], b2); cmax[3] = max(cmax[3], b3);
    if (rows) {
      unsigned rm = max(max(b0, b1), max(b2, b3));
      rm = max(rm, (unsigned)__shfl_xor((int)rm, 1));
      rm = max(rm, (unsigned)__shfl_xor((int)rm, 2));
      rm = max(rm, (unsigned)__shfl_xor((int)rm, 4));
      rm = max(rm, (unsigned)__shfl_xor((int)rm, 8));
      if ((tid & 15) == 0 && r < R && rm) atomicMax(rows + r, rm);
    }
  }
  if (!cols) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) cm[rl][c4 + q] = cmax[q];
  __syncthreads();
  if (tid < 64 && c0 + tid < C) {
    unsigned m = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) m = max(m, cm[q][tid]);
    if (m) atomicMax(cols + c0 + tid, m);
    if (m && cols_b) atomicMax(cols_b + c0 + tid, m);
  }
}

// row maxima from per-unit partial maxima (the backward recurrence keeps the largest |dz| of every gate column per
// unit): dst[n] = bits of max_r src[r][n]
__global__ void pk_colmax_kernel(int rows, int N, const float *src, int ld, unsigned *dst) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float m = 0.f;
  for (int r = 0; r < rows; ++r) m = fmaxf(m, fabsf(src[(size_t)r * ld + n]));
  dst[n] = __builtin_bit_cast(unsigned, m);
}

// f16x3 maxima of dz out of the persistent backward kernel's own bookkeeping, one launch, no read of dz:
//   blocks [0, nb_rows): adz[r] = max over the nparts per-workgroup row maxima (bit patterns) of frame row r = b T + t
//                        (frames t >= max_len were never visited: 0; rows >= BT up to rows_pad: 0)
//   the others:          adzT[n] = bits of max_r |colpart[r][n]| (pk_colmax_kernel); and, where asked for, the bias
//                        gradients db0 | db1 [n] = sum_r sumpart[r][n] in row order (colsum_pair_kernel's sum: the
//                        kernel's per-unit partial sums sit next to its maxima — one launch less per layer)
__global__ __launch_bounds__(256) void pk_amax_persist_kernel(int nb_rows, int BT, int rows_pad, int T, int max_len, int nparts,
                                                              const unsigned *rowpart, unsigned *adz, int crow, int N,
                                                              const float *colpart, int ld, unsigned *adzT,
                                                              const float *sumpart, float *db0, float *db1) {
  if ((int)blockIdx.x < nb_rows) {
    // 64 frame rows per block, the partial rows in four groups (one per wave) that meet in LDS
    __shared__ unsigned red[4][64];
    const int r = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    unsigned m = 0;
    if (r < BT && r % T < max_len) {
      const int q0 = g * ((nparts + 3) / 4), q1 = min(nparts, q0 + (nparts + 3) / 4);
#pragma unroll 8
      for (int q = q0; q < q1; ++q) m = max(m, rowpart[(size_t)q * BT + r]);
    }
    red[g][threadIdx.x & 63] = m;
    __syncthreads();
    if (g == 0 && r < rows_pad) adz[r] = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
    return;
  }
  const int n = (blockIdx.x - nb_rows) * 256 + threadIdx.x;
  if (n >= N) return;
  float m = 0.f;
  for (int r = 0; r < crow; ++r) m = fmaxf(m, fabsf(colpart[(size_t)r * ld + n]));
  adzT[n] = __builtin_bit_cast(unsigned, m);
  if (sumpart) {
    float s = 0.f;
    for (int r = 0; r < crow; ++r) s += sumpart[(size_t)r * ld + n];
    if (n < N / 2) db0[n] = s;
    else db1[n - N / 2] = s;
  }
}

__global__ void pk_fill_u32_kernel(unsigned *dst, int n, unsigned v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = v;
}

static int pk_cu_count() {
  static thread_local int cached_dev = -1, cached = 256;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
      (void)hipGetLastError();
      v = 256;
    }
    cached = v;
    cached_dev = dev;
  }
  return cached;
}

// the tile loop needs 144 KiB of LDS per workgroup: devices that do not offer it take the other GEMM kernels
bool gemm_pk_device_ok() {
  static thread_local int cached_dev = -1;
  static thread_local bool cached = false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (dev != cached_dev) {
    int lds = 0;
    const hipDeviceAttribute_t names[3] = {hipDeviceAttributeMaxSharedMemoryPerBlock, hipDeviceAttributeSharedMemPerBlockOptin,
                                           hipDeviceAttributeMaxSharedMemoryPerMultiprocessor};
    for (hipDeviceAttribute_t n : names) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, n, dev) == hipSuccess && v > lds) lds = v;
    }
    (void)hipGetLastError();
    cached = lds >= PK_LDS;
    cached_dev = dev;
  }
  return cached;
}

// split-K policy: one workgroup per CU; with few output tiles cut the reduction so that the number of
// workgroups comes close to a multiple of the CU count (>= 24 stages per workgroup; 48 of f16x3's half-length stages)
static int pk_choose_split(int ntiles, int nkb, int kbs, int min_stages, int *kb_per_split) {
  const int ncu = pk_cu_count(), nst = nkb / kbs;
  int best = 1;
  double best_eff = 0.0;
  const int maxs = nst / min_stages < 1 ? 1 : (nst / min_stages > 32 ? 32 : nst / min_stages);
  for (int s = 1; s <= maxs; ++s) {
    const int per = (nst + s - 1) / s, ns = (nst + per - 1) / per;
    if (ns != s) continue;
    const long long wgs = (long long)ntiles * ns, rounds = (wgs + ncu - 1) / ncu;
    double eff = (double)ntiles * nst / ((double)rounds * ncu * per);   // useful stages / stage slots
    eff -= 0.01 * (ns - 1);                                             // the reduce pass is not free
    if (eff > best_eff + 1e-9) { best_eff = eff; best = ns; }
  }
  const int per = (nst + best - 1) / best;
  *kb_per_split = per * kbs;
  return (nst + per - 1) / per;
}

}  // namespace nabu

using namespace nabu;

extern "C" int nabu_pk_rows_pad(int rows) { return (rows + PK_T - 1) / PK_T * PK_T; }
extern "C" int nabu_pk_kblocks(int K, int planes) {
  const int kbs = planes == 1 ? 3 : 1, nkb = (K + 15) / 16;
  return (nkb + kbs - 1) / kbs * kbs;
}
extern "C" size_t nabu_pk_bytes(int rows, int K, int planes) {
  if (rows <= 0 || K <= 0 || planes < 1 || planes > 3) return 0;
  return (size_t)nabu_pk_kblocks(K, planes) * planes * nabu_pk_rows_pad(rows) * 32;
}

static int pk_pack_impl(int planes, int transposed, const float *src, long long ld, int R, int C, void *dst,
                        int dst_rows_pad, int row_off, int kb_off, int fill_rows, int fill_kb, int period, int shift,
                        const unsigned *amax, nabu_stream_t stream) {
  NABU_CHECK_ARG(planes >= 1 && planes <= 3, "pk_pack: planes must be 1, 2 (f16x3) or 3");
  NABU_CHECK_ARG((planes == 2) == (amax != nullptr), "pk_pack: the row maxima belong to planes = 2 (nabu_pk_pack_f16)");
  NABU_CHECK_ARG(src && dst && R >= 0 && C >= 0 && ld >= 0, "pk_pack: bad source");
  NABU_CHECK_ARG(dst_rows_pad > 0 && dst_rows_pad % PK_T == 0 && row_off >= 0 && kb_off >= 0 && fill_rows >= 0 &&
                     fill_kb >= 0 && row_off + fill_rows <= dst_rows_pad,
                 "pk_pack: bad destination geometry");
  if (ld % 4 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15))
    return fail(NABU_EUNSUP, "pk_pack: source rows and both buffers must be 16-byte aligned");
  if (fill_rows == 0 || fill_kb == 0) return 0;
  PackArgs a;
  a.src = src; a.ld = ld; a.R = R; a.C = C; a.fill_rows = fill_rows; a.fill_kb = fill_kb;
  a.dst = static_cast<char *>(dst);
  a.plane_stride = (unsigned)dst_rows_pad * 32u;
  a.kb_stride = (unsigned long long)planes * a.plane_stride;
  a.row_off = row_off; a.kb_off = kb_off; a.period = period; a.shift = shift; a.amax = amax;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!transposed) {
    NABU_CHECK_ARG(period == 0, "pk_pack: shift only in the transposed form");
    const dim3 grid((fill_kb + 3) / 4, (fill_rows + 63) / 64);
    if (planes == 3) hipLaunchKernelGGL(pk_pack_rows_kernel<3>, grid, dim3(256), 0, s, a);
    else if (planes == 2) hipLaunchKernelGGL(pk_pack_rows_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pk_pack_rows_kernel<1>, grid, dim3(256), 0, s, a);
  } else {
    const dim3 grid((fill_rows + 63) / 64, (fill_kb + 3) / 4);
    if (planes == 3) hipLaunchKernelGGL(pk_pack_cols_kernel<3>, grid, dim3(256), 0, s, a);
    else if (planes == 2) hipLaunchKernelGGL(pk_pack_cols_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pk_pack_cols_kernel<1>, grid, dim3(256), 0, s, a);
  }
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_pk_pack(int planes, int transposed, const float *src, long long ld, int R, int C, void *dst,
                            int dst_rows_pad, int row_off, int kb_off, int fill_rows, int fill_kb, int period,
                            int shift, nabu_stream_t stream) {
  NABU_CHECK_ARG(planes == 1 || planes == 3, "pk_pack: planes must be 1 or 3 (f16x3 operands: nabu_pk_pack_f16)");
  return pk_pack_impl(planes, transposed, src, ld, R, C, dst, dst_rows_pad, row_off, kb_off, fill_rows, fill_kb, period,
                      shift, nullptr, stream);
}

extern "C" int nabu_pk_pack_f16(int transposed, const float *src, long long ld, int R, int C, void *dst, int dst_rows_pad,
                                int row_off, int kb_off, int fill_rows, int fill_kb, int period, int shift,
                                const uint32_t *amax, nabu_stream_t stream) {
  NABU_CHECK_ARG(amax != nullptr && (reinterpret_cast<uintptr_t>(amax) & 15) == 0, "pk_pack_f16: amax missing or unaligned");
  return pk_pack_impl(2, transposed, src, ld, R, C, dst, dst_rows_pad, row_off, kb_off, fill_rows, fill_kb, period, shift,
                      amax, stream);
}

extern "C" int nabu_pk_amax(const float *src, long long ld, int R, int C, uint32_t *rows, uint32_t *cols,
                            nabu_stream_t stream) {
  NABU_CHECK_ARG(src && R >= 0 && C >= 0 && ld >= 0 && (rows || cols), "pk_amax: bad arguments");
  if (ld % 4 || (reinterpret_cast<uintptr_t>(src) & 15)) return fail(NABU_EUNSUP, "pk_amax: source rows must be 16-byte aligned");
  if (R == 0 || C == 0) return 0;
  hipLaunchKernelGGL(pk_amax_kernel, dim3((C + 63) / 64, (R + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     ld, R, C, rows, cols);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_pk_amax_fill(uint32_t *dst, int n, float value, nabu_stream_t stream) {
  NABU_CHECK_ARG(dst && n >= 0 && value >= 0.f, "pk_amax_fill: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pk_fill_u32_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), dst, n,
                     __builtin_bit_cast(unsigned, value));
  NABU_LAUNCH_CHECK();
  return 0;
}

// internal (lstm.hip): natural pack into (dst_n: rows_pad_n, kb_off_n, fill_kb_n) AND transposed pack into
// (dst_t: rows_pad_t, row_off_t, fill_rows_t) of the same [R x C] source in one pass.  The natural pack fills rows
// [0, fill_rows_n), the transposed pack k-blocks [0, fill_kb_t).
namespace nabu {
int pk_pack_multi(int planes, int transposed, const PkPackReq *req, int n, hipStream_t s) {
  if (n < 1 || n > PK_NREQ) return fail(NABU_EINVAL, "pk_pack_multi: 1 .. %d requests", PK_NREQ);
  if (planes < 1 || planes > 3) return fail(NABU_EINVAL, "pk_pack_multi: planes must be 1, 2 or 3");
  PackArgsN a = {};
  int gx = 0, gy = 0;
  for (int i = 0; i < n; ++i) {
    const PkPackReq &q = req[i];
    if ((planes == 2) != (q.amax != nullptr)) return fail(NABU_EINVAL, "pk_pack_multi: the row maxima belong to planes = 2");
    if (!q.src || !q.dst || q.R < 0 || q.C < 0 || q.ld < 0 || q.rows_pad <= 0 || q.rows_pad % PK_T || q.row_off < 0 ||
        q.kb_off < 0 || q.fill_rows < 0 || q.fill_kb < 0 || q.row_off + q.fill_rows > q.rows_pad)
      return fail(NABU_EINVAL, "pk_pack_multi: bad request %d", i);
    if (q.ld % 4 || (reinterpret_cast<uintptr_t>(q.src) & 15) || (reinterpret_cast<uintptr_t>(q.dst) & 15))
      return fail(NABU_EUNSUP, "pk_pack_multi: source rows and both buffers must be 16-byte aligned");
    if (!transposed && q.period) return fail(NABU_EINVAL, "pk_pack_multi: shift only in the transposed form");
    PackArgs &p = a.r[i];
    p.src = q.src; p.ld = q.ld; p.R = q.R; p.C = q.C; p.fill_rows = q.fill_rows; p.fill_kb = q.fill_kb;
    p.dst = static_cast<char *>(q.dst);
    p.plane_stride = (unsigned)q.rows_pad * 32u;
    p.kb_stride = (unsigned long long)planes * p.plane_stride;
    p.row_off = q.row_off; p.kb_off = q.kb_off; p.period = q.period; p.shift = q.shift; p.amax = q.amax;
    const int bx = transposed ? (q.fill_rows + 63) / 64 : (q.fill_kb + 3) / 4;
    const int by = transposed ? (q.fill_kb + 3) / 4 : (q.fill_rows + 63) / 64;
    gx = bx > gx ? bx : gx; gy = by > gy ? by : gy;
  }
  if (!gx || !gy) return 0;
  const dim3 grid(gx, gy, n);
#define PK_MULTI(NP_)                                                                              \
  if (transposed) hipLaunchKernelGGL(pk_pack_cols_multi_kernel<NP_>, grid, dim3(256), 0, s, a);     \
  else hipLaunchKernelGGL(pk_pack_rows_multi_kernel<NP_>, grid, dim3(256), 0, s, a);
  if (planes == 3) { PK_MULTI(3) } else if (planes == 2) { PK_MULTI(2) } else { PK_MULTI(1) }
#undef PK_MULTI
  NABU_LAUNCH_CHECK();
  return 0;
}

int pk_amax_pair(const float *src0, const float *src1, long long ld, int R, int C, uint32_t *rows, uint32_t *cols0,
                 uint32_t *cols1, uint32_t *cols_b, hipStream_t s) {
  if (!src0 || R < 0 || C < 0 || ld < 0 || !(rows || cols0)) return fail(NABU_EINVAL, "pk_amax_pair: bad arguments");
  if (ld % 4 || (reinterpret_cast<uintptr_t>(src0) & 15) || (reinterpret_cast<uintptr_t>(src1) & 15))
    return fail(NABU_EUNSUP, "pk_amax_pair: source rows must be 16-byte aligned");
  if (R == 0 || C == 0) return 0;
  hipLaunchKernelGGL(pk_amax_kernel, dim3((C + 63) / 64, (R + 255) / 256, src1 ? 2 : 1), dim3(256), 0, s, src0, ld, R, C, rows,
                     cols0, src1, cols1, cols_b);
  NABU_LAUNCH_CHECK();
  return 0;
}

int pk_amax_from_persist(int BT, int rows_pad, int T, int max_len, int nparts, const uint32_t *rowpart, uint32_t *adz,
                         int crow, int N, const float *colpart, int ld, uint32_t *adzT, hipStream_t s,
                         const float *sumpart, float *db0, float *db1) {
  const int nb_rows = adz ? (rows_pad + 63) / 64 : 0, nb_cols = adzT ? (N + 255) / 256 : 0;
  if (sumpart && (!adzT || !db0 || !db1 || N % 2)) return fail(NABU_EINVAL, "pk_amax_from_persist: the column sums ride on the column maxima");
  if (nb_rows + nb_cols == 0) return 0;
  hipLaunchKernelGGL(pk_amax_persist_kernel, dim3(nb_rows + nb_cols), dim3(256), 0, s, nb_rows, BT, rows_pad, T, max_len, nparts,
                     rowpart, adz, crow, N, colpart, ld, adzT, sumpart, db0, db1);
  NABU_LAUNCH_CHECK();
  return 0;
}

int pk_amax_from_partials(int rows, int N, const float *part, int ld, uint32_t *amax, hipStream_t s) {
  if (rows <= 0 || N <= 0) return 0;
  hipLaunchKernelGGL(pk_colmax_kernel, dim3((N + 255) / 256), dim3(256), 0, s, rows, N, part, ld, amax);
  NABU_LAUNCH_CHECK();
  return 0;
}
int pk_pack_both(int planes, const float *src, long long ld, int R, int C, void *dst_n, int rows_pad_n, int kb_off_n,
                 int fill_rows_n, int fill_kb_n, void *dst_t, int rows_pad_t, int row_off_t, int fill_rows_t,
                 int fill_kb_t, hipStream_t s, const unsigned *amax_n, const unsigned *amax_t) {
  if (ld % 4 || (reinterpret_cast<uintptr_t>(src) & 15)) return fail(NABU_EUNSUP, "pk_pack_both: unaligned source");
  PackArgs a, b;
  a.src = b.src = src; a.ld = b.ld = ld; a.R = b.R = R; a.C = b.C = C;
  a.fill_rows = fill_rows_n; a.fill_kb = fill_kb_n; a.dst = static_cast<char *>(dst_n);
  a.plane_stride = (unsigned)rows_pad_n * 32u; a.kb_stride = (unsigned long long)planes * a.plane_stride;
  a.row_off = 0; a.kb_off = kb_off_n; a.period = 0; a.shift = 0;
  a.amax = amax_n; b.amax = amax_t;
  if (planes == 2 && (!amax_n || !amax_t)) return fail(NABU_EINVAL, "pk_pack_both: f16x3 operands need the row maxima");
  b.fill_rows = fill_rows_t; b.fill_kb = fill_kb_t; b.dst = static_cast<char *>(dst_t);
  b.plane_stride = (unsigned)rows_pad_t * 32u; b.kb_stride = (unsigned long long)planes * b.plane_stride;
  b.row_off = row_off_t; b.kb_off = 0; b.period = 0; b.shift = 0;
  const int gx = ((fill_kb_n + 3) / 4 > (fill_rows_t + 63) / 64) ? (fill_kb_n + 3) / 4 : (fill_rows_t + 63) / 64;
  const int gy = ((fill_rows_n + 63) / 64 > (fill_kb_t + 3) / 4) ? (fill_rows_n + 63) / 64 : (fill_kb_t + 3) / 4;
  if (planes == 3) hipLaunchKernelGGL(pk_pack_both_kernel<3>, dim3(gx, gy), dim3(256), 0, s, a, b);
  else if (planes == 2) hipLaunchKernelGGL(pk_pack_both_kernel<2>, dim3(gx, gy), dim3(256), 0, s, a, b);
  else hipLaunchKernelGGL(pk_pack_both_kernel<1>, dim3(gx, gy), dim3(256), 0, s, a, b);
  NABU_LAUNCH_CHECK();
  return 0;
}
}  // namespace nabu

static int pk_fill(PkArgs &p, const nabu_pk_gemm_desc *d, int *planes) {
  if (!d || d->size != sizeof(nabu_pk_gemm_desc)) return fail(NABU_EINVAL, "gemm_pk: bad descriptor size");
  *planes = d->planes;
  if (d->planes < 1 || d->planes > 3) return fail(NABU_EINVAL, "gemm_pk: planes must be 1, 2 or 3");
  if (d->direct < 0 || d->direct > 2) return fail(NABU_EINVAL, "gemm_pk: direct must be 0, 1 or 2");
  if (d->planes == 2) {
    if (d->a_planes != 2 || d->b_planes != 2) return fail(NABU_EINVAL, "gemm_pk: an f16x3 product takes f16x3 operands (2 planes)");
    for (int b = 0; b < d->nbatch; ++b)
      if (!d->a_amax[b] || !d->b_amax[b] || (reinterpret_cast<uintptr_t>(d->b_amax[b]) & 15))
        return fail(NABU_EINVAL, "gemm_pk: f16x3 operands need their row maxima (b_amax 16-byte aligned)");
  } else if (d->a_planes == 2 || d->b_planes == 2) {
    return fail(NABU_EINVAL, "gemm_pk: f16x3 operands only feed planes = 2 products");
  }
  if (d->M <= 0 || d->N <= 0 || d->nkb <= 0 || d->nbatch < 1 || d->nbatch > 2)
    return fail(NABU_EINVAL, "gemm_pk: bad dimensions");
  const int kbs = d->planes == 1 ? 3 : 1;
  if (d->nkb % kbs) return fail(NABU_EINVAL, "gemm_pk: nkb must be a multiple of %d", kbs);
  if (d->a_rows_pad % PK_T || d->b_rows_pad % PK_T || d->a_rows_pad < nabu_pk_rows_pad(d->M) ||
      d->b_rows_pad < nabu_pk_rows_pad(d->N))
    return fail(NABU_EINVAL, "gemm_pk: packed row counts must be multiples of 256 covering M and N");
  if (d->a_planes < d->planes || d->b_planes < d->planes)
    return fail(NABU_EINVAL, "gemm_pk: operand holds fewer planes than the product needs");
  if (d->n_split && (d->n_split % PK_T || !d->C2[0])) return fail(NABU_EINVAL, "gemm_pk: n_split must be a multiple of 256 with C2 set");
  if (d->N % 4 || d->ldc % 4) return fail(NABU_EUNSUP, "gemm_pk: N and ldc must be multiples of 4");
  auto misaligned = [](const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
  if (misaligned(d->bias) || misaligned(d->bias2)) return fail(NABU_EUNSUP, "gemm_pk: bias must be 16-byte aligned");
  for (int b = 0; b < d->nbatch; ++b)
    if (misaligned(d->C[b]) || misaligned(d->C2[b])) return fail(NABU_EUNSUP, "gemm_pk: C must be 16-byte aligned");
  for (int b = 0; b < d->nbatch; ++b)
    if (!d->A[b] || !d->B[b] || !d->C[b]) return fail(NABU_EINVAL, "gemm_pk: null pointer");
  p.A.plane_stride = (unsigned)d->a_rows_pad * 32u;
  p.A.kb_stride = (unsigned long long)d->a_planes * p.A.plane_stride;
  p.B.plane_stride = (unsigned)d->b_rows_pad * 32u;
  p.B.kb_stride = (unsigned long long)d->b_planes * p.B.plane_stride;
  if (2 * (unsigned long long)(d->planes != 1 ? p.A.plane_stride : p.A.kb_stride) >= (1ull << 32) ||
      2 * (unsigned long long)(d->planes != 1 ? p.B.plane_stride : p.B.kb_stride) >= (1ull << 32))
    return fail(NABU_EUNSUP, "gemm_pk: operand too tall for 32-bit piece offsets");
  for (int b = 0; b < 2; ++b) {
    const int s = b < d->nbatch ? b : 0;
    p.A.base[b] = static_cast<const char *>(d->A[s]);
    p.B.base[b] = static_cast<const char *>(d->B[s]);
    p.C[b] = d->C[s];
    p.C2[b] = d->C2[s];
    p.a_amax[b] = d->planes == 2 ? d->a_amax[s] : nullptr;
    p.b_amax[b] = d->planes == 2 ? d->b_amax[s] : nullptr;
  }
  p.M = d->M; p.N = d->N;
  p.tiles_m = (d->M + PK_T - 1) / PK_T; p.tiles_n = (d->N + PK_T - 1) / PK_T; p.nbatch = d->nbatch;
  p.nkb = d->nkb;
  p.ldc = d->ldc; p.n_split = d->n_split; p.bias = d->bias; p.bias2 = d->bias2;
  p.alpha = d->alpha; p.beta = d->beta; p.partial = nullptr;
  static int force = -2, throttle = 0;
  if (force == -2) {
    const char *e = getenv("NABU_PK_SPLIT");
    force = e ? atoi(e) : -1;
    e = getenv("NABU_PK_THROTTLE");
    throttle = e ? atoi(e) : 0;
  }
  p.throttle = throttle;
  p.nsplit = pk_choose_split(p.tiles_m * p.tiles_n * p.nbatch, p.nkb, kbs, d->planes == 2 ? 48 : 24, &p.kb_per_split);
  if (force > 0) {
    const int nst = p.nkb / kbs, per = (nst + force - 1) / force;
    p.kb_per_split = per * kbs;
    p.nsplit = (nst + per - 1) / per;
  }
  return 0;
}

extern "C" size_t nabu_gemm_pk_ws_bytes(const nabu_pk_gemm_desc *d) {
  PkArgs p;
  int planes;
  if (pk_fill(p, d, &planes)) return 0;
  return p.nsplit > 1 ? (size_t)p.nsplit * p.nbatch * p.M * p.N * sizeof(float) : 0;
}

extern "C" int nabu_gemm_pk(const nabu_pk_gemm_desc *d, void *ws, size_t ws_bytes, nabu_stream_t stream) {
  PkArgs p;
  int planes;
  if (int e = pk_fill(p, d, &planes)) return e;
  if (!gemm_pk_device_ok()) return fail(NABU_EUNSUP, "gemm_pk: the device offers less than %d bytes of LDS per workgroup", PK_LDS);
  if (p.nsplit > 1) {
    const size_t need = (size_t)p.nsplit * p.nbatch * p.M * p.N * sizeof(float);
    if (!ws || ws_bytes < need) return fail(NABU_EWS, "gemm_pk: workspace %zu < %zu", ws_bytes, need);
    p.partial = static_cast<float *>(ws);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  static int var_env = -2;
  if (var_env == -2) { const char *e = getenv("NABU_PK_VAR"); var_env = e ? atoi(e) : -1; }
  // promoted accumulation unless the caller asks for the direct chain (planes = 2, d->direct) or the environment
  // overrides (A/B measurements).  direct = 2 decides by rounding count: the direct chain rounds 3 times per 16 k,
  // v_mfma_f32_32x32x2_f32 8 times — so it is taken when a workgroup's reduction here is at most twice as long as the
  // exact-fp32 kernel's would be for the same product (its split-K policy: nabu_gemm_ws_bytes)
  bool direct = planes == 2 && d->direct == 1;
  if (planes == 2 && d->direct == 2) {
    const long long K = (long long)p.nkb * 16;
    const size_t w32 = nabu_gemm_ws_bytes(p.M, p.N, (int)K);
    const long long ns32 = w32 ? (long long)(w32 / ((size_t)p.M * p.N * sizeof(float))) : 1;
    direct = (long long)p.kb_per_split * 16 <= 2 * (K / (ns32 > 0 ? ns32 : 1));
  }
  const int var = var_env >= 0 ? var_env : (direct ? 0 : 1);
  const int grid = p.tiles_m * p.tiles_n * p.nbatch * p.nsplit;
#define PK_LAUNCH(NP_, VAR_)                                                                                         \
  {                                                                                                                   \
    const int lds = NP_ == 2 ? PK_LDS2 : PK_LDS;                                                                       \
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_pk_kernel<NP_, VAR_>),                           \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                   \
    hipLaunchKernelGGL((gemm_pk_kernel<NP_, VAR_>), dim3(grid), dim3(512), lds, s, p);                                 \
  }
  if (planes == 1) PK_LAUNCH(1, 0)
  else if (planes == 2 && var == 0) PK_LAUNCH(2, 0)
  else if (planes == 2) PK_LAUNCH(2, 1)
  else if (var == 0) PK_LAUNCH(3, 0)
  else PK_LAUNCH(3, 1)
#undef PK_LAUNCH
  NABU_LAUNCH_CHECK();
  if (p.nsplit > 1) {
    const size_t n4 = (size_t)p.M * p.N * p.nbatch / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_pk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
    NABU_LAUNCH_CHECK();
  }
  return 0;
}
