// lstm_persist_mxh.h — device helpers of the fp16-plane persistent recurrent kernels (lstm_persist_mxh.hip: 16 hidden
// units per workgroup; lstm_persist_mxf.hip: 32 per workgroup, two units per XCD): row scales, plane splits, the tagged
// backward ring
#pragma once
#include "lstm_persist_mx.h"

namespace nabu {

// timing experiments (never defined in the library build): exchange volume cut to a quarter / no backward product
#ifdef MXH_EXP_QVOL
#define MXH_QVOL(x) ((x) != 0)
#else
#define MXH_QVOL(x) false
#endif

typedef _Float16 mxh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mxh16x2 __attribute__((ext_vector_type(2)));
#define MXH_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mxh16x8, a), __builtin_bit_cast(mxh16x8, b), c, 0, 0, 0)

// A plane of W_h held in the ACCUMULATION half of the unified register file and fed to the matrix instruction from there:
// hipcc keeps matrix operands in ordinary registers, and once those run out (256 per lane) it parks values in accumulation
// registers and copies them back in front of every use (v_accvgpr_read, one VALU slot each — dozens per step in these
// kernels).  An instruction issued this way is invisible to hipcc's hazard padding, so it is ONLY used where the
// compiler's own matrix instruction on the same accumulators follows before any vector-ALU read of them (the l plane in
// front of the h plane of the same tile).
__device__ __forceinline__ void mxf_pin_acc(u32x4 &w) { asm volatile("" : "+a"(w)); }
__device__ __forceinline__ mxf32x4 mxf_mfma_acc(const u32x4 wa, const u32x4 b, mxf32x4 acc) {
  asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(wa), "v"(b));
  return acc;
}

constexpr int DPP_ROW_MIRROR = 0x140;
// backward exchange ring: 2 slots, NO hand-back.  The backward step is bound by the volume of its exchange (16 KB
// written, 16 KB read and — with sentinels — 16 KB handed back per workgroup and step, at ~2.5 TB/s per XCD: cutting the
// volume to a quarter takes 0.45 us off a 2.2 us step, removing the whole product nothing).  So
// the flag is ONE BIT of the data: the least significant bit of every published fp32 partial sum carries the
// generation of its slot (iteration it = max_len - 1 - s: slot it & 1, generation it >> 1, tag = generation & 1; the ring
// starts as 0xFF bytes, tag 1, and generation 0 has tag 0), the reader repeats its loads until every word carries the
// tag it waits for.  A published partial sum thus has 23 significant bits and is off by at most one unit of its last
// place — the size of the rounding it went through anyway — low in even generations, high in odd ones: no bias over
// time.  Ring of 2 is safe without hand-back: a slot written in iteration it is overwritten in it + 2 by a writer that
// has polled the reader's own publish of it + 1, issued behind the barrier that follows the reader's poll of the slot.
// Non-finite values: the tag replaces the last mantissa bit whatever the exponent, so a +-Inf partial sum (0x7F800000)
// published with tag 1 reads as a NaN pattern, and exact zeros (padded rows) travel as the smallest denormal — an
// overflow in the backward recurrence therefore shows up as NaN where the exact-fp32 and step-wise kernels show Inf;
// finite training is unaffected (the denormal is below every other term of the sum it joins).
constexpr int MXHRINGB = 2;
constexpr float MXH_HSCALE = 16384.f, MXH_HINV = 1.f / 16384.f;     // forward h: |h| < 1 + 2^-22

// row scales from the bit pattern of the row's largest magnitude (the convention of gemm_pk.hip, pk_scale_of): exponent
// field clamped so that scale and inverse are normal numbers; an all-zero row takes amax = 1; inf / NaN rows keep a
// finite scale and propagate through the planes
__device__ __forceinline__ unsigned mxh_amax_exp(unsigned bits) {
  unsigned e = (bits >> 23) & 0xFFu;
  if ((bits & 0x7FFFFFFFu) == 0) e = 127;
  return e < 15 ? 15 : (e > 253 ? 253 : e);
}
__device__ __forceinline__ float mxh_scale_of(float amax) {
  return __builtin_bit_cast(float, (268u - mxh_amax_exp(__builtin_bit_cast(unsigned, amax))) << 23);
}
__device__ __forceinline__ float mxh_inv_scale_of(float amax) {
  return __builtin_bit_cast(float, (mxh_amax_exp(__builtin_bit_cast(unsigned, amax)) - 14u) << 23);
}
__device__ __forceinline__ unsigned mxh_cvt2(float a, float b) {   // (fp16(a), fp16(b)), round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, mxh16x2));
}
// two scaled values -> one word per plane (a | b << 16)
__device__ __forceinline__ void mxh_split2x2(float a, float b, unsigned &h, unsigned &l) {
  h = mxh_cvt2(a, b);
  const mxh16x2 hv = __builtin_bit_cast(mxh16x2, h);
  l = mxh_cvt2(a - (float)hv.x, b - (float)hv.y);
}
// 8 consecutive-k scaled values -> the two plane operands
__device__ __forceinline__ void mxh_split8(const float *x, u32x4 &h, u32x4 &l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) mxh_split2x2(x[2 * i], x[2 * i + 1], hh[i], ll[i]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}
__device__ __forceinline__ float mxh_xor16(float v) {      // lane ^ 16 inside every group of 32 (bit-mask swizzle)
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

// launch of one fp16-plane kernel (256 threads): the grid is validated against the runtime's occupancy answer first
template <typename K>
static int mxh_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[16] = {};
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  int blocks = -1;
  for (const Seen &c : seen)
    if (c.fn == fn && c.dev == dev) blocks = c.blocks;
  if (blocks < 0) {
    NABU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 256, lds));
    for (Seen &c : seen)
      if (!c.fn) { c = Seen{fn, dev, blocks}; break; }
  }
  if (blocks < 1 || grid > NCU)
    return fail(NABU_EUNSUP, "persistent LSTM (mxh): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;          // validation pass (lstm_persist.hip, run)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

int lstm_mxh_bwd_launch(int H, const PersistArgs &a, hipStream_t stream, bool dry);   // lstm_persist_mxh_bwd.hip

}  // namespace nabu
