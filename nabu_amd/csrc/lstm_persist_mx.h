// lstm_persist_mx.h — device helpers shared by the bf16-plane persistent recurrent kernels
// (lstm_persist_mxh.hip, lstm_persist_mxf.hip; the bf16 helpers served the parked kernels of tools/experiments/variants/)
#pragma once
#include "lstm_persist_dev.h"

namespace nabu {

typedef __bf16 mxbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mxbf16x2 __attribute__((ext_vector_type(2)));
typedef float mxf32x4 __attribute__((ext_vector_type(4)));

constexpr int MXR = 8;                 // batch rows per unit
constexpr int MXNU = 8;                // units per launch = XCDs
constexpr unsigned MXOOB = 0x80000000u;   // masked lanes: + any in-slot offset stays out of range
// backward exchange ring: 3 slots.  A piece is reset by its reader in step s (slot (s + 1) % 3) and written again in
// step s - 2; in between the writer polls the reader's publish of step s - 1, which every wave of the reader issues
// behind that step's barrier, i.e. behind every wave's poll of step s - 1, whose loads were issued behind the reset
// stores (vector-memory operations complete in issue order): no drain, no second barrier (lstm_persist.hip needs
// both for its ring of 2).  One unit per XCD: 3 x 512 KiB at H = 512 stay in the 4 MiB L2.
constexpr int MXRINGB = 3;

__device__ __forceinline__ unsigned mx_cvt2(float a, float b) {   // (bf16(a), bf16(b)) round to nearest even
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, mxbf16x2));
}
// x = h + m + l exactly (bf16 bit patterns in the low halves)
__device__ __forceinline__ void mx_split3(float x, unsigned &h, unsigned &m, unsigned &l) {
  h = mx_cvt2(x, 0.f) & 0xFFFFu;
  const float r1 = x - __builtin_bit_cast(float, h << 16);
  m = mx_cvt2(r1, 0.f) & 0xFFFFu;
  const float r2 = r1 - __builtin_bit_cast(float, m << 16);
  l = mx_cvt2(r2, 0.f) & 0xFFFFu;
}
// two values -> one word per plane: (a | b << 16)
__device__ __forceinline__ void mx_split3x2(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  h = mx_cvt2(a, b);
  const float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xFFFF0000u);
  m = mx_cvt2(ra, rb);
  l = mx_cvt2(ra - __builtin_bit_cast(float, m << 16), rb - __builtin_bit_cast(float, m & 0xFFFF0000u));
}
// 8 consecutive-k values -> the three plane operands
__device__ __forceinline__ void mx_split8(const float *x, u32x4 &h, u32x4 &m, u32x4 &l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) mx_split3x2(x[2 * i], x[2 * i + 1], hh[i], mm[i], ll[i]);
  h = (u32x4){hh[0], hh[1], hh[2], hh[3]};
  m = (u32x4){mm[0], mm[1], mm[2], mm[3]};
  l = (u32x4){ll[0], ll[1], ll[2], ll[3]};
}
#define MX_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mxbf16x8, a), __builtin_bit_cast(mxbf16x8, b), c, 0, 0, 0)

template <int CTRL>
__device__ __forceinline__ float mx_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ unsigned mx_dppu(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141;
constexpr int DPP_SHL2 = 0x102, DPP_SHL4 = 0x104, DPP_SHL6 = 0x106;

__device__ __forceinline__ unsigned mx_max4(unsigned m, const u32x4 v) {
  return max(max(m, max(v.x, v.y)), max(v.z, v.w));
}

// logical identity: unit = XCD (b % 8), slot = b / 8; units >= NU leave at once
__device__ __forceinline__ void mx_identity(int *unit, int *slot) {
  *unit = blockIdx.x % MXNU;
  *slot = blockIdx.x / MXNU;
}

}  // namespace nabu
