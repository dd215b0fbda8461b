// common.h — shared helpers of libnabu_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nabu_hip.h"

namespace nabu {

// thread-local last-error text returned by nabu_last_error()
char *err_buf();
int fail(int code, const char *fmt, ...);

#define NABU_CHECK_ARG(cond, ...)                          \
  do {                                                     \
    if (!(cond)) return ::nabu::fail(NABU_EINVAL, __VA_ARGS__); \
  } while (0)

#define NABU_HIP(call)                                                        \
  do {                                                                        \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
      return ::nabu::fail((int)e_, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define NABU_LAUNCH_CHECK()                                                   \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess)                                                     \
      return ::nabu::fail((int)e_, "kernel launch failed (%s:%d): %s", __FILE__, \
                          __LINE__, hipGetErrorString(e_));                   \
  } while (0)

// speller.hip: x[r*ld] = 1 for r < rows (initial alignments of the windowed attention)
int first_col_one(int rows, int ld, float *x, hipStream_t s);

// elementwise.hip: dropout / scheduled sampling on a sub-batch of rows with the whole batch's random stream
int dropout_rows(size_t n, const float *x, float *y, float keep_prob, unsigned long long seed, unsigned long long offset,
                 size_t first_elem, hipStream_t stream);
// one decoder step's scheduled sampling in one launch ([h | ctx] . Wout + bias evaluated only for the rows that are
// sampled): elementwise.hip, sample_step_kernel
bool sample_step_ok(int C);
int sample_step(int B, int C, int U, int E, const float *h, int ldh, const float *ctx, int ldc, const float *Wout,
                const float *bias, float prob, unsigned long long seed, unsigned long long offset, const int32_t *teacher_ids,
                int32_t *out_ids, int b0, hipStream_t stream);
int sample_ids_rows(int B, int C, const float *logits, float prob, unsigned long long seed, unsigned long long offset,
                    const int32_t *teacher_ids, int32_t *out_ids, int b0, hipStream_t stream);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// elementwise.hip: up to 8 regions set to a 32-bit pattern each by ONE launch (ptr 16-byte aligned, a whole number of
// 4-byte words).  Stands where a layer call needs several hipMemsetAsync in a row — exchange rings (0xFF), row-maximum
// arrays (0 or an a-priori bound): every memset is its own dispatch and costs ~6 us of queue gap behind a kernel.
// elementwise.hip: out0[c] = sum_r part[r * ld + c], out1[c] = sum_r part[r * ld + N + c] for a FEW rows (the per-unit
// bias-gradient partials of the persistent backward kernel, both cells): one launch, fixed order
int colsum_pair(int rows, int N, const float *part, int ld, float *out0, float *out1, hipStream_t stream);
struct FillSeg { void *ptr; size_t words; unsigned value; };
int multi_fill(const FillSeg *segs, int n, hipStream_t stream);

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh through one exp: 2*sigmoid(2x)-1 (abs error ~1e-7)
__device__ __forceinline__ float tanhf_(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

// counter-based RNG (Philox4x32-10) for dropout masks, scheduled sampling and Gaussian input noise:
// element i uses counter (i/4, offset) and key seed, so the backward pass
// regenerates the forward mask instead of storing it.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // (one 32 x 32 -> 64 multiply per product — v_mad_u64_u32 — instead of a high and a low one: the multiplies run at
    //  a quarter of the vector rate and are what a dropout pass is bound by)
    const unsigned long long p0 = (unsigned long long)M0 * c.x, p1 = (unsigned long long)M1 * c.z;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// dropout scale factors (0 or 1/keep) of the 4-element group `group` of the array the stream is defined on
// (dropout_kernel of elementwise.hip: element e = 4 group + j keeps its value when u01(r[j]) < keep)
__device__ __forceinline__ float4 dropout_scale4(unsigned long long group, float keep, unsigned long long seed,
                                                 unsigned long long offset) {
  const uint4 r = philox4x32_10(make_uint4((unsigned)group, (unsigned)(group >> 32), (unsigned)offset, (unsigned)(offset >> 32)),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  const float inv = 1.0f / keep;
  return make_float4(u01(r.x) < keep ? inv : 0.f, u01(r.y) < keep ? inv : 0.f, u01(r.z) < keep ? inv : 0.f,
                     u01(r.w) < keep ? inv : 0.f);
}

}  // namespace nabu
