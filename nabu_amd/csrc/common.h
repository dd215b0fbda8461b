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
int sample_ids_rows(int B, int C, const float *logits, float prob, unsigned long long seed, unsigned long long offset,
                    const int32_t *teacher_ids, int32_t *out_ids, int b0, hipStream_t stream);

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// tanh through one exp: 2*sigmoid(2x)-1 (abs error ~1e-7)
__device__ __forceinline__ float tanhf_(float x) { return 2.0f / (1.0f + __expf(-2.0f * x)) - 1.0f; }

}  // namespace nabu
