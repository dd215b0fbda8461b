// elementwise.hip — HBM-bound helpers: fused clip+Adam, bias-gradient column
// sums, time padding for odd-length pyramid stacks.
#include "common.h"

namespace nabu {

// ---------------------------------------------------------------------------
// fused clip + TF-style Adam: 4 streams read (param, grad, m, v), 3 written.
// 16-byte accesses, grid-stride, 7*4 = 28 algorithmic bytes per parameter.
__global__ __launch_bounds__(256) void adam_clip_kernel(size_t n, float *__restrict__ p,
                                                        const float *__restrict__ g,
                                                        float *__restrict__ m,
                                                        float *__restrict__ v, float lr_t,
                                                        float b1, float b2, float eps,
                                                        float clip, float gscale) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = tid; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4 *>(p)[i];
    float4 gg = reinterpret_cast<const float4 *>(g)[i];
    float4 mm = reinterpret_cast<float4 *>(m)[i];
    float4 vv = reinterpret_cast<float4 *>(v)[i];
#define NABU_ADAM1(c)                                           \
  {                                                             \
    float x = fminf(fmaxf(gg.c * gscale, -clip), clip);         \
    mm.c = b1 * mm.c + (1.f - b1) * x;                          \
    vv.c = b2 * vv.c + (1.f - b2) * x * x;                      \
    pp.c -= lr_t * mm.c / (sqrtf(vv.c) + eps);                  \
  }
    NABU_ADAM1(x) NABU_ADAM1(y) NABU_ADAM1(z) NABU_ADAM1(w)
    reinterpret_cast<float4 *>(p)[i] = pp;
    reinterpret_cast<float4 *>(m)[i] = mm;
    reinterpret_cast<float4 *>(v)[i] = vv;
  }
  for (size_t i = n4 * 4 + tid; i < n; i += stride) {
    float x = fminf(fmaxf(g[i] * gscale, -clip), clip);
    float mm = b1 * m[i] + (1.f - b1) * x;
    float vv = b2 * v[i] + (1.f - b2) * x * x;
    m[i] = mm;
    v[i] = vv;
    p[i] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

__global__ __launch_bounds__(256) void clip_kernel(size_t n, float *__restrict__ g, float clip) {
  const size_t n4 = n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = tid; i < n4; i += stride) {
    float4 x = reinterpret_cast<float4 *>(g)[i];
    x.x = fminf(fmaxf(x.x, -clip), clip);
    x.y = fminf(fmaxf(x.y, -clip), clip);
    x.z = fminf(fmaxf(x.z, -clip), clip);
    x.w = fminf(fmaxf(x.w, -clip), clip);
    reinterpret_cast<float4 *>(g)[i] = x;
  }
  for (size_t i = n4 * 4 + tid; i < n; i += stride) g[i] = fminf(fmaxf(g[i], -clip), clip);
}

// ---------------------------------------------------------------------------
// column sums: stage 1 -> partial[rs][N], stage 2 -> out[N].  Fixed summation
// order => bitwise reproducible bias gradients.
constexpr int CS_ROWS = 512;  // rows per stage-1 block

__global__ __launch_bounds__(256) void colsum_stage1(int M, int N, const float *__restrict__ A,
                                                     int lda, float *__restrict__ partial) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * CS_ROWS;
  const int r1 = min(M, r0 + CS_ROWS);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += A[(size_t)r * lda + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N)
    partial[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                          (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void colsum_stage2(int nparts, int N,
                                                     const float *__restrict__ partial, float beta,
                                                     float *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += partial[(size_t)p * N + c];
  out[c] = (beta != 0.f ? beta * out[c] : 0.f) + s;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_time_kernel(int B, int T, int Tp, int F,
                                                       const float *__restrict__ x,
                                                       float *__restrict__ y, int unpad) {
  // one thread per float4 (F % 4 == 0 checked by the host) of the larger tensor
  const size_t F4 = F / 4;
  const size_t total = (size_t)B * (unpad ? T : Tp) * F4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t f = i % F4;
    const size_t bt = i / F4;
    const int TT = unpad ? T : Tp;
    const size_t b = bt / TT, t = bt % TT;
    if (unpad) {
      reinterpret_cast<float4 *>(y)[(b * T + t) * F4 + f] =
          reinterpret_cast<const float4 *>(x)[(b * Tp + t) * F4 + f];
    } else {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((int)t < T) v = reinterpret_cast<const float4 *>(x)[(b * T + t) * F4 + f];
      reinterpret_cast<float4 *>(y)[(b * Tp + t) * F4 + f] = v;
    }
  }
}

static int grid_for(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace nabu

using namespace nabu;

extern "C" int nabu_adam_clip_step(size_t n, float *param, const float *grad, float *m, float *v,
                                   float lr_t, float b1, float b2, float eps, float clip,
                                   float grad_scale, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(param && grad && m && v, "adam: null pointer");
  NABU_CHECK_ARG(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)m | (uintptr_t)v) % 16 == 0,
                 "adam: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_clip_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, param, grad, m, v, lr_t, b1, b2, eps,
                     clip, grad_scale);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_clip_f32(size_t n, float *g, float clip, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(g && (uintptr_t)g % 16 == 0, "clip: null or unaligned pointer");
  hipLaunchKernelGGL(clip_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, g, clip);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t nabu_colsum_ws_bytes(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (size_t)((M + CS_ROWS - 1) / CS_ROWS) * N * sizeof(float);
}

extern "C" int nabu_colsum_f32(int M, int N, const float *A, int lda, float beta, float *out,
                               void *ws, size_t ws_bytes, nabu_stream_t stream) {
  NABU_CHECK_ARG(M >= 0 && N > 0 && A && out, "colsum: bad argument");
  const int parts = (M + CS_ROWS - 1) / CS_ROWS;
  const size_t need = (size_t)parts * N * sizeof(float);
  if (parts > 0 && (!ws || ws_bytes < need)) return fail(NABU_EWS, "colsum: workspace %zu < %zu", ws_bytes, need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (parts > 0) {
    hipLaunchKernelGGL(colsum_stage1, dim3((N + 63) / 64, parts), dim3(256), 0, s, M, N, A, lda,
                       static_cast<float *>(ws));
    NABU_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(colsum_stage2, dim3((N + 255) / 256), dim3(256), 0, s, parts, N,
                     static_cast<const float *>(ws), beta, out);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_pad_time_f32(int B, int T, int Tp, int F, const float *x, float *y,
                                 nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && T > 0 && Tp >= T && F > 0 && F % 4 == 0 && x && y, "pad_time: bad argument");
  hipLaunchKernelGGL(pad_time_kernel, dim3(grid_for((size_t)B * Tp * (F / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, T, Tp, F, x, y, 0);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_unpad_time_f32(int B, int T, int Tp, int F, const float *y, float *x,
                                   nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && T > 0 && Tp >= T && F > 0 && F % 4 == 0 && x && y, "unpad_time: bad argument");
  hipLaunchKernelGGL(pad_time_kernel, dim3(grid_for((size_t)B * T * (F / 4))), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, T, Tp, F, y, x, 1);
  NABU_LAUNCH_CHECK();
  return 0;
}
