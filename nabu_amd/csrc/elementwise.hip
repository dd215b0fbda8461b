// elementwise.hip — HBM-bound helpers: fused clip+Adam, bias-gradient column
// sums, time padding for odd-length pyramid stacks.
#include "common.h"

namespace nabu {

// tf.clip_by_value propagates NaN (min/max of the TF kernels are plain comparisons), fminf/fmaxf
// return the non-NaN operand: a diverged gradient must stay visible, not become -clip.
__device__ __forceinline__ float clip_value(float x, float clip) {
  return x != x ? x : fminf(fmaxf(x, -clip), clip);
}

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
    float x = clip_value(gg.c * gscale, clip);                  \
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
    float x = clip_value(g[i] * gscale, clip);
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
    x.x = clip_value(x.x, clip);
    x.y = clip_value(x.y, clip);
    x.z = clip_value(x.z, clip);
    x.w = clip_value(x.w, clip);
    reinterpret_cast<float4 *>(g)[i] = x;
  }
  for (size_t i = n4 * 4 + tid; i < n; i += stride) g[i] = clip_value(g[i], clip);
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


// counter-based RNG (Philox4x32-10): philox4x32_10 / u01 in common.h (shared with the persistent decoder)

// i0: index of the first 4-element group of x inside the array the random stream is defined on (a
// sub-batch of rows draws exactly the numbers the whole batch would draw for those rows)
__global__ __launch_bounds__(256) void dropout_kernel(size_t n, const float *__restrict__ x,
                                                      float *__restrict__ y, float keep,
                                                      unsigned long long seed, unsigned long long offset, size_t i0) {
  const size_t n4 = (n + 3) / 4;
  const float inv = 1.0f / keep;
  for (size_t il = (size_t)blockIdx.x * 256 + threadIdx.x; il < n4; il += (size_t)gridDim.x * 256) {
    const size_t i = il + i0;
    const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(i >> 32), (unsigned)offset,
                                             (unsigned)(offset >> 32)),
                                  make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    const unsigned rr[4] = {r.x, r.y, r.z, r.w};
    if (4 * il + 3 < n && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
      // whole group, 16-byte aligned arrays: one load, one store
      const float4 v = *reinterpret_cast<const float4 *>(x + 4 * il);
      *reinterpret_cast<float4 *>(y + 4 * il) = make_float4(u01(rr[0]) < keep ? v.x * inv : 0.f, u01(rr[1]) < keep ? v.y * inv : 0.f,
                                                            u01(rr[2]) < keep ? v.z * inv : 0.f, u01(rr[3]) < keep ? v.w * inv : 0.f);
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t e = 4 * il + j;
      if (e < n) y[e] = u01(rr[j]) < keep ? x[e] * inv : 0.f;
    }
  }
}

// scheduled sampling (ScheduledEmbeddingTrainingHelper): one thread per row
__global__ __launch_bounds__(256) void sample_ids_kernel(int B, int C, const float *__restrict__ logits,
                                                         float prob, unsigned long long seed,
                                                         unsigned long long offset,
                                                         const int32_t *__restrict__ teacher,
                                                         int32_t *__restrict__ out, int b0) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const uint4 r = philox4x32_10(make_uint4((unsigned)(b + b0), 0u, (unsigned)offset, (unsigned)(offset >> 32)),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  int id = teacher[b];
  if (u01(r.x) < prob) {
    const float *l = logits + (size_t)b * C;
    float m = l[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, l[c]);
    float tot = 0.f;
    for (int c = 0; c < C; ++c) tot += expf(l[c] - m);
    const float target = u01(r.y) * tot;        // inverse CDF of softmax(l)
    float acc = 0.f;
    id = C - 1;
    for (int c = 0; c < C; ++c) {
      acc += expf(l[c] - m);
      if (acc > target) { id = c; break; }
    }
  }
  out[b] = id;
}

// Scheduled sampling of ONE decoder step in ONE launch (the step chain of nabu_speller_fwd): row b draws the Bernoulli of
// sample_ids_kernel first and only a row that IS to be sampled (sample_prob = 0.1 in the reference's defaults: one in
// ten) evaluates its logits [h | ctx] . W_out + b — 25 k slots x (C / 4) class quads, 16-byte loads of the weight rows —
// and draws from softmax(logits) by the inverse CDF in class order, exactly as sample_ids_kernel does.  Replaces two
// [Bn, C] products through the general GEMM entry point and the sampling launch per step and sub-batch
// (rnn_decoder.py:59-66, ScheduledEmbeddingTrainingHelper).  Requires C % 4 == 0, C <= 256.
constexpr int SAMPLE_KS = 25;
__global__ __launch_bounds__(256) void sample_step_kernel(int C, int U, int E, const float *__restrict__ h, int ldh,
                                                          const float *__restrict__ ctx, int ldc,
                                                          const float *__restrict__ Wout, const float *__restrict__ bias,
                                                          float prob, unsigned long long seed, unsigned long long offset,
                                                          const int32_t *__restrict__ teacher, int32_t *__restrict__ out, int b0) {
  __shared__ float part[SAMPLE_KS][256];
  __shared__ float lg[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint4 r = philox4x32_10(make_uint4((unsigned)(b + b0), 0u, (unsigned)offset, (unsigned)(offset >> 32)),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  if (!(u01(r.x) < prob)) {
    if (tid == 0) out[b] = teacher[b];
    return;
  }
  const int CQ = C / 4, ks = tid / CQ, cq = tid - ks * CQ, K = U + E;
  if (ks < SAMPLE_KS) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *hb = h + (size_t)b * ldh, *cb = ctx + (size_t)b * ldc;
#pragma unroll 8
    for (int k = ks; k < K; k += SAMPLE_KS) {
      const float x = k < U ? hb[k] : cb[k - U];
      const float4 wv = *reinterpret_cast<const float4 *>(Wout + (size_t)k * C + 4 * cq);
      acc.x = fmaf(x, wv.x, acc.x); acc.y = fmaf(x, wv.y, acc.y); acc.z = fmaf(x, wv.z, acc.z); acc.w = fmaf(x, wv.w, acc.w);
    }
    *reinterpret_cast<float4 *>(&part[ks][4 * cq]) = acc;
  }
  __syncthreads();
  if (tid < C) {
    float v = bias[tid];
    const int nks = min(SAMPLE_KS, 256 / CQ);
    for (int i = 0; i < nks; ++i) v += part[i][tid];
    lg[tid] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float m = lg[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, lg[c]);
    float tot = 0.f;
    for (int c = 0; c < C; ++c) tot += expf(lg[c] - m);
    const float target = u01(r.y) * tot;        // inverse CDF of softmax(l)
    float acc = 0.f;
    int id = C - 1;
    for (int c = 0; c < C; ++c) {
      acc += expf(lg[c] - m);
      if (acc > target) { id = c; break; }
    }
    out[b] = id;
  }
}

__global__ __launch_bounds__(256) void gaussian_noise_kernel(size_t n, const float *__restrict__ x,
                                                             float *__restrict__ y, float stddev,
                                                             unsigned long long seed,
                                                             unsigned long long offset) {
  const size_t n4 = (n + 3) / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const uint4 r = philox4x32_10(make_uint4((unsigned)i, (unsigned)(i >> 32), (unsigned)offset,
                                             (unsigned)(offset >> 32)),
                                  make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
    // Box-Muller on (0,1] uniforms
    const float u1 = 1.0f - u01(r.x), u2 = u01(r.y), u3 = 1.0f - u01(r.z), u4 = u01(r.w);
    const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float sa, ca, sb, cb;
    sincosf(6.283185307179586f * u2, &sa, &ca);
    sincosf(6.283185307179586f * u4, &sb, &cb);
    const float z[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t e = 4 * i + j;
      if (e < n) y[e] = x[e] + stddev * z[j];
    }
  }
}

// out[0] = scale * sum(x[0..n)) — one block, fixed tree (deterministic).
__global__ __launch_bounds__(256) void sum_kernel(size_t n, const float *__restrict__ x, float scale,
                                                  float *__restrict__ out) {
  __shared__ float red[256];
  float s = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) s += x[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = scale * red[0];
}

__global__ __launch_bounds__(256) void axpy_kernel(size_t n, float a, const float *__restrict__ x,
                                                   float *__restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = fmaf(a, x[i], y[i]);
}

__global__ __launch_bounds__(256) void relu_kernel(size_t n, const float *__restrict__ x, float *__restrict__ y) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(size_t n, const float *__restrict__ y,
                                                       const float *__restrict__ dy, float *__restrict__ dx) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// block-wide sums of two values (fixed tree: deterministic)
__device__ __forceinline__ void block_sum2(float &a, float &b, float *red) {
  red[threadIdx.x] = a;
  red[256 + threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[threadIdx.x] += red[threadIdx.x + o];
      red[256 + threadIdx.x] += red[256 + threadIdx.x + o];
    }
    __syncthreads();
  }
  a = red[0];
  b = red[256];
  __syncthreads();
}

// one workgroup per batch row; moments over all N = T*F elements of the row
__global__ __launch_bounds__(256) void layer_norm_fwd_kernel(int N, int F, const float *__restrict__ x,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ beta, float eps,
                                                             float *__restrict__ y, float *__restrict__ mean,
                                                             float *__restrict__ rstd) {
  __shared__ float red[512];
  const float *xr = x + (size_t)blockIdx.x * N;
  float *yr = y + (size_t)blockIdx.x * N;
  float s = 0.f, dummy = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) s += xr[i];
  block_sum2(s, dummy, red);
  const float mu = s / N;
  float v = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { const float d = xr[i] - mu; v = fmaf(d, d, v); }
  block_sum2(v, dummy, red);
  const float rs = rsqrtf(v / N + eps);
  for (int i = threadIdx.x; i < N; i += 256) yr[i] = (xr[i] - mu) * rs * gamma[i % F] + beta[i % F];
  if (threadIdx.x == 0) { mean[blockIdx.x] = mu; rstd[blockIdx.x] = rs; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma; partial dgamma/dbeta per row
__global__ __launch_bounds__(256) void layer_norm_bwd_kernel(int N, int F, const float *__restrict__ x,
                                                             const float *__restrict__ gamma,
                                                             const float *__restrict__ dy,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ rstd, float *__restrict__ dx,
                                                             float *__restrict__ dgp, float *__restrict__ dbp) {
  __shared__ float red[512];
  const size_t row = (size_t)blockIdx.x * N;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float g = dy[row + i] * gamma[i % F], xh = (x[row + i] - mu) * rs;
    s1 += g;
    s2 = fmaf(g, xh, s2);
  }
  block_sum2(s1, s2, red);
  const float m1 = s1 / N, m2 = s2 / N;
  for (int i = threadIdx.x; i < N; i += 256) {
    const float g = dy[row + i] * gamma[i % F], xh = (x[row + i] - mu) * rs;
    dx[row + i] = rs * (g - m1 - xh * m2);
  }
  // per-feature sums over the row's T frames (thread f walks its column: fixed order)
  for (int f = threadIdx.x; f < F; f += 256) {
    float a = 0.f, b = 0.f;
    for (int i = f; i < N; i += F) {
      const float d = dy[row + i];
      a = fmaf(d, (x[row + i] - mu) * rs, a);
      b += d;
    }
    dgp[(size_t)blockIdx.x * F + f] = a;
    dbp[(size_t)blockIdx.x * F + f] = b;
  }
}

__global__ __launch_bounds__(256) void ceil_div_kernel(int n, const int32_t *__restrict__ in, int d,
                                                       int32_t *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (in[i] + d - 1) / d;
}

__global__ __launch_bounds__(256) void colsum_pair_kernel(int rows, int N, const float *__restrict__ part, int ld,
                                                         float *__restrict__ out0, float *__restrict__ out1) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= 2 * N) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += part[(size_t)r * ld + c];
  if (c < N) out0[c] = s;
  else out1[c - N] = s;
}

// several regions, one launch (common.h: multi_fill).  A block covers FILL_CHUNK words of one region.
constexpr int FILL_MAX = 8;
constexpr unsigned FILL_CHUNK = 4096;
struct FillArgs {
  unsigned *ptr[FILL_MAX];
  unsigned long long words[FILL_MAX];
  unsigned value[FILL_MAX];
  unsigned first_block[FILL_MAX + 1];
  int n;
};
__global__ __launch_bounds__(256) void multi_fill_kernel(FillArgs a) {
  int i = 0;
  while (i + 1 < a.n && blockIdx.x >= a.first_block[i + 1]) ++i;
  const unsigned long long w0 = (unsigned long long)(blockIdx.x - a.first_block[i]) * FILL_CHUNK;
  const unsigned long long w1 = w0 + FILL_CHUNK < a.words[i] ? w0 + FILL_CHUNK : a.words[i];
  const unsigned v = a.value[i];
  unsigned *p = a.ptr[i];
  for (unsigned long long w = w0 + 4ull * threadIdx.x; w < w1; w += 1024) {
    if (w + 3 < w1) *reinterpret_cast<uint4 *>(p + w) = make_uint4(v, v, v, v);
    else
      for (unsigned long long k = w; k < w1; ++k) p[k] = v;
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

extern "C" int nabu_dropout_f32(size_t n, const float *x, float *y, float keep_prob,
                                unsigned long long seed, unsigned long long offset,
                                nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(x && y && keep_prob > 0.f && keep_prob <= 1.f, "dropout: bad argument");
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, x, y, keep_prob, seed, offset, (size_t)0);
  NABU_LAUNCH_CHECK();
  return 0;
}

namespace nabu {
int colsum_pair(int rows, int N, const float *part, int ld, float *out0, float *out1, hipStream_t stream) {
  if (rows < 0 || N <= 0 || !part || !out0 || !out1) return fail(NABU_EINVAL, "colsum_pair: bad argument");
  hipLaunchKernelGGL(colsum_pair_kernel, dim3((2 * N + 255) / 256), dim3(256), 0, stream, rows, N, part, ld, out0, out1);
  NABU_LAUNCH_CHECK();
  return 0;
}

int multi_fill(const FillSeg *segs, int n, hipStream_t stream) {
  FillArgs a = {};
  unsigned blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!segs[i].words) continue;
    if (a.n >= FILL_MAX) return fail(NABU_EINVAL, "multi_fill: more than %d regions", FILL_MAX);
    if (!segs[i].ptr || (reinterpret_cast<uintptr_t>(segs[i].ptr) & 15)) return fail(NABU_EINVAL, "multi_fill: null or unaligned region");
    a.ptr[a.n] = static_cast<unsigned *>(segs[i].ptr);
    a.words[a.n] = segs[i].words;
    a.value[a.n] = segs[i].value;
    a.first_block[a.n] = blocks;
    blocks += (unsigned)((segs[i].words + FILL_CHUNK - 1) / FILL_CHUNK);
    ++a.n;
  }
  if (!a.n) return 0;
  a.first_block[a.n] = blocks;
  hipLaunchKernelGGL(multi_fill_kernel, dim3(blocks), dim3(256), 0, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// rows [first_elem, first_elem + n) of a larger array (first_elem % 4 == 0): the same random numbers the
// call on the whole array would use for them
int dropout_rows(size_t n, const float *x, float *y, float keep_prob, unsigned long long seed, unsigned long long offset,
                 size_t first_elem, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, stream, n, x, y, keep_prob, seed, offset,
                     first_elem / 4);
  NABU_LAUNCH_CHECK();
  return 0;
}
bool sample_step_ok(int C) { return C % 4 == 0 && C >= 4 && C <= 256; }
int sample_step(int B, int C, int U, int E, const float *h, int ldh, const float *ctx, int ldc, const float *Wout,
                const float *bias, float prob, unsigned long long seed, unsigned long long offset, const int32_t *teacher_ids,
                int32_t *out_ids, int b0, hipStream_t stream) {
  if (B == 0) return 0;
  if (!sample_step_ok(C)) return fail(NABU_EUNSUP, "sample_step: C = %d", C);
  hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(256), 0, stream, C, U, E, h, ldh, ctx, ldc, Wout, bias, prob, seed,
                     offset, teacher_ids, out_ids, b0);
  NABU_LAUNCH_CHECK();
  return 0;
}
int sample_ids_rows(int B, int C, const float *logits, float prob, unsigned long long seed, unsigned long long offset,
                    const int32_t *teacher_ids, int32_t *out_ids, int b0, hipStream_t stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(sample_ids_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, B, C, logits, prob, seed, offset,
                     teacher_ids, out_ids, b0);
  NABU_LAUNCH_CHECK();
  return 0;
}
}  // namespace nabu

extern "C" int nabu_sample_ids(int B, int C, const float *logits, float prob, unsigned long long seed,
                               unsigned long long offset, const int32_t *teacher_ids, int32_t *out_ids,
                               nabu_stream_t stream) {
  if (B == 0) return 0;
  NABU_CHECK_ARG(B > 0 && C > 0 && logits && teacher_ids && out_ids && prob >= 0.f && prob <= 1.f,
                 "sample_ids: bad argument");
  hipLaunchKernelGGL(sample_ids_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     B, C, logits, prob, seed, offset, teacher_ids, out_ids, 0);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_gaussian_noise_f32(size_t n, const float *x, float *y, float stddev,
                                       unsigned long long seed, unsigned long long offset,
                                       nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(x && y && stddev >= 0.f, "gaussian_noise: bad argument");
  hipLaunchKernelGGL(gaussian_noise_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), n, x, y, stddev, seed, offset);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_sum_f32(size_t n, const float *x, float scale, float *out, nabu_stream_t stream) {
  NABU_CHECK_ARG(out && (n == 0 || x), "sum: null pointer");
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), n, x, scale, out);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_axpy_f32(size_t n, float a, const float *x, float *y, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(x && y, "axpy: null pointer");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), n, a, x, y);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_relu_f32(size_t n, const float *x, float *y, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(x && y, "relu: null pointer");
  hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), n, x, y);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_relu_bwd_f32(size_t n, const float *y, const float *dy, float *dx, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(y && dy && dx, "relu_bwd: null pointer");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), n, y, dy, dx);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_layer_norm_fwd(int B, int N, int F, const float *x, const float *gamma, const float *beta,
                                   float eps, float *y, float *mean, float *rstd, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && N > 0 && F > 0 && N % F == 0 && x && gamma && beta && y && mean && rstd,
                 "layer_norm_fwd: bad argument");
  hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), N, F, x, gamma,
                     beta, eps, y, mean, rstd);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_layer_norm_bwd(int B, int N, int F, const float *x, const float *gamma, const float *dy,
                                   const float *mean, const float *rstd, float *dx, float *dgamma_part,
                                   float *dbeta_part, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && N > 0 && F > 0 && N % F == 0 && x && gamma && dy && mean && rstd && dx && dgamma_part &&
                     dbeta_part, "layer_norm_bwd: bad argument");
  hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), N, F, x, gamma,
                     dy, mean, rstd, dx, dgamma_part, dbeta_part);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_ceil_div_i32(int n, const int32_t *in, int d, int32_t *out, nabu_stream_t stream) {
  if (n == 0) return 0;
  NABU_CHECK_ARG(n > 0 && d > 0 && in && out, "ceil_div: bad argument");
  hipLaunchKernelGGL(ceil_div_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), n, in, d, out);
  NABU_LAUNCH_CHECK();
  return 0;
}
