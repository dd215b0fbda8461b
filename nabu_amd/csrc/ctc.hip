// ctc.hip — CTC negative log-likelihood and its gradient w.r.t. the logits
// (tf.nn.ctc_loss semantics: softmax inside, blank = C-1, merge_repeated).
//
// One workgroup (4 wave64) per utterance.  The S = 2L+1 lattice states map to
// threads; the alpha/beta rows of the current and previous frame live in LDS
// (log space, float32, log-sum-exp); alpha is spilled to an HBM workspace
// [B,T,Smax] (L2-resident at these sizes) and re-read by the beta sweep, which
// emits the gradient frame by frame:
//   dlogits[t,c] = scale * ( softmax[t,c] - sum_{s: ext[s]=c} gamma[t,s] ),
//   gamma[t,s] = exp(alpha[t,s] + beta[t,s] - y[t,ext[s]] + nll).
// The per-class sums are deterministic (fixed order, no float atomics): blank
// states are tree-reduced by wave 0, every other class walks the linked list of
// its occurrences.  Algorithmic HBM bytes: read logits once per sweep + write
// dlogits = 3*B*T*C*4, plus 2*B*T*S*4 for the alpha spill.
#include "common.h"

namespace nabu {

constexpr float CTC_NEG = -1e30f;  // stands in for log(0); keeps inf-inf out of the math

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b), n = fminf(a, b);
  return m + log1pf(expf(n - m));
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  // accurate expf/logf: the recursion runs T times and its rounding errors add up
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

struct CtcArgs {
  int B, T, C, Lmax, Smax;
  const float *logits;
  const int32_t *logit_len, *labels, *label_len;
  float scale;
  float *nll, *dlogits, *alpha;
  int32_t *status;
};

__global__ __launch_bounds__(256) void ctc_kernel(CtcArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int T = p.T, C = p.C, Smax = p.Smax, blank = p.C - 1;
  float *lse = smem;                       // [T]
  float *row0 = lse + T;                   // [Smax] alpha/beta ping
  float *row1 = row0 + Smax;               // [Smax] alpha/beta pong
  float *gam = row1 + Smax;                // [Smax]
  float *csum = gam + Smax;                // [C]
  int *ext = reinterpret_cast<int *>(csum + C);  // [Smax]
  int *nxt = ext + Smax;                   // [Smax] next state with the same label
  int *head = nxt + Smax;                  // [C]
  __shared__ int s_bad;
  __shared__ float s_ll;

  int Tb = p.logit_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  int L = p.label_len[b];
  L = L < 0 ? 0 : (L > p.Lmax ? p.Lmax : L);
  const int S = 2 * L + 1;
  const float *lg = p.logits + (size_t)b * T * C;
  float *dl = p.dlogits + (size_t)b * T * C;
  float *al = p.alpha + (size_t)b * T * Smax;

  if (tid == 0) s_bad = 0;
  for (int s = tid; s < S; s += 256) ext[s] = (s & 1) ? p.labels[(size_t)b * p.Lmax + (s >> 1)] : blank;
  for (int c = tid; c < C; c += 256) head[c] = -1;
  // frames past the sequence end: zero gradient
  for (int i = Tb * C + tid; i < T * C; i += 256) dl[i] = 0.f;
  // per-frame log-sum-exp of the logits
  for (int t = tid; t < Tb; t += 256) {
    const float *x = lg + (size_t)t * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(x[c] - m);
    lse[t] = m + logf(z);
  }
  __syncthreads();
  if (tid == 0) {
    int rep = 0, bad = 0;
    for (int i = 0; i < L; ++i) {
      const int l = ext[2 * i + 1];
      if (l < 0 || l >= blank) bad = 1;
      if (i > 0 && l == ext[2 * i - 1]) ++rep;
    }
    if (Tb <= 0 || L + rep > Tb) bad = 1;   // tf: "Not enough time for target transition sequence"
    s_bad = bad;
    if (!bad)
      for (int s = S - 2; s >= 1; s -= 2) {  // occurrence lists in increasing-s order
        nxt[s] = head[ext[s]];
        head[ext[s]] = s;
      }
  }
  __syncthreads();
  if (s_bad) {
    for (int i = tid; i < Tb * C; i += 256) dl[i] = 0.f;
    if (tid == 0) {
      p.nll[b] = __builtin_inff();
      atomicCAS(p.status, 0, b + 1);
    }
    return;
  }

  // ---- alpha sweep --------------------------------------------------------
  float *prev = row0, *cur = row1;
  for (int s = tid; s < S; s += 256) {
    float a = CTC_NEG;
    if (s < 2) a = lg[ext[s]] - lse[0];
    prev[s] = a;
    al[s] = a;
  }
  __syncthreads();
  for (int t = 1; t < Tb; ++t) {
    const float *x = lg + (size_t)t * C;
    const float z = lse[t];
    for (int s = tid; s < S; s += 256) {
      const int e = ext[s];
      const float a0 = prev[s];
      const float a1 = s >= 1 ? prev[s - 1] : CTC_NEG;
      const float a2 = (s >= 2 && e != blank && e != ext[s - 2]) ? prev[s - 2] : CTC_NEG;
      const float a = lse3(a0, a1, a2) + (x[e] - z);
      cur[s] = a;
      al[(size_t)t * Smax + s] = a;
    }
    __syncthreads();
    float *tmp = prev; prev = cur; cur = tmp;
  }
  if (tid == 0) {
    const float ll = S > 1 ? lse2(prev[S - 1], prev[S - 2]) : prev[0];
    s_ll = ll;
    p.nll[b] = -ll;
  }
  __syncthreads();
  const float ll = s_ll;

  // ---- beta sweep + gradient ------------------------------------------------
  float *bnext = row0, *bcur = row1;
  for (int t = Tb - 1; t >= 0; --t) {
    const float *x = lg + (size_t)t * C;
    const float z = lse[t];
    for (int s = tid; s < S; s += 256) {
      const int e = ext[s];
      const float ye = x[e] - z;
      float bt;
      if (t == Tb - 1) {
        bt = (s >= S - 2) ? ye : CTC_NEG;
      } else {
        const float b0 = bnext[s];
        const float b1 = s + 1 < S ? bnext[s + 1] : CTC_NEG;
        const float b2 = (s + 2 < S && ext[s + 2] != blank && ext[s + 2] != e) ? bnext[s + 2] : CTC_NEG;
        bt = lse3(b0, b1, b2) + ye;
      }
      bcur[s] = bt;
      gam[s] = expf((al[(size_t)t * Smax + s] - ll) + (bt - ye));
    }
    __syncthreads();
    if (tid < 64) {  // blank: even states, fixed-shape tree
      float v = 0.f;
      for (int s = 2 * tid; s < S; s += 128) v += gam[s];
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
      if (tid == 0) csum[blank] = v;
    } else {
      for (int c = tid - 64; c < blank; c += 192) {
        float v = 0.f;
        for (int s = head[c]; s >= 0; s = nxt[s]) v += gam[s];
        csum[c] = v;
      }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256)
      dl[(size_t)t * C + c] = p.scale * (expf(x[c] - z) - csum[c]);
    float *tmp = bnext; bnext = bcur; bcur = tmp;
  }
}

static size_t ctc_lds_bytes(int T, int C, int Smax) {
  return ((size_t)T + 3 * Smax + C) * sizeof(float) + ((size_t)2 * Smax + C) * sizeof(int);
}

}  // namespace nabu

using namespace nabu;

extern "C" size_t nabu_ctc_ws_bytes(int B, int T, int Lmax) {
  if (B <= 0 || T <= 0 || Lmax < 0) return 0;
  return (size_t)B * T * (2 * (size_t)Lmax + 1) * sizeof(float);
}

extern "C" int nabu_ctc_loss_grad(int B, int T, int C, int Lmax, const float *logits,
                                  const int32_t *logit_len, const int32_t *labels,
                                  const int32_t *label_len, float grad_scale, float *nll,
                                  float *dlogits, int32_t *status, void *ws, size_t ws_bytes,
                                  nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && T > 0 && C > 1 && Lmax >= 0, "ctc: bad dimensions");
  NABU_CHECK_ARG(logits && logit_len && label_len && nll && dlogits && status && ws, "ctc: null pointer");
  NABU_CHECK_ARG(Lmax == 0 || labels, "ctc: null labels");
  const int Smax = 2 * Lmax + 1;
  const size_t need = nabu_ctc_ws_bytes(B, T, Lmax);
  if (ws_bytes < need) return fail(NABU_EWS, "ctc: workspace %zu < %zu", ws_bytes, need);
  const size_t shm = ctc_lds_bytes(T, C, Smax);
  if (shm > 150 * 1024) return fail(NABU_EUNSUP, "ctc: T=%d, Lmax=%d, C=%d need %zu B of LDS", T, Lmax, C, shm);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  CtcArgs p;
  p.B = B; p.T = T; p.C = C; p.Lmax = Lmax; p.Smax = Smax;
  p.logits = logits; p.logit_len = logit_len; p.labels = labels; p.label_len = label_len;
  p.scale = grad_scale; p.nll = nll; p.dlogits = dlogits; p.status = status;
  p.alpha = static_cast<float *>(ws);
  NABU_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(ctc_kernel, dim3(B), dim3(256), shm, s, p);
  NABU_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// Masked sparse softmax cross-entropy averaged over the target length
// (loss_functions.average_cross_entropy / cross_entropy of the reference):
//   loss[b] = sum_{t < logit_len[b]} -log softmax(logits[b,t])[targets[b,t]] / target_len[b]
//   dlogits[b,t,:] = grad_scale * (softmax - onehot) / target_len[b]   (0 for t >= logit_len[b])
// One workgroup per utterance, one thread per frame, fixed-order block reduction.
namespace nabu {
__global__ __launch_bounds__(256) void xent_kernel(int B, int L, int C, int ldt,
                                                   const float *__restrict__ logits,
                                                   const int32_t *__restrict__ targets,
                                                   const int32_t *__restrict__ logit_len,
                                                   const int32_t *__restrict__ target_len,
                                                   float grad_scale, float *__restrict__ loss,
                                                   float *__restrict__ dlogits) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  const int n = min(max(logit_len[b], 0), L);
  const float inv = 1.0f / (float)target_len[b];
  float acc = 0.f;
  for (int t = threadIdx.x; t < L; t += 256) {
    const float *x = logits + ((size_t)b * L + t) * C;
    float *d = dlogits + ((size_t)b * L + t) * C;
    if (t >= n) {
      for (int c = 0; c < C; ++c) d[c] = 0.f;
      continue;
    }
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += expf(x[c] - m);
    const float lz = m + logf(z);
    const int y = targets[(size_t)b * ldt + t];
    acc += lz - x[y];
    const float s = grad_scale * inv;
    for (int c = 0; c < C; ++c) d[c] = s * (expf(x[c] - lz) - (c == y ? 1.f : 0.f));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[b] = red[0] * inv;
}
}  // namespace nabu

extern "C" int nabu_xent_loss_grad(int B, int L, int C, int ldt, const float *logits,
                                   const int32_t *targets, const int32_t *logit_len,
                                   const int32_t *target_len, float grad_scale, float *loss,
                                   float *dlogits, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && L > 0 && C > 0 && ldt >= L, "xent: bad dimensions");
  NABU_CHECK_ARG(logits && targets && logit_len && target_len && loss && dlogits, "xent: null pointer");
  hipLaunchKernelGGL(nabu::xent_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), B, L, C,
                     ldt, logits, targets, logit_len, target_len, grad_scale, loss, dlogits);
  NABU_LAUNCH_CHECK();
  return 0;
}
