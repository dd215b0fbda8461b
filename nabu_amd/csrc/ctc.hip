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
#include <stdlib.h>

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


// ---------------------------------------------------------------------------
// Wave-synchronous variant: ONE wave64 per utterance, no workgroup barriers.  Lane i owns the state
// pair (blank 2i, label 2i+1) of the lattice, so a frame of the alpha (beta) recursion needs exactly one
// (two) neighbour values, fetched with a DPP wave shift; the frame's logits are prefetched one frame
// ahead, its log-sum-exp is reduced with DPP while the recursion of the current frame waits for its
// exp/log latencies.  For label sequences of up to 63 symbols and up to 256 classes (cfg1-cfg5:
// <= 60 phones, 40 classes); longer ones take the workgroup kernel above.  Same arithmetic, same
// outputs, deterministic (class sums walk fixed occurrence chains).
__device__ __forceinline__ float dpp_up(float v, float fill) {     // lane i <- lane i-1 (lane 0 <- fill)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_dn(float v, float fill) {     // lane i <- lane i+1 (lane 63 <- fill)
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
// log-sum-exp on the hardware exp2/log2 (1 ulp each; the argument scaling adds ~|x| 2^-24 relative):
// the library expf/logf/log1pf cost 10-30 dependent instructions each and sit on the frame-to-frame chain
// The wave kernel keeps the lattice in BASE-2 logarithms, so log-sum-exp is max + v_log_f32(sum of
// v_exp_f32) with no scaling on the chain (arguments of exp2 are <= 0, of log2 in [1,3]: no denormal or
// range handling needed; both instructions are accurate to 1 ulp).
__device__ __forceinline__ float lse2_fast(float a, float b) {
  const float m = fmaxf(a, b), n = fminf(a, b);
  return m + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(n - m));
}
__device__ __forceinline__ float lse3_fast(float a, float b, float c) {
  const float m = fmaxf(a, fmaxf(b, c));
  return m + __builtin_amdgcn_logf(__builtin_amdgcn_exp2f(a - m) + __builtin_amdgcn_exp2f(b - m) +
                                   __builtin_amdgcn_exp2f(c - m));
}
// sum over the wave with DPP row shifts / broadcasts (no LDS crossbar); the total is returned to all lanes
#define NABU_DPP_ADD(v, ctrl, rmask, bmask) \
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, bmask, true))
__device__ __forceinline__ float wave_sum_dpp(float v) {
  NABU_DPP_ADD(v, 0x111, 0xf, 0xf);     // row_shr:1
  NABU_DPP_ADD(v, 0x112, 0xf, 0xf);     // row_shr:2
  NABU_DPP_ADD(v, 0x114, 0xf, 0xe);     // row_shr:4
  NABU_DPP_ADD(v, 0x118, 0xf, 0xc);     // row_shr:8
  NABU_DPP_ADD(v, 0x142, 0xa, 0xf);     // row_bcast:15
  NABU_DPP_ADD(v, 0x143, 0xc, 0xf);     // row_bcast:31
  return __shfl(v, 63);
}
#undef NABU_DPP_ADD
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per workgroup: LDS traffic between its lanes needs no s_barrier (LDS operations of a wave
// execute in order, so a read issued after a write of another lane sees it) — only that the compiler
// keeps the order.
#define WSYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
__global__ __launch_bounds__(256) void ctc_wave_kernel(CtcArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = threadIdx.x & 63;
  const int T = p.T, C = p.C, Smax = p.Smax, blank = p.C - 1;
  // everything the two sweeps touch lives in LDS: the utterance's logits, their per-frame
  // log-sum-exp and the alpha lattice (HBM latency would otherwise sit on every frame)
  float *xs = smem;                  // [T*C] logits
  float *lse = xs + (size_t)T * C;   // [T]
  float *as = lse + T;               // [T*Smax] alpha
  float *csum = as + (size_t)T * Smax;   // [C] per-class occupation of the current frame
  float *gam = csum + C;             // [64]
  int *lbl = reinterpret_cast<int *>(gam + 64);   // [64] labels
  int *nxtl = lbl + 64;                           // [64] next lane with the same label

  int Tb = p.logit_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  int L = p.label_len[b];
  L = L < 0 ? 0 : (L > p.Lmax ? p.Lmax : L);
  const float *lg = p.logits + (size_t)b * T * C;
  float *dl = p.dlogits + (size_t)b * T * C;

  // prologue, all four waves: stage the logits, zero the gradient of the padding frames, per-frame
  // log-sum-exp (threads over frames); then wave 0 runs the two sweeps alone
  {
    const int n = Tb * C;
    if ((reinterpret_cast<uintptr_t>(lg) & 15) == 0 && (n & 3) == 0) {
      const float4 *src = reinterpret_cast<const float4 *>(lg);
      float4 *dst = reinterpret_cast<float4 *>(xs);
#pragma unroll 4
      for (int i = tid; i < n / 4; i += 256) dst[i] = src[i];
    } else {
#pragma unroll 4
      for (int i = tid; i < n; i += 256) xs[i] = lg[i];
    }
    for (int i = n + tid; i < T * C; i += 256) dl[i] = 0.f;
    for (int c = tid; c < C; c += 256) csum[c] = 0.f;
    if (tid < 2) (nxtl + 64)[tid] = 0;           // the sweep / worker hand-over words (below)
  }
  const int lab = lane < L ? p.labels[(size_t)b * p.Lmax + lane] : -1;
  if (tid < 64) lbl[lane] = lab;
  __syncthreads();
  for (int t = tid; t < Tb; t += 256) {
    const float *x = xs + (size_t)t * C;
    float m = x[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float z = 0.f;
    for (int c = 0; c < C; ++c) z += __expf(x[c] - m);
    lse[t] = m + __logf(z);
  }
  __syncthreads();
  // ONE wave runs the two sweeps (the dependent chain: 2 T frames of a few dozen instructions); the other three turn the
  // frames it has finished into gradients BESIDE it — occupation sums per class, softmax, the stores: what used to sit
  // in the beta sweep's own instruction stream (round 5).  Hand-over through LDS: the sweeping wave overwrites alpha[t]
  // with the frame's occupations and then raises `prog` (release); a worker polls it (acquire).  LDS operations of a
  // wave are performed in order, so a worker that sees the counter sees the frame.
  const int wv = tid >> 6;
  int *prog = nxtl + 64;                 // [0] frames finished by the beta sweep, [1] 1 = go
  float *llp = reinterpret_cast<float *>(prog + 2);
  float *gamw = llp + 2;                 // [3][64] occupation of the label states, per worker wave
  float *csumw = gamw + 3 * 64;          // [3][C]  class sums of the worker's current frame
  const int lab_up = lane >= 1 ? lbl[lane - 1] : -1, lab_dn = lane < 63 ? lbl[lane + 1] : -1;
  const bool badl = lane < L && (lab < 0 || lab >= blank);
  const int rep = (int)wave_sum64((lane >= 1 && lane < L && lab == lab_up) ? 1.f : 0.f);
  const bool bad = __any(badl) || Tb <= 0 || L + rep > Tb;   // tf: "Not enough time for target transition sequence"
  // occurrence chains of the labels (fixed order -> deterministic class sums); every wave keeps its own copy in registers
  bool first = lane < L;
  int nxt = -1;
  if (lane < L) {
    for (int j = 0; j < lane; ++j) first = first && lbl[j] != lab;
    for (int j = L - 1; j > lane; --j) nxt = lbl[j] == lab ? j : nxt;
  }
  const bool hasB = lane <= L, hasL = lane < L;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  if (wv > 0) {
    if (bad) return;                     // (every wave reaches the same verdict from the same LDS contents)
    // (nxtl is written by the sweeping wave below; read here only behind `go`)
    while (__hip_atomic_load(prog + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    float *gam = gamw + (wv - 1) * 64, *csum_w = csumw + (wv - 1) * C;
    for (int c = lane; c < C; c += 64) csum_w[c] = 0.f;
    for (int i = wv - 1; i < Tb; i += 3) {
      const int t = Tb - 1 - i;
      while (__hip_atomic_load(prog, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= i) __builtin_amdgcn_s_sleep(1);
      const float z = lse[t];
      const float gB = hasB ? as[(size_t)t * Smax + 2 * lane] : 0.f;
      const float gL = hasL ? as[(size_t)t * Smax + 2 * lane + 1] : 0.f;
      const float sumB = wave_sum_dpp(gB);
      gam[lane] = gL;
      WSYNC();
      if (first) {
        float v = gL;
        for (int j = nxt; j >= 0; j = nxtl[j]) v += gam[j];
        csum_w[lab] = v;
      }
      if (lane == 0) csum_w[blank] = sumB;
      WSYNC();
      for (int c = lane; c < C; c += 64)
        dl[(size_t)t * C + c] = p.scale * (__builtin_amdgcn_exp2f((xs[(size_t)t * C + c] - z) * LOG2E) - csum_w[c]);
      WSYNC();
    }
    return;
  }
  if (bad) {
    for (int i = lane; i < Tb * C; i += 64) dl[i] = 0.f;
    if (lane == 0) {
      p.nll[b] = __builtin_inff();
      atomicCAS(p.status, 0, b + 1);
    }
    return;
  }
  nxtl[lane] = nxt;
  WSYNC();
  const bool skipA = lane >= 1 && lane < L && lab != lab_up;   // 2i-1 -> 2i+1
  const bool skipB = lane + 1 < L && lab_dn != lab;            // 2i+1 -> 2i+3
  const int labc = hasL ? lab : blank;                          // a valid column for idle lanes

  // ---- alpha sweep --------------------------------------------------------
  float aB = CTC_NEG, aL = CTC_NEG;
  float yB = (xs[blank] - lse[0]) * LOG2E, yL = (xs[labc] - lse[0]) * LOG2E;     // base-2 from here on
  for (int t = 0; t < Tb; ++t) {
    // next frame's emission log-probabilities: independent of the recursion, issued first
    const int tn = t + 1 < Tb ? t + 1 : t;
    const float zn = lse[tn], xBn = xs[(size_t)tn * C + blank], xLn = xs[(size_t)tn * C + labc];
    if (t == 0) {
      aB = lane == 0 ? yB : CTC_NEG;
      aL = (lane == 0 && hasL) ? yL : CTC_NEG;
    } else {
      const float up = dpp_up(aL, CTC_NEG);
      const float nB = lse2_fast(aB, up) + yB;
      const float nL = lse3_fast(aL, aB, skipA ? up : CTC_NEG) + yL;
      aB = hasB ? nB : CTC_NEG;
      aL = hasL ? nL : CTC_NEG;
    }
    if (hasB) as[(size_t)t * Smax + 2 * lane] = aB;
    if (hasL) as[(size_t)t * Smax + 2 * lane + 1] = aL;
    yB = (xBn - zn) * LOG2E;
    yL = (xLn - zn) * LOG2E;
  }
  const float ll = L >= 1 ? lse2_fast(__shfl(aB, L), __shfl(aL, L - 1)) : __shfl(aB, 0);
  if (lane == 0) p.nll[b] = -ll * LN2;
  WSYNC();
  if (lane == 0) __hip_atomic_store(prog + 1, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);

  // ---- beta sweep: the frame's occupations replace alpha[t], the workers take it from there ----------
  float bB = CTC_NEG, bL = CTC_NEG;
  for (int t = Tb - 1; t >= 0; --t) {
    const float z = lse[t];
    yB = (xs[(size_t)t * C + blank] - z) * LOG2E;
    yL = (xs[(size_t)t * C + labc] - z) * LOG2E;
    const float aB0 = hasB ? as[(size_t)t * Smax + 2 * lane] : 0.f;
    const float aL0 = hasL ? as[(size_t)t * Smax + 2 * lane + 1] : 0.f;
    if (t == Tb - 1) {
      bB = lane == L ? yB : CTC_NEG;
      bL = (hasL && lane == L - 1) ? yL : CTC_NEG;
    } else {
      const float dnB = dpp_dn(bB, CTC_NEG), dnL = dpp_dn(bL, CTC_NEG);
      const float nB = lse2_fast(bB, hasL ? bL : CTC_NEG) + yB;
      const float nL = lse3_fast(bL, dnB, skipB ? dnL : CTC_NEG) + yL;
      bB = hasB ? nB : CTC_NEG;
      bL = hasL ? nL : CTC_NEG;
    }
    if (hasB) as[(size_t)t * Smax + 2 * lane] = __builtin_amdgcn_exp2f((aB0 - ll) + (bB - yB));
    if (hasL) as[(size_t)t * Smax + 2 * lane + 1] = __builtin_amdgcn_exp2f((aL0 - ll) + (bL - yL));
    WSYNC();
    if (lane == 0) __hip_atomic_store(prog, Tb - t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

#undef WSYNC
static size_t ctc_wave_lds_bytes(int T, int C, int Smax) {
  return ((size_t)T * C + T + (size_t)T * Smax + C + 64 + 2 + 3 * 64 + 3 * (size_t)C) * sizeof(float) + (128 + 2) * sizeof(int);
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
  hipStream_t s = static_cast<hipStream_t>(stream);
  CtcArgs p;
  p.B = B; p.T = T; p.C = C; p.Lmax = Lmax; p.Smax = Smax;
  p.logits = logits; p.logit_len = logit_len; p.labels = labels; p.label_len = label_len;
  p.scale = grad_scale; p.nll = nll; p.dlogits = dlogits; p.status = status;
  p.alpha = static_cast<float *>(ws);
  NABU_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), s));
  static int force_wg = -1;            // NABU_CTC_WORKGROUP=1: the workgroup kernel for every shape (A/B tests)
  if (force_wg < 0) { const char *e = getenv("NABU_CTC_WORKGROUP"); force_wg = e ? atoi(e) : 0; }
  const size_t wshm = ctc_wave_lds_bytes(T, C, Smax);
  if (!force_wg && Lmax <= 63 && wshm <= 150 * 1024) {
    // wave-synchronous kernel: one wave64 per utterance
    auto kern = ctc_wave_kernel;
    if (wshm > 64 * 1024)
      NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wshm));
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), wshm, s, p);
    NABU_LAUNCH_CHECK();
    return 0;
  }
  const size_t shm = ctc_lds_bytes(T, C, Smax);
  if (shm > 150 * 1024) return fail(NABU_EUNSUP, "ctc: T=%d, Lmax=%d, C=%d need %zu B of LDS", T, Lmax, C, shm);
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ctc_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
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
