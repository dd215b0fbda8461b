// speller.hip — per-step kernels of the Speller decoder (RNNDecoder + tf LSTMCell +
// AttentionWrapper with Bahdanau / location-aware attention + projection wrapper).
//
// The dense products of a step ([ctx,h]·K, h·Wq, [h,ctx]·Wout and their gradients)
// run on nabu_gemm_f32; this file holds what is not a GEMM:
//   lstm_cell_fwd/bwd : gate nonlinearities, state update, one-hot input as a row
//                       gather of the kernel, finished-row freezing;
//   attn_fwd          : score + mask + softmax + context fused over the encoder
//                       length, one workgroup per utterance: keys/values rows are
//                       read once per step in 2-4 KiB coalesced rows, the location
//                       features are computed from the previous alignment in LDS;
//   attn_bwd          : the matching gradient (tanh recomputed, not stored).
// Layouts: keys [B,Te,U], values [B,Te,E] batch-major; per-step state time-major.
#include "common.h"
#include "gemm_args.h"
#include "speller_persist.h"

#include <stdlib.h>

#include <string>
#include <thread>

namespace nabu {

typedef float f32x4_ __attribute__((ext_vector_type(4)));
constexpr int AT = 512;   // threads per attention workgroup (8 wave64)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------
// LSTM cell (tf.contrib.rnn.LSTMCell: gate order i,j,f,o, forget_bias 1)
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(
    int B, int U, int step, const int32_t *__restrict__ seq_len, const float *__restrict__ z,
    const float *__restrict__ bias, const float *__restrict__ emb, const int32_t *__restrict__ ids,
    const float *__restrict__ c_prev, const float *__restrict__ h_prev, float *__restrict__ acts,
    float *__restrict__ c_new, float *__restrict__ h_new) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * U) return;
  const int b = idx / U, u = idx % U;
  const size_t zo = (size_t)b * 4 * U + u;
  if (step >= seq_len[b]) {   // finished row: dynamic_decode(impute_finished) freezes the state
    c_new[idx] = c_prev[idx];
    h_new[idx] = h_prev[idx];
    acts[zo] = acts[zo + U] = acts[zo + 2 * U] = acts[zo + 3 * U] = 0.f;
    return;
  }
  float zi = z[zo] + bias[u], zj = z[zo + U] + bias[U + u];
  float zf = z[zo + 2 * U] + bias[2 * U + u], zq = z[zo + 3 * U] + bias[3 * U + u];
  if (emb) {   // one-hot input times kernel == one row of the kernel
    const float *e = emb + (size_t)ids[b] * 4 * U + u;
    zi += e[0]; zj += e[U]; zf += e[2 * U]; zq += e[3 * U];
  }
  const float i = sigmoidf_(zi), g = tanhf_(zj), f = sigmoidf_(zf + 1.0f), o = sigmoidf_(zq);
  const float c = c_prev[idx] * f + i * g;
  acts[zo] = i; acts[zo + U] = g; acts[zo + 2 * U] = f; acts[zo + 3 * U] = o;
  c_new[idx] = c;
  h_new[idx] = tanhf_(c) * o;
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(
    int B, int U, int step, const int32_t *__restrict__ seq_len, const float *__restrict__ acts,
    const float *__restrict__ c_new, const float *__restrict__ c_prev, const float *__restrict__ dh,
    const float *__restrict__ dh2, const float *__restrict__ dc_in, float *__restrict__ dz,
    float *__restrict__ dc_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * U) return;
  const int b = idx / U, u = idx % U;
  const size_t zo = (size_t)b * 4 * U + u;
  if (step >= seq_len[b]) {
    dz[zo] = dz[zo + U] = dz[zo + 2 * U] = dz[zo + 3 * U] = 0.f;
    dc_out[idx] = dc_in[idx];
    return;
  }
  const float i = acts[zo], g = acts[zo + U], f = acts[zo + 2 * U], o = acts[zo + 3 * U];
  const float tc = tanhf_(c_new[idx]);
  const float dht = dh[idx] + (dh2 ? dh2[idx] : 0.f);
  const float dct = dc_in[idx] + dht * o * (1.f - tc * tc);
  dz[zo] = dct * g * i * (1.f - i);
  dz[zo + U] = dct * i * (1.f - g * g);
  dz[zo + 2 * U] = dct * c_prev[idx] * f * (1.f - f);
  dz[zo + 3 * U] = dht * tc * o * (1.f - o);
  dc_out[idx] = dct * f;
}

// ---------------------------------------------------------------------------
struct AttnArgs {
  int B, Te, E, U, kind, K, F, step, prob_fn;
  float *znorm;      // [B] normaliser of normalized_sigmoid (forward writes, backward reads)
  const int32_t *dec_len, *enc_len;
  const float *keys, *values, *q, *v, *ck, *wf, *align_prev;
  // forward
  const float *ctx_prev;
  float *align, *ctx;
  // backward
  const float *dctx, *dalign_in;
  float *dq, *dkeys, *dv_part, *dwf_part, *dck_part, *dalign_out;
  float *dq_part, *dcf_g;   // [B,S,U] per-slice dq, [B,Te,F] d location features (backward scratch)
  float *fwd_part;          // [B,S,E+4] per-slice context + (local max, local sum) (forward scratch)
  float *ds_out, *cf_out;   // DEFER: d score [B,Te] and location features [B,Te,F] of this step, for attn_param_grads_kernel
  unsigned *tickets;        // [B] zeroed counters: the slice that arrives last finishes its utterance inside
                            // the launch (no finish kernel); nullptr = separate finish kernel
};

// store / load of data handed from one workgroup to another inside a launch: write-through store, L1-bypassing
// load (MI355X_MICROARCH.md "sc1 stores AND sc1 loads"); plain when the hand-off is a kernel boundary
__device__ __forceinline__ void xst(float *p, float v, bool x) {
  if (x) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else   *p = v;
}
__device__ __forceinline__ float xld(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave has drained its stores; returns true in the workgroup that arrives last of n
__device__ __forceinline__ bool last_arriver(unsigned *ticket, unsigned n, int *flag) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == n - 1;
    if (old == n - 1) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for reuse
  }
  __syncthreads();
  return *flag != 0;
}

// LDS carve for the attention kernels (floats): al[Te] prev alignment (padded conv input),
// sc[Te] scores / alignments, cf[Te*F] location features, red[...] reductions
// the conv kernel [K*F] is staged at the start of the dynamic LDS (ck_floats, 16-byte multiple)
__device__ __forceinline__ int ck_floats(const AttnArgs &p) { return p.kind == 1 ? (p.K * p.F + 3) & ~3 : 0; }
__device__ __forceinline__ void conv_features(const AttnArgs &p, const float *al_prev, float *cf, int lo, int hi,
                                              const float *ck_s) {
  // cf[t,f] = sum_d a[t + d - pb] ck[d,f], 'same' padding, pb = (K-1)/2 (tf.layers.conv1d); frames [lo,hi)
  const int pb = (p.K - 1) / 2;
  for (int i = lo * p.F + threadIdx.x; i < hi * p.F; i += AT) {
    const int t = i / p.F, f = i % p.F;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // independent chains: the loop is LDS-latency bound
    const int d0 = max(0, pb - t), d1 = min(p.K, p.Te + pb - t);
    const float *a = al_prev + t - pb, *c = ck_s + f;
    int d = d0;
    for (; d + 3 < d1; d += 4) {
      s0 = fmaf(a[d], c[d * p.F], s0);
      s1 = fmaf(a[d + 1], c[(d + 1) * p.F], s1);
      s2 = fmaf(a[d + 2], c[(d + 2) * p.F], s2);
      s3 = fmaf(a[d + 3], c[(d + 3) * p.F], s3);
    }
    for (; d < d1; ++d) s0 = fmaf(a[d], c[d * p.F], s0);
    cf[i] = (s0 + s1) + (s2 + s3);
  }
}

// REG = location-aware attention with U <= 512 and F <= 12: the Dense weights of the location
// features (conv_proj) stay in registers for the whole kernel (forward), and their gradient is
// accumulated in registers instead of a second pass over all frames (backward).
constexpr int RJ = 2, RF = 12;
template <int MODE>   // 0 vanilla, 1 location-aware (generic), 2 location-aware (REG)
__global__ __launch_bounds__(AT) void attn_fwd_kernel(AttnArgs p) {
  constexpr bool REG = MODE == 2, KIND = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // grid (B, S).  S = 1: the workgroup does the whole utterance.  S > 1 (small batches: fill the chip):
  // workgroup (b, sl) scores the frames [lo, n) only and leaves exp(score - local max), the local
  // max / sum and its part of the context; attn_fwd_finish_kernel rescales and combines them.
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Te = p.Te, U = p.U, E = p.E;
  float *base = smem + ck_floats(p);
  float *alp = base, *sc = alp + Te, *cf = sc + Te, *red = cf + (KIND ? Te * p.F : 0);
  float *align = p.align + (size_t)b * Te;
  float *ctx = S > 1 ? p.fwd_part + ((size_t)b * S + sl) * (E + 4) : p.ctx + (size_t)b * E;
  const bool fused = S > 1 && p.tickets != nullptr;   // the last slice to arrive finishes the utterance
  __shared__ int last_flag;
  if (p.step >= p.dec_len[b]) {   // finished row: state frozen
    if (S > 1 && !(fused && sl == 0)) return;         // ... by the finish kernel / by slice 0
    float *cx = p.ctx + (size_t)b * E;
    for (int t = tid; t < Te; t += AT) align[t] = p.align_prev[(size_t)b * Te + t];
    for (int e = tid; e < E; e += AT) cx[e] = p.ctx_prev[(size_t)b * E + e];
    return;
  }
  const int nfull = min(max(p.enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, nfull), n = min(lo + per, nfull);   // my frames [lo, n)
  const float *keys = p.keys + (size_t)b * Te * U;
  const float *vals = p.values + (size_t)b * Te * E;
  const float *q = p.q + (size_t)b * U;
  if (KIND) {
    for (int t = tid; t < Te; t += AT) alp[t] = p.align_prev[(size_t)b * Te + t];
    for (int i = tid; i < p.K * p.F; i += AT) smem[i] = p.ck[i];
    __syncthreads();
    conv_features(p, alp, cf, lo, n, smem);
    __syncthreads();
  }
  // WindowedAttention (attention.py:294-396): only frames in [m - left - 1, m + right) may be attended,
  // m = first frame at which the cumulated previous alignment exceeds 0.5 (Te if it never does)
  int w_lo = 0, w_hi = Te;
  if (!KIND && p.kind == 2) {
    if (w == 0) {
      float carry = 0.f;
      int m = Te;
      for (int base = 0; base < Te && m == Te; base += 64) {
        const int t = base + lane;
        float c = t < Te ? p.align_prev[(size_t)b * Te + t] : 0.f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float up = __shfl_up(c, o);
          if (lane >= o) c += up;
        }
        c += carry;
        const unsigned long long hit = __ballot(t < Te && c > 0.5f);
        if (hit) m = base + __ffsll((long long)hit) - 1;
        carry = __shfl(c, 63);
      }
      if (lane == 0) { red[0] = __int_as_float(m); }
    }
    __syncthreads();
    const int m = __float_as_int(red[0]);
    w_lo = max(m - p.K - 1, 0);
    w_hi = min(m + p.F, Te);
    __syncthreads();
  }
  // scores: waves over encoder frames (4 frames of a wave in flight), lanes over 16-byte groups of
  // units — the keys are streamed once with coalesced 1 KiB wave loads
  {
    const int U4 = U / 4;
    constexpr int NW = AT / 64, FR = 4;
    const float4 *keys4 = reinterpret_cast<const float4 *>(keys);
    const float4 *q4 = reinterpret_cast<const float4 *>(q), *v4 = reinterpret_cast<const float4 *>(p.v);
    float4 wfr[REG ? RF : 1][REG ? RJ : 1];
    if (REG) {
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
          const int u4 = lane + 64 * j;
          wfr[f][j] = (f < p.F && u4 < U4) ? *reinterpret_cast<const float4 *>(p.wf + (size_t)f * U + 4 * u4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int t0 = lo + w; t0 < n; t0 += FR * NW) {
      float s[FR];
#pragma unroll
      for (int i = 0; i < FR; ++i) s[i] = 0.f;
#pragma unroll
      for (int j = 0; j < (REG ? RJ : 4); ++j) {
        const int u4 = lane + 64 * j;
        if (u4 >= U4) continue;
        const float4 qq = q4[u4], vv = v4[u4];
        float4 kx[FR];
#pragma unroll
        for (int i = 0; i < FR; ++i) {
          const int t = min(t0 + i * NW, n - 1);
          kx[i] = keys4[(size_t)t * U4 + u4];
        }
#pragma unroll
        for (int i = 0; i < FR; ++i) {
          const int t = t0 + i * NW;
          float4 x = make_float4(kx[i].x + qq.x, kx[i].y + qq.y, kx[i].z + qq.z, kx[i].w + qq.w);
          if (REG) {
            if (t < n) {
#pragma unroll
              for (int f = 0; f < RF; ++f)
                if (f < p.F) {
                  const float c = cf[t * p.F + f];
                  const float4 wf = wfr[f][j];
                  x.x = fmaf(c, wf.x, x.x); x.y = fmaf(c, wf.y, x.y); x.z = fmaf(c, wf.z, x.z); x.w = fmaf(c, wf.w, x.w);
                }
            }
          } else if (KIND && t < n)
            for (int f = 0; f < p.F; ++f) {
              const float c = cf[t * p.F + f];
              const float4 wf = *reinterpret_cast<const float4 *>(p.wf + (size_t)f * U + 4 * u4);
              x.x = fmaf(c, wf.x, x.x); x.y = fmaf(c, wf.y, x.y); x.z = fmaf(c, wf.z, x.z); x.w = fmaf(c, wf.w, x.w);
            }
          s[i] = fmaf(vv.x, tanhf_(x.x), s[i]);
          s[i] = fmaf(vv.y, tanhf_(x.y), s[i]);
          s[i] = fmaf(vv.z, tanhf_(x.z), s[i]);
          s[i] = fmaf(vv.w, tanhf_(x.w), s[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        const float tot = wave_sum(s[i]);
        if (lane == 0 && t0 + i * NW < n) sc[t0 + i * NW] = tot;
      }
    }
  }
  __syncthreads();
  if (!KIND && p.kind == 2) {
    for (int t = lo + tid; t < n; t += AT)
      if (t < w_lo || t >= w_hi) sc[t] = -INFINITY;
    __syncthreads();
  }
  if (S > 1) {
    // partial result of this slice: e[t] = exp(score - local max) (softmax) or sigmoid(score), the
    // local max and sum; the context loop below then accumulates e[t] * values[t] over my frames
    float m = -3.0e38f;
    if (p.prob_fn == 0) {
      for (int t = lo + tid; t < n; t += AT) m = fmaxf(m, sc[t]);
      m = fmaxf(m, __shfl_xor(m, 32)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 8));
      m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 1));
      if (lane == 0) red[w] = m;
      __syncthreads();
      m = red[0];
      for (int i = 1; i < AT / 64; ++i) m = fmaxf(m, red[i]);
      __syncthreads();
    }
    float z = 0.f;
    for (int t = lo + tid; t < n; t += AT) {
      const float e = p.prob_fn == 0 ? expf(sc[t] - m) : 1.0f / (1.0f + expf(-sc[t]));
      sc[t] = e;
      xst(align + t, e, fused);
      z += e;
    }
    z = wave_sum(z);
    if (lane == 0) red[w] = z;
    __syncthreads();
    z = 0.f;
    for (int i = 0; i < AT / 64; ++i) z += red[i];
    if (tid == 0) { xst(ctx + E, m, fused); xst(ctx + E + 1, z, fused); }
    __syncthreads();
  } else
  if (p.prob_fn != 0) {
    // sigmoid / normalized_sigmoid (attention.py:9-13, 41-55): masked frames (score -inf) give 0
    float z = 0.f;
    for (int t = tid; t < n; t += AT) {
      const float e = 1.0f / (1.0f + expf(-sc[t]));
      sc[t] = e;
      z += e;
    }
    z = wave_sum(z);
    if (lane == 0) red[w] = z;
    __syncthreads();
    z = 0.f;
    for (int i = 0; i < AT / 64; ++i) z += red[i];
    const float inv = p.prob_fn == 2 ? 1.0f / z : 1.0f;
    if (p.prob_fn == 2 && tid == 0 && p.znorm) p.znorm[b] = z;
    for (int t = tid; t < Te; t += AT) {
      const float a = t < n ? sc[t] * inv : 0.f;
      sc[t] = a;
      align[t] = a;
    }
    __syncthreads();
  } else {
  // softmax over the valid frames (score_mask_value = -inf past the length)
  float m = -3.0e38f;
  for (int t = tid; t < n; t += AT) m = fmaxf(m, sc[t]);
  m = fmaxf(m, __shfl_xor(m, 32)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 8));
  m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 1));
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < AT / 64; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float z = 0.f;
  for (int t = tid; t < n; t += AT) {
    const float e = expf(sc[t] - m);
    sc[t] = e;
    z += e;
  }
  z = wave_sum(z);
  if (lane == 0) red[w] = z;
  __syncthreads();
  z = 0.f;
  for (int i = 0; i < AT / 64; ++i) z += red[i];
  const float inv = 1.0f / z;
  for (int t = tid; t < Te; t += AT) {
    const float a = t < n ? sc[t] * inv : 0.f;
    sc[t] = a;
    align[t] = a;
  }
  __syncthreads();
  }
  // context = alignments^T · values: threads over 16-byte column groups, the frames split over
  // the AT / (E/4) thread groups (8 frames of a thread in flight), partial sums through LDS
  {
    const int E4 = E / 4;
    const int nsp = max(1, min(AT / max(E4, 1), 8));    // frame partitions
    const float4 *vals4 = reinterpret_cast<const float4 *>(vals);
    // [nsp][AT / nsp] partial sums behind the scalars of `red`, 16-byte aligned
    float4 *part = reinterpret_cast<float4 *>(base + ((2 * Te + (KIND ? Te * p.F : 0) + 64 + 3) & ~3));
    for (int c0 = 0; c0 < E4; c0 += AT / nsp) {
      const int e4 = c0 + tid % (AT / nsp), pt = tid / (AT / nsp);
      if (e4 < E4 && pt < nsp) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        int t = lo + pt;
        for (; t + 3 * nsp < n; t += 4 * nsp) {
          const float4 a0 = vals4[(size_t)t * E4 + e4], a1 = vals4[(size_t)(t + nsp) * E4 + e4];
          const float4 a2 = vals4[(size_t)(t + 2 * nsp) * E4 + e4], a3 = vals4[(size_t)(t + 3 * nsp) * E4 + e4];
          const float w0 = sc[t], w1 = sc[t + nsp], w2 = sc[t + 2 * nsp], w3 = sc[t + 3 * nsp];
          c.x = fmaf(w0, a0.x, c.x); c.y = fmaf(w0, a0.y, c.y); c.z = fmaf(w0, a0.z, c.z); c.w = fmaf(w0, a0.w, c.w);
          c.x = fmaf(w1, a1.x, c.x); c.y = fmaf(w1, a1.y, c.y); c.z = fmaf(w1, a1.z, c.z); c.w = fmaf(w1, a1.w, c.w);
          c.x = fmaf(w2, a2.x, c.x); c.y = fmaf(w2, a2.y, c.y); c.z = fmaf(w2, a2.z, c.z); c.w = fmaf(w2, a2.w, c.w);
          c.x = fmaf(w3, a3.x, c.x); c.y = fmaf(w3, a3.y, c.y); c.z = fmaf(w3, a3.z, c.z); c.w = fmaf(w3, a3.w, c.w);
        }
        for (; t < n; t += nsp) {
          const float4 a0 = vals4[(size_t)t * E4 + e4];
          const float w0 = sc[t];
          c.x = fmaf(w0, a0.x, c.x); c.y = fmaf(w0, a0.y, c.y); c.z = fmaf(w0, a0.z, c.z); c.w = fmaf(w0, a0.w, c.w);
        }
        part[pt * (AT / nsp) + (e4 - c0)] = c;
      }
      __syncthreads();
      if (tid < AT / nsp && c0 + tid < E4) {
        float4 c = part[tid];
        for (int i = 1; i < nsp; ++i) {
          const float4 o = part[i * (AT / nsp) + tid];
          c.x += o.x; c.y += o.y; c.z += o.z; c.w += o.w;
        }
        if (fused) {
          float *o = ctx + 4 * (c0 + tid);
          xst(o, c.x, true); xst(o + 1, c.y, true); xst(o + 2, c.z, true); xst(o + 3, c.w, true);
        } else {
          reinterpret_cast<float4 *>(ctx)[c0 + tid] = c;
        }
      }
      __syncthreads();
    }
  }
  if (!fused) return;
  // ---- the slice that arrives last combines the utterance's slices (what attn_fwd_finish_kernel does)
  if (!last_arriver(p.tickets + b, (unsigned)S, &last_flag)) return;
  {
    float *fac = red;
    const float *part = p.fwd_part + (size_t)b * S * (E + 4);
    if (tid == 0) {
      float M = -3.0e38f, Z = 0.f;
      for (int i = 0; i < S; ++i) M = fmaxf(M, xld(part + (size_t)i * (E + 4) + E));
      for (int i = 0; i < S; ++i) {
        const float f = p.prob_fn == 0 ? expf(xld(part + (size_t)i * (E + 4) + E) - M) : 1.0f;
        fac[i] = f;
        Z += f * xld(part + (size_t)i * (E + 4) + E + 1);
      }
      const float inv = p.prob_fn == 1 ? 1.0f : 1.0f / Z;
      for (int i = 0; i < S; ++i) fac[i] *= inv;
      if (p.prob_fn == 2 && p.znorm) p.znorm[b] = Z;
    }
    __syncthreads();
    for (int t = tid; t < Te; t += AT) align[t] = t < nfull ? xld(align + t) * fac[t / per] : 0.f;
    float *cx = p.ctx + (size_t)b * E;
    for (int e = tid; e < E; e += AT) {
      float c = 0.f;
      for (int i = 0; i < S; ++i) c = fmaf(fac[i], xld(part + (size_t)i * (E + 4) + e), c);
      cx[e] = c;
    }
  }
}

// Location-aware attention forward of the step chain's sliced form (softmax, S > 1 slices of <= 32 frames, the last
// slice to arrive finishes the utterance; U <= 512, F <= 12, E <= 2 AT): attn_fwd_kernel<2> with everything that depends
// on nothing — previous alignment, conv kernel, the slice's values and keys, q, v, conv_proj — loaded at clamped addresses
// before the first wait, and the score's feature projection on the matrix pipe (exact fp32):
//   x^T[16 units x 16 frames] = conv_proj^T[16 x 12] . features^T[12 x 16] + (keys^T + q),  score[frame] = sum_u v[u] tanh(x)
// (lane = frame, register r = unit 4 kq + r of the tile, as in attn_bwd_loc_mfma_kernel).  21.9 -> see profiles.
template <int UT>
__global__ __launch_bounds__(AT) void attn_fwd_loc_mfma_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Te = p.Te, U = p.U, E = p.E, F = p.F;
  constexpr int NW = AT / 64;
  float *base = smem + ck_floats(p);
  float *alp = base, *sc = alp + Te, *cf = sc + Te, *red = cf + Te * F;
  float4 *part = reinterpret_cast<float4 *>(base + ((2 * Te + Te * F + 64 + 3) & ~3));      // [2][AT / 2] (also the waves' scores)
  float *align = p.align + (size_t)b * Te;
  float *ctx = p.fwd_part + ((size_t)b * S + sl) * (E + 4);
  __shared__ int last_flag;
  if (p.step >= p.dec_len[b]) {   // finished row: state frozen (by slice 0)
    if (sl != 0) return;
    float *cx = p.ctx + (size_t)b * E;
    for (int t = tid; t < Te; t += AT) align[t] = p.align_prev[(size_t)b * Te + t];
    for (int e = tid; e < E; e += AT) cx[e] = p.ctx_prev[(size_t)b * E + e];
    return;
  }
  const int nfull = min(max(p.enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, nfull), n = min(lo + per, nfull);   // my frames [lo, n)
  const float *keys = p.keys + (size_t)b * Te * U;
  const float *vals = p.values + (size_t)b * Te * E;
  const float *q = p.q + (size_t)b * U;
  const int fl = lane & 15, kq = lane >> 4;
  const int tl = max(n - 1, 0);
  typedef const f32x4_ *V4;
  const f32x4_ zv = {0.f, 0.f, 0.f, 0.f};
  // ---- loads that depend on nothing, in the order of use
  constexpr int NAL = (1024 + AT - 1) / AT;
  float alp_r[NAL], ck_r[8];
#pragma unroll
  for (int i = 0; i < NAL; ++i) alp_r[i] = p.align_prev[(size_t)b * Te + min(tid + AT * i, Te - 1)];
#pragma unroll
  for (int i = 0; i < 8; ++i) ck_r[i] = p.ck[min(tid + AT * i, p.K * F - 1)];
  f32x4_ kc[2][UT], qv[UT], v4[UT];
  float a1[UT][3];
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const int u0 = min(16 * (w + NW * j), U - 16);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
      kc[ft][j] = *reinterpret_cast<V4>(keys + (size_t)min(lo + 16 * ft + fl, tl) * U + u0 + 4 * kq);
    qv[j] = *reinterpret_cast<V4>(q + u0 + 4 * kq);
    v4[j] = *reinterpret_cast<V4>(p.v + u0 + 4 * kq);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a1[j][ks] = p.wf[(size_t)min(4 * ks + kq, F - 1) * U + u0 + fl];
  }
  // values of the context: thread (frame partition pt, 16-byte column e4), frames lo + pt, lo + pt + 2, ...
  const int E4 = E / 4, e4 = min(tid & (AT / 2 - 1), E4 - 1), pt = tid / (AT / 2);
  f32x4_ vv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) vv[i] = reinterpret_cast<V4>(vals + (size_t)min(lo + pt + 2 * i, tl) * E)[e4];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NAL; ++i)
    if (tid + AT * i < Te) alp[tid + AT * i] = alp_r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + AT * i < p.K * F) smem[tid + AT * i] = ck_r[i];
  for (int i = tid + 8 * AT; i < p.K * F; i += AT) smem[i] = p.ck[i];
  for (int t = tid + AT * NAL; t < Te; t += AT) alp[t] = p.align_prev[(size_t)b * Te + t];
  __syncthreads();
  {
    // location features of my frames, two threads per (frame, filter)
    const int pb = (p.K - 1) / 2, nout = (n - lo) * F, hf = tid & 1;
    for (int o = tid >> 1; o < nout; o += AT / 2) {
      const int i = lo * F + o, t = i / F, f = i % F;
      const int d0 = max(0, pb - t), d1 = min(p.K, Te + pb - t), mid = (d0 + d1 + 1) >> 1;
      const float *a = alp + t - pb, *c = smem + f;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int d = hf ? mid : d0;
      const int de = hf ? d1 : mid;
      for (; d + 3 < de; d += 4) {
        s0 = fmaf(a[d], c[d * F], s0);
        s1 = fmaf(a[d + 1], c[(d + 1) * F], s1);
        s2 = fmaf(a[d + 2], c[(d + 2) * F], s2);
        s3 = fmaf(a[d + 3], c[(d + 3) * F], s3);
      }
      for (; d < de; ++d) s0 = fmaf(a[d], c[d * F], s0);
      float sm = (s0 + s1) + (s2 + s3);
      sm += __shfl_xor(sm, 1);
      if (!hf) cf[i] = sm;
    }
  }
  __syncthreads();
  // ---- scores
  {
    float b1[2][3];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      const int t = lo + 16 * ft + fl;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) b1[ft][ks] = (t < n && 4 * ks + kq < F) ? cf[min(t, Te - 1) * F + min(4 * ks + kq, F - 1)] : 0.f;
    }
    float sacc[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < UT; ++j) {
      if (16 * (w + NW * j) >= U) break;
#pragma unroll
      for (int ft = 0; ft < 2; ++ft) {
        f32x4_ x = kc[ft][j] + qv[j];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) x = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j][ks], b1[ft][ks], x, 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c)      // one exp + one rcp per tanh (as the persistent kernel's score; relative error ~1e-7)
          sacc[ft] = fmaf(v4[j][c], 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[c])) - 1.0f, sacc[ft]);
      }
    }
    float *wsc = reinterpret_cast<float *>(part);          // [NW][32]
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      float v = sacc[ft];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (kq == 0) wsc[w * 32 + 16 * ft + fl] = v;
    }
    __syncthreads();
    if (tid < 32 && lo + tid < n) {
      float tot = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) tot += wsc[i * 32 + tid];
      sc[lo + tid] = tot;
    }
    __syncthreads();
  }
  // ---- partial result of this slice: e[t] = exp(score - local max), the local max and sum
  float m = -3.0e38f;
  for (int t = lo + tid; t < n; t += AT) m = fmaxf(m, sc[t]);
  m = fmaxf(m, __shfl_xor(m, 32)); m = fmaxf(m, __shfl_xor(m, 16)); m = fmaxf(m, __shfl_xor(m, 8));
  m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 1));
  if (lane == 0) red[w] = m;
  __syncthreads();
  m = red[0];
  for (int i = 1; i < NW; ++i) m = fmaxf(m, red[i]);
  __syncthreads();
  float z = 0.f;
  for (int t = lo + tid; t < n; t += AT) {
    const float e = expf(sc[t] - m);
    sc[t] = e;
    xst(align + t, e, true);
    z += e;
  }
  z = wave_sum(z);
  if (lane == 0) red[w] = z;
  __syncthreads();
  z = 0.f;
  for (int i = 0; i < NW; ++i) z += red[i];
  if (tid == 0) { xst(ctx + E, m, true); xst(ctx + E + 1, z, true); }
  // ---- partial context = e^T . values over my frames
  {
    f32x4_ c = zv;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int t = lo + pt + 2 * i;
      const float wgt = t < n ? sc[t] : 0.f;
      c += wgt * vv[i];
    }
    part[tid] = make_float4(c[0], c[1], c[2], c[3]);
    __syncthreads();
    if (tid < AT / 2 && tid < E4) {
      const float4 c0 = part[tid], c1 = part[AT / 2 + tid];
      float *o = ctx + 4 * tid;
      xst(o, c0.x + c1.x, true); xst(o + 1, c0.y + c1.y, true); xst(o + 2, c0.z + c1.z, true); xst(o + 3, c0.w + c1.w, true);
    }
  }
  // ---- the slice that arrives last combines the utterance's slices (attn_fwd_finish_kernel's work)
  if (!last_arriver(p.tickets + b, (unsigned)S, &last_flag)) return;
  {
    float *fac = red;
    const float *pr = p.fwd_part + (size_t)b * S * (E + 4);
    if (tid == 0) {
      float M = -3.0e38f, Z = 0.f;
      for (int i = 0; i < S; ++i) M = fmaxf(M, xld(pr + (size_t)i * (E + 4) + E));
      for (int i = 0; i < S; ++i) {
        const float f = expf(xld(pr + (size_t)i * (E + 4) + E) - M);
        fac[i] = f;
        Z += f * xld(pr + (size_t)i * (E + 4) + E + 1);
      }
      const float inv = 1.0f / Z;
      for (int i = 0; i < S; ++i) fac[i] *= inv;
    }
    __syncthreads();
    for (int t = tid; t < Te; t += AT) align[t] = t < nfull ? xld(align + t) * fac[t / per] : 0.f;
    float *cx = p.ctx + (size_t)b * E;
    for (int e = tid; e < E; e += AT) {
      float pv[8];                        // S <= 8 (attn_bwd_nslices): every slice's load in flight before the first fma
#pragma unroll
      for (int i = 0; i < 8; ++i) pv[i] = xld(pr + (size_t)min(i, S - 1) * (E + 4) + e);
      float c = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) c = i < S ? fmaf(fac[i], pv[i], c) : c;
      cx[e] = c;
    }
  }
}

// Combines the slices of a sliced forward pass (flash-style): M = max of the local maxima, every
// slice's weights and partial context are rescaled by exp(m_s - M) / Z.  grid B, 256 threads.
__global__ __launch_bounds__(256) void attn_fwd_finish_kernel(AttnArgs p, int S) {
  __shared__ float fac[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Te = p.Te, E = p.E;
  float *align = p.align + (size_t)b * Te;
  float *ctx = p.ctx + (size_t)b * E;
  if (p.step >= p.dec_len[b]) {   // finished row: state frozen
    for (int t = tid; t < Te; t += 256) align[t] = p.align_prev[(size_t)b * Te + t];
    for (int e = tid; e < E; e += 256) ctx[e] = p.ctx_prev[(size_t)b * E + e];
    return;
  }
  const int n = min(max(p.enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S;
  const float *part = p.fwd_part + (size_t)b * S * (E + 4);
  if (tid == 0) {
    float M = -3.0e38f, Z = 0.f;
    for (int i = 0; i < S; ++i) M = fmaxf(M, part[(size_t)i * (E + 4) + E]);
    for (int i = 0; i < S; ++i) {
      const float sc = p.prob_fn == 0 ? expf(part[(size_t)i * (E + 4) + E] - M) : 1.0f;
      fac[i] = sc;
      Z += sc * part[(size_t)i * (E + 4) + E + 1];
    }
    const float inv = p.prob_fn == 1 ? 1.0f : 1.0f / Z;
    for (int i = 0; i < S; ++i) fac[i] *= inv;
    if (p.prob_fn == 2 && p.znorm) p.znorm[b] = Z;
  }
  __syncthreads();
  for (int t = tid; t < Te; t += 256) align[t] = t < n ? align[t] * fac[t / per] : 0.f;
  for (int e = tid; e < E; e += 256) {
    float c = 0.f;
    for (int i = 0; i < S; ++i) c = fmaf(fac[i], part[(size_t)i * (E + 4) + e], c);
    ctx[e] = c;
  }
}

// DEFER: the sums over decoder steps that nobody in the step chain waits for — d keys, d attention_v, d conv_proj —
// are left to attn_param_grads_kernel (one launch after the chain, all steps in parallel); this kernel then only saves
// d score (and the location features) of the step and is rid of the dkeys read-modify-write, of 96 accumulation
// registers and of a dozen cross-wave reductions
template <int MODE, bool DEFER>
__global__ __launch_bounds__(AT) void attn_bwd_kernel(AttnArgs p) {
  constexpr bool REG = MODE == 2, KIND = MODE != 0;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // grid (B, S): workgroup (b, s) owns the encoder frames [lo, hi) of utterance b — the per-utterance
  // sums over frames (dq, dv, d conv_proj) leave as per-slice partials, the location features' gradient
  // goes to HBM, and attn_bwd_finish_kernel does what needs all frames of an utterance
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Te = p.Te, U = p.U, E = p.E, F = p.F;
  constexpr int NW = AT / 64;
  float *base = smem + ck_floats(p);
  float *alp = base;                       // [Te] previous alignment
  float *ds = alp + Te;                    // [Te] d score
  float *cf = ds + Te;                     // [Te*F]
  // [NW * U] cross-wave partials (dq / dv), also scalars; 16-byte aligned
  float *red = base + ((2 * Te + (KIND ? Te * F : 0) + 3) & ~3);
  float *dq = p.dq_part + ((size_t)b * S + sl) * U;
  // vanilla / windowed attention: the slice that arrives last sums the utterance's dq partials inside the launch
  const bool fused = !KIND && p.tickets != nullptr;
  __shared__ int last_flag;
  if (p.step >= p.dec_len[b]) {            // finished row: the finish kernel (fused: slice 0) writes its zeros
    if (fused && sl == 0) {
      for (int u = tid; u < U; u += AT) p.dq[(size_t)b * U + u] = 0.f;
      if (p.dalign_out)
        for (int t = tid; t < Te; t += AT)
          p.dalign_out[(size_t)b * Te + t] = p.dalign_in ? p.dalign_in[(size_t)b * Te + t] : 0.f;
    }
    return;
  }
  const int n = min(max(p.enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, n), hi = min(lo + per, n);
  float *dcf = p.dcf_g + (size_t)b * Te * F;   // [Te*F] in HBM
  const float *keys = p.keys + (size_t)b * Te * U;
  const float *vals = p.values + (size_t)b * Te * E;
  const float *q = p.q + (size_t)b * U;
  const float *al = p.align + (size_t)b * Te;       // this step's alignments
  const float *dctx = p.dctx + (size_t)b * E;
  float *dkeys = p.dkeys + (size_t)b * Te * U;
  if (KIND) {
    for (int t = tid; t < Te; t += AT) alp[t] = p.align_prev[(size_t)b * Te + t];
    for (int i = tid; i < p.K * F; i += AT) smem[i] = p.ck[i];
    __syncthreads();
    conv_features(p, alp, cf, lo, hi, smem);
  }
  // d alignment[t] = dctx · values[t] (+ the gradient arriving through next step's location features):
  // waves over frames (4 in flight), lanes over 16-byte groups of the encoder dimension
  {
    const int E4 = E / 4;
    constexpr int FR = 4;
    const float4 *vals4 = reinterpret_cast<const float4 *>(vals);
    const float4 *dctx4 = reinterpret_cast<const float4 *>(dctx);
    for (int t0 = lo + w; t0 < hi; t0 += FR * NW) {
      float s[FR];
#pragma unroll
      for (int i = 0; i < FR; ++i) s[i] = 0.f;
      for (int e4 = lane; e4 < E4; e4 += 64) {
        const float4 dc = dctx4[e4];
        float4 vv[FR];
#pragma unroll
        for (int i = 0; i < FR; ++i) vv[i] = vals4[(size_t)min(t0 + i * NW, hi - 1) * E4 + e4];
#pragma unroll
        for (int i = 0; i < FR; ++i) {
          s[i] = fmaf(dc.x, vv[i].x, s[i]); s[i] = fmaf(dc.y, vv[i].y, s[i]);
          s[i] = fmaf(dc.z, vv[i].z, s[i]); s[i] = fmaf(dc.w, vv[i].w, s[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        const int t = t0 + i * NW;
        const float tot = wave_sum(s[i]);
        if (lane == 0 && t < hi) ds[t] = tot + (p.dalign_in ? p.dalign_in[(size_t)b * Te + t] : 0.f);
      }
    }
  }
  __syncthreads();
  // softmax backward: dscore = a * (da - sum a*da)
  // sum_t a[t] da[t] over ALL frames of the utterance without visiting them:
  //   da[t] = dctx·values[t] + dalign_in[t]  =>  sum_t a[t] da[t] = dctx·context + sum_t a[t] dalign_in[t]
  float r = 0.f;
  {
    const float *cx = p.ctx + (size_t)b * E;
    for (int e = tid; e < E; e += AT) r = fmaf(dctx[e], cx[e], r);
    if (p.dalign_in)
      for (int t = tid; t < n; t += AT) r = fmaf(al[t], p.dalign_in[(size_t)b * Te + t], r);
  }
  r = wave_sum(r);
  if (lane == 0) red[w] = r;
  __syncthreads();
  r = 0.f;
  for (int i = 0; i < NW; ++i) r += red[i];
  __syncthreads();
  if (p.prob_fn == 0) {
    for (int t = lo + tid; t < hi; t += AT) ds[t] = al[t] * (ds[t] - r);
  } else if (p.prob_fn == 1) {        // a = sigmoid(s): ds = da a (1 - a)
    for (int t = lo + tid; t < hi; t += AT) ds[t] = ds[t] * al[t] * (1.f - al[t]);
  } else {                            // a = sg / z, sg = sigmoid(s): ds = (da - sum a da) a (1 - a z)
    const float z = p.znorm[b];
    for (int t = lo + tid; t < hi; t += AT) ds[t] = (ds[t] - r) * al[t] * (1.f - al[t] * z);
  }
  __syncthreads();
  if (DEFER) {
    for (int t = lo + tid; t < hi; t += AT) p.ds_out[(size_t)b * Te + t] = ds[t];
    if (KIND)
      for (int i = lo * F + tid; i < hi * F; i += AT) p.cf_out[(size_t)b * Te * F + i] = cf[i];
  }
  // through v·tanh(keys + q + f): lanes own 16-byte groups of units (u4 = lane + 64 j), waves split
  // the frames (2 frames of a wave in flight); keys are read and dkeys updated with 1 KiB wave accesses
  constexpr int MAXJ = REG ? RJ : 4;       // U <= 1024 (REG: U <= 512)
  constexpr int NF = REG ? RF : 16;
  const int U4 = U / 4;
  float4 dq_l[MAXJ], dv_l[MAXJ];
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) dq_l[j] = dv_l[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 dwf_l[REG ? RF : 1][REG ? RJ : 1];   // d conv_proj[f, my units], summed over my frames
  if (REG) {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int j = 0; j < RJ; ++j) dwf_l[f][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // DEFER: those 96 registers hold the projection rows instead (every frame needs them twice: for x and for d features)
  constexpr bool WREG = REG && DEFER;
  float4 wfr[WREG ? RF : 1][WREG ? RJ : 1];
  if (WREG) {
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        const int u4 = lane + 64 * j;
        wfr[f][j] = (f < F && u4 < U / 4) ? *reinterpret_cast<const float4 *>(p.wf + (size_t)f * U + 4 * u4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  {
    const float4 *keys4 = reinterpret_cast<const float4 *>(keys);
    float4 *dkeys4 = reinterpret_cast<float4 *>(dkeys);
    const float4 *q4 = reinterpret_cast<const float4 *>(q), *v4 = reinterpret_cast<const float4 *>(p.v);
    constexpr int FR = (MODE == 1 || (KIND && !DEFER)) ? 1 : 2;         // frames of a wave in flight (registers!)
    for (int t0 = lo + w; t0 < hi; t0 += FR * NW) {
      float dcf_l[FR][NF];
#pragma unroll
      for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) dcf_l[i][f] = 0.f;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int u4 = lane + 64 * j;
        if (u4 < U4) {
          const float4 qq = q4[u4], vv = v4[u4];
          float4 kx[FR], dk[FR];
#pragma unroll
          for (int i = 0; i < FR; ++i) {
            const size_t o = (size_t)min(t0 + i * NW, hi - 1) * U4 + u4;
            kx[i] = keys4[o];
            if (!DEFER) dk[i] = dkeys4[o];
          }
#pragma unroll
          for (int i = 0; i < FR; ++i) {
            const int t = t0 + i * NW;
            if (t < hi) {
              const float g = ds[t];
              float x[4] = {kx[i].x + qq.x, kx[i].y + qq.y, kx[i].z + qq.z, kx[i].w + qq.w};
              if (WREG) {
#pragma unroll
                for (int f = 0; f < RF; ++f)
                  if (f < F) {
                    const float c = cf[t * F + f];
                    const float4 wf = wfr[f][j];
                    x[0] = fmaf(c, wf.x, x[0]); x[1] = fmaf(c, wf.y, x[1]); x[2] = fmaf(c, wf.z, x[2]); x[3] = fmaf(c, wf.w, x[3]);
                  }
              } else if (KIND)
                for (int f = 0; f < F; ++f) {
                  const float c = cf[t * F + f];
                  const float4 wf = *reinterpret_cast<const float4 *>(p.wf + (size_t)f * U + 4 * u4);
                  x[0] = fmaf(c, wf.x, x[0]); x[1] = fmaf(c, wf.y, x[1]); x[2] = fmaf(c, wf.z, x[2]); x[3] = fmaf(c, wf.w, x[3]);
                }
              const float vvv[4] = {vv.x, vv.y, vv.z, vv.w};
              float d[4], th[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                th[c] = tanhf_(x[c]);
                d[c] = g * vvv[c] * (1.f - th[c] * th[c]);
              }
              dq_l[j].x += d[0]; dq_l[j].y += d[1]; dq_l[j].z += d[2]; dq_l[j].w += d[3];
              if (!DEFER) {
                dv_l[j].x = fmaf(g, th[0], dv_l[j].x); dv_l[j].y = fmaf(g, th[1], dv_l[j].y);
                dv_l[j].z = fmaf(g, th[2], dv_l[j].z); dv_l[j].w = fmaf(g, th[3], dv_l[j].w);
                dkeys4[(size_t)t * U4 + u4] = make_float4(dk[i].x + d[0], dk[i].y + d[1], dk[i].z + d[2], dk[i].w + d[3]);
              }
              if (KIND) {
#pragma unroll
                for (int f = 0; f < (REG ? RF : 16); ++f)
                  if (f < F) {
                    const float4 wf = WREG ? wfr[WREG ? f : 0][WREG ? j : 0] : *reinterpret_cast<const float4 *>(p.wf + (size_t)f * U + 4 * u4);
                    dcf_l[i][f] = fmaf(d[0], wf.x, fmaf(d[1], wf.y, fmaf(d[2], wf.z, fmaf(d[3], wf.w, dcf_l[i][f]))));
                    if (REG && !DEFER) {
                      const float c = cf[t * F + f];
                      dwf_l[f][j].x = fmaf(c, d[0], dwf_l[f][j].x); dwf_l[f][j].y = fmaf(c, d[1], dwf_l[f][j].y);
                      dwf_l[f][j].z = fmaf(c, d[2], dwf_l[f][j].z); dwf_l[f][j].w = fmaf(c, d[3], dwf_l[f][j].w);
                    }
                  }
              }
            }
          }
        }
      }
      if (KIND) {
#pragma unroll
        for (int i = 0; i < FR; ++i)
#pragma unroll
          for (int f = 0; f < NF; ++f)
            if (f < F) {
              const float tot = wave_sum(dcf_l[i][f]);
              if (lane == 0 && t0 + i * NW < hi) dcf[(t0 + i * NW) * F + f] = tot;
            }
      }
    }
  }
  // cross-wave sums of dq and dv (fixed order)
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int u4 = lane + 64 * j;
    if (u4 < U4) *reinterpret_cast<float4 *>(red + (size_t)w * U + 4 * u4) = dq_l[j];
  }
  __syncthreads();
  for (int u = tid; u < U; u += AT) {
    float s = 0.f;
    for (int i = 0; i < NW; ++i) s += red[i * U + u];
    xst(dq + u, s, fused);
  }
  if (DEFER && !fused) return;
  __syncthreads();
  if (!DEFER) {
#pragma unroll
  for (int j = 0; j < MAXJ; ++j) {
    const int u4 = lane + 64 * j;
    if (u4 < U4) *reinterpret_cast<float4 *>(red + (size_t)w * U + 4 * u4) = dv_l[j];
  }
  __syncthreads();
  for (int u = tid; u < U; u += AT) {
    float s = 0.f;
    for (int i = 0; i < NW; ++i) s += red[i * U + u];
    p.dv_part[((size_t)b * S + sl) * U + u] += s;
  }
  }
  if (fused) {
    if (!last_arriver(p.tickets + b, (unsigned)S, &last_flag)) return;
    for (int u = tid; u < U; u += AT) {
      float s = 0.f;
      for (int i = 0; i < S; ++i) s += xld(p.dq_part + ((size_t)b * S + i) * U + u);
      p.dq[(size_t)b * U + u] = s;
    }
    if (p.dalign_out)
      for (int t = tid; t < Te; t += AT) p.dalign_out[(size_t)b * Te + t] = 0.f;
    return;
  }
  if (KIND && !DEFER) {
    __syncthreads();
    if (REG) {
      // d conv_proj[f,u] += sum_t cf[t,f] * d[t,u]: the per-wave register sums, reduced across waves
      // one feature at a time (fixed order)
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        if (f < F) {
#pragma unroll
          for (int j = 0; j < RJ; ++j) {
            const int u4 = lane + 64 * j;
            if (u4 < U4) *reinterpret_cast<float4 *>(red + (size_t)w * U + 4 * u4) = dwf_l[f][j];
          }
          __syncthreads();
          for (int u = tid; u < U; u += AT) {
            float sm = 0.f;
            for (int i = 0; i < NW; ++i) sm += red[i * U + u];
            p.dwf_part[(((size_t)b * S + sl) * F + f) * U + u] += sm;
          }
          __syncthreads();
        }
      }
    } else
    // d conv_proj[f,u] += sum_t cf[t,f] * d[t,u] with d recomputed in a second pass over the frames
    for (int u = tid; u < U; u += AT) {
      float acc[16];
#pragma unroll
      for (int f = 0; f < 16; ++f) acc[f] = 0.f;
      for (int t = lo; t < hi; ++t) {
        float x = keys[(size_t)t * U + u] + q[u];
        for (int f = 0; f < F; ++f) x = fmaf(cf[t * F + f], p.wf[f * U + u], x);
        const float th = tanhf_(x);
        const float d = ds[t] * p.v[u] * (1.f - th * th);
#pragma unroll
        for (int f = 0; f < 16; ++f)
          if (f < F) acc[f] = fmaf(cf[t * F + f], d, acc[f]);
      }
#pragma unroll
      for (int f = 0; f < 16; ++f)
        if (f < F) p.dwf_part[(((size_t)b * S + sl) * F + f) * U + u] += acc[f];
    }
  }
}

// Location-aware attention backward of the step chain with its two small dense products on the matrix pipe (exact fp32:
// v_mfma_f32_16x16x4_f32) and every load that depends on nothing issued before the first wait (DEFER only; U <= 512,
// F <= 12, <= 32 frames per slice).  attn_bwd_kernel<2, true> spends its 29 us per launch (cfg5: 16 utterances x 8 slices
// of 25 frames) on ~9 dependent round trips to memory — values in 4, keys in 4, one frame pair of a wave at a time,
// 96 registers of projection rows and 24 of feature sums leaving room for nothing in flight — not on arithmetic.  Here
//   x^T[16 units x 16 frames]   = conv_proj^T[16 x 12] . features^T[12 x 16] + (keys^T + q)        3 instructions
//   d features[16 frames x 16]  = d[16 frames x 16 units] . conv_proj^T[16 units x 16 filters]      4 instructions
// per (unit tile, frame tile): the first product's accumulator IS the second one's A operand (lane = frame, register r =
// unit 4 kq + r of the tile: the k index of the second product is permuted the same way on both sides), so tanh and
// the score gradient are applied in place and nothing goes through LDS between the two.  A wave owns U/128 unit tiles
// and both frame tiles; d features are added over the waves in LDS, dq over the 16 frame lanes in the wave.
// -DATTN_STAMPS: wall_clock64 at the phase boundaries of workgroup (0, 0) (10 ns ticks), printed by attn_bwd_impl
#ifdef ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[16];
#define ASTAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_attn_stamps[i] = wall_clock64(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif
template <int UT>
__global__ __launch_bounds__(AT) void attn_bwd_loc_mfma_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int Te = p.Te, U = p.U, E = p.E, F = p.F;
  constexpr int NW = AT / 64;
  float *base = smem + ck_floats(p);
  float *alp = base;                       // [Te] previous alignment
  float *ds = alp + Te;                    // [Te] d score
  float *cf = ds + Te;                     // [Te*F]
  float *red = base + ((2 * Te + Te * F + 3) & ~3);       // [NW][2][16 frames][16 filters] (also scalars)
  if (p.step >= p.dec_len[b]) return;      // finished row: the finish kernel writes its zeros
  const int n = min(max(p.enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, n), hi = min(lo + per, n);
  float *dcf = p.dcf_g + (size_t)b * Te * F;
  const float *keys = p.keys + (size_t)b * Te * U;
  const float *vals = p.values + (size_t)b * Te * E;
  const float *q = p.q + (size_t)b * U;
  const float *al = p.align + (size_t)b * Te;
  const float *dctx = p.dctx + (size_t)b * E;
  const float *cx = p.ctx + (size_t)b * E;
  const float *dal_in = p.dalign_in ? p.dalign_in + (size_t)b * Te : nullptr;
  const int fl = lane & 15, kq = lane >> 4;
  const int E4 = E / 4, EC = (E4 + 63) / 64;
  ASTAMP(0);

  // ---- every load that depends on nothing, in the order of use (vector loads return in order).  Every address is
  // clamped into its array and the value masked afterwards: a conditional load is a branch around the load and a
  // wait in front of the next one (the first version of this kernel had 60 of them)
  typedef const f32x4_ *V4;
  const f32x4_ zv = {0.f, 0.f, 0.f, 0.f};
  constexpr int NAL = (1024 + AT - 1) / AT;
  // (1) previous alignment and conv kernel -> LDS
  float alp_r[NAL], ck_r[8];
#pragma unroll
  for (int i = 0; i < NAL; ++i) alp_r[i] = p.align_prev[(size_t)b * Te + min(tid + AT * i, Te - 1)];
#pragma unroll
  for (int i = 0; i < 8; ++i) ck_r[i] = p.ck[min(tid + AT * i, p.K * F - 1)];
  // (2) sum_t a[t] da[t] = dctx . context + sum_t a[t] dalign_in[t]
  float dr[4], cr[4], ar = 0.f, gr = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = min(tid + AT * i, E - 1);
    dr[i] = dctx[e];
    cr[i] = cx[e];
  }
  if (dal_in) { ar = al[min(tid, Te - 1)]; gr = dal_in[min(tid, Te - 1)]; }
  // (3) values of my frames (wave w: lo + w + NW i), 4 chunks of 64 x 16 bytes of the encoder dimension at a time
  f32x4_ vv[4][4], dc[4];
  const int tl = max(hi - 1, 0);
  auto issue_vals = [&](int c0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e4 = min(lane + 64 * (c0 + c), E4 - 1);
      dc[c] = reinterpret_cast<V4>(dctx)[e4];
#pragma unroll
      for (int i = 0; i < 4; ++i) vv[i][c] = reinterpret_cast<V4>(vals + (size_t)min(lo + w + NW * i, tl) * E)[e4];
    }
  };
  issue_vals(0);
  // (4) operands of the two products: keys (accumulator init), q, v, conv_proj in both layouts
  f32x4_ kc[2][UT], qv[UT], v4[UT], b2[UT];
  float a1[UT][3];
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const int u0 = min(16 * (w + NW * j), U - 16);
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
      kc[ft][j] = *reinterpret_cast<V4>(keys + (size_t)min(lo + 16 * ft + fl, tl) * U + u0 + 4 * kq);
    qv[j] = *reinterpret_cast<V4>(q + u0 + 4 * kq);
    v4[j] = *reinterpret_cast<V4>(p.v + u0 + 4 * kq);
    b2[j] = *reinterpret_cast<V4>(p.wf + (size_t)min(fl, F - 1) * U + u0 + 4 * kq);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a1[j][ks] = p.wf[(size_t)min(4 * ks + kq, F - 1) * U + u0 + fl];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- conv features of my frames
#pragma unroll
  for (int i = 0; i < NAL; ++i)
    if (tid + AT * i < Te) alp[tid + AT * i] = alp_r[i];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (tid + AT * i < p.K * F) smem[tid + AT * i] = ck_r[i];
  for (int i = tid + 8 * AT; i < p.K * F; i += AT) smem[i] = p.ck[i];          // (K F > 4096: not a shape of this path)
  for (int t = tid + AT * NAL; t < Te; t += AT) alp[t] = p.align_prev[(size_t)b * Te + t];
  __syncthreads();
  ASTAMP(1);
  {
    // location features of my frames, two threads per (frame, filter): the chain of one load pair + fma per tap is
    // LDS-latency bound, and (hi - lo) F outputs would leave half of the workgroup idle
    const int pb = (p.K - 1) / 2, nout = (hi - lo) * F, hf = tid & 1;
    for (int o = tid >> 1; o < nout; o += AT / 2) {
      const int i = lo * F + o, t = i / F, f = i % F;
      const int d0 = max(0, pb - t), d1 = min(p.K, Te + pb - t), mid = (d0 + d1 + 1) >> 1;
      const float *a = alp + t - pb, *c = smem + f;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      int d = hf ? mid : d0;
      const int de = hf ? d1 : mid;
      for (; d + 3 < de; d += 4) {
        s0 = fmaf(a[d], c[d * F], s0);
        s1 = fmaf(a[d + 1], c[(d + 1) * F], s1);
        s2 = fmaf(a[d + 2], c[(d + 2) * F], s2);
        s3 = fmaf(a[d + 3], c[(d + 3) * F], s3);
      }
      for (; d < de; ++d) s0 = fmaf(a[d], c[d * F], s0);
      float sm = (s0 + s1) + (s2 + s3);
      sm += __shfl_xor(sm, 1);
      if (!hf) cf[i] = sm;
    }
  }
  ASTAMP(2);
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) r = fmaf(tid + AT * i < E ? dr[i] : 0.f, cr[i], r);
  for (int e = tid + 4 * AT; e < E; e += AT) r = fmaf(dctx[e], cx[e], r);
  r = fmaf(tid < n ? ar : 0.f, gr, r);
  if (dal_in)
    for (int t = tid + AT; t < n; t += AT) r = fmaf(al[t], dal_in[t], r);
  r = wave_sum(r);
  if (lane == 0) red[w] = r;
  // ---- d alignment[t] = dctx . values[t] (+ the gradient arriving through next step's location features)
  {
    float sda[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < EC; c0 += 4) {
      if (c0) issue_vals(c0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4_ dcm = lane + 64 * (c0 + c) < E4 ? dc[c] : zv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4_ m = dcm * vv[i][c];
          sda[i] += (m[0] + m[1]) + (m[2] + m[3]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = lo + w + NW * i;
      const float tot = wave_sum(sda[i]);
      if (lane == 0 && t < hi) ds[t] = tot + (dal_in ? dal_in[t] : 0.f);
    }
  }
  ASTAMP(3);
  // mask what the clamped loads brought in for tiles / filters / frames that do not exist
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    if (fl >= F) b2[j] = zv;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
      if (4 * ks + kq >= F) a1[j][ks] = 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  // softmax backward: dscore = a * (da - sum a*da)
  r = 0.f;
  for (int i = 0; i < NW; ++i) r += red[i];
  __syncthreads();
  if (p.prob_fn == 0) {
    for (int t = lo + tid; t < hi; t += AT) ds[t] = al[t] * (ds[t] - r);
  } else if (p.prob_fn == 1) {
    for (int t = lo + tid; t < hi; t += AT) ds[t] = ds[t] * al[t] * (1.f - al[t]);
  } else {
    const float z = p.znorm[b];
    for (int t = lo + tid; t < hi; t += AT) ds[t] = (ds[t] - r) * al[t] * (1.f - al[t] * z);
  }
  __syncthreads();
  for (int t = lo + tid; t < hi; t += AT) p.ds_out[(size_t)b * Te + t] = ds[t];
  for (int i = lo * F + tid; i < hi * F; i += AT) p.cf_out[(size_t)b * Te * F + i] = cf[i];
  ASTAMP(4);
  // ---- through v . tanh(keys + q + features . conv_proj)
  float b1[2][3], g[2];
#pragma unroll
  for (int ft = 0; ft < 2; ++ft) {
    const int t = lo + 16 * ft + fl;
    g[ft] = t < hi ? ds[t] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) b1[ft][ks] = (t < hi && 4 * ks + kq < F) ? cf[t * F + 4 * ks + kq] : 0.f;
  }
  f32x4_ acc2[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const int u0 = 16 * (w + NW * j);
    if (u0 >= U) break;
    f32x4_ dqa = {0.f, 0.f, 0.f, 0.f};
    const f32x4_ vj = v4[j], bj = b2[j];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
      f32x4_ x = kc[ft][j] + qv[j];
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) x = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j][ks], b1[ft][ks], x, 0, 0, 0);
      f32x4_ d;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // one exp + one rcp per tanh (as attn_param_grads_kernel and the persistent kernels' score; relative error ~1e-7)
        const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[c])) - 1.0f;
        d[c] = g[ft] * vj[c] * (1.f - th * th);
      }
      dqa += d;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc2[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[c], bj[c], acc2[ft], 0, 0, 0);
    }
    // dq of my 4 units: the 16 frame lanes of my k group
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = dqa[c];
      v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
      dqa[c] = v;
    }
    if (fl == 0)
      *reinterpret_cast<float4 *>(p.dq_part + ((size_t)b * S + sl) * U + u0 + 4 * kq) = make_float4(dqa[0], dqa[1], dqa[2], dqa[3]);
  }
  ASTAMP(5);
  // d features: add the waves (fixed order); accumulator register c = frame 4 kq + c of the tile, lane fl = filter
#pragma unroll
  for (int ft = 0; ft < 2; ++ft)
#pragma unroll
    for (int c = 0; c < 4; ++c) red[((w * 2 + ft) * 16 + 4 * kq + c) * 16 + fl] = acc2[ft][c];
  __syncthreads();
  {
    const int ft = tid >> 8, fr = (tid >> 4) & 15, f = tid & 15, t = lo + 16 * ft + fr;
    if (t < hi && f < F) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < NW; ++i) sm += red[((i * 2 + ft) * 16 + fr) * 16 + f];
      dcf[t * F + f] = sm;
    }
  }
  ASTAMP(6);
}

// The sums over decoder steps that the step chain does not wait for (DEFER): for its frames [lo, hi) of utterance b a
// workgroup walks the steps l < dec_len[b] and accumulates in registers
//   d keys[t,u] = sum_l d_l[t,u],  d v[u] += sum_{l,t} ds_l[t] tanh(x_l[t,u]),  d conv_proj[f,u] += sum_{l,t} cf_l[t,f] d_l[t,u]
// with x_l[t,u] = keys[t,u] + q_l[u] + cf_l[t,:].conv_proj[:,u],  d_l[t,u] = ds_l[t] v[u] (1 - tanh(x)^2)
// from the d scores (and location features) the chain's attention kernels saved: no dependency between steps here, one
// launch of B x S workgroups instead of a read-modify-write of dkeys in every step.  grid (B, S), PT threads:
// thread = (16-byte unit group u4, frame group fg); frames of a thread: lo + fg, lo + fg + NG, ...
constexpr int PT = 512, PNF = 12, PSB = 4;   // threads; location filters in registers (max); steps per barrier pair
template <bool KIND, int PFR>                // PFR = frames per thread (max)
__global__ __launch_bounds__(PT) void attn_param_grads_kernel(int B, int Te, int U, int F, int L, const int32_t *dec_len,
                                                              const int32_t *enc_len, const float *keys, const float *q_all,
                                                              const float *v, const float *wf, const float *ds_all,
                                                              const float *cf_all, float *dkeys, float *dv_part,
                                                              float *dwf_part) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
  const int n = min(max(enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, n), hi = min(lo + per, n);
  const int U4 = U / 4, NG = PT / U4;            // frame groups
  const int u4 = tid % U4, fg = tid / U4;
  const bool act = fg < NG;
  float *ds_s = psm;                              // [PSB][per] d scores of the steps in flight
  float *cf_s = ds_s + PSB * ((per + 3) & ~3);    // [PSB][per][F]
  float *red = cf_s + PSB * ((per * F + 3) & ~3); // [NG][U] cross-group sums at the end
  float4 kx[PFR], dk[PFR], wfr[KIND ? PNF : 1], dwf[KIND ? PNF : 1];
  float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 vv = act ? reinterpret_cast<const float4 *>(v)[u4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < PFR; ++i) {
    const int t = lo + fg + NG * i;
    dk[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    kx[i] = (act && t < hi) ? reinterpret_cast<const float4 *>(keys + ((size_t)b * Te + t) * U)[u4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (KIND) {
#pragma unroll
    for (int f = 0; f < PNF; ++f) {
      dwf[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      wfr[f] = (act && f < F) ? reinterpret_cast<const float4 *>(wf + (size_t)f * U)[u4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const int steps = min(max(dec_len[b], 0), L);
  const int DSP = (per + 3) & ~3, CFP = (per * F + 3) & ~3;      // per-step strides of the staged d scores / features
  for (int l0 = 0; l0 < steps; l0 += PSB) {
    // PSB steps per barrier pair: their d scores, features and queries travel together
    __syncthreads();
    for (int i = tid; i < PSB * (hi - lo); i += PT) {
      const int sb = i / (hi - lo), t = lo + i % (hi - lo);
      ds_s[sb * DSP + t - lo] = l0 + sb < steps ? ds_all[((size_t)(l0 + sb) * B + b) * Te + t] : 0.f;
    }
    if (KIND)
      for (int i = tid; i < PSB * (hi - lo) * F; i += PT) {
        const int sb = i / ((hi - lo) * F), r = i % ((hi - lo) * F);
        cf_s[sb * CFP + r] = l0 + sb < steps ? cf_all[((size_t)(l0 + sb) * B + b) * Te * F + lo * F + r] : 0.f;
      }
    float4 qs[PSB];
#pragma unroll
    for (int sb = 0; sb < PSB; ++sb)
      qs[sb] = (act && l0 + sb < steps) ? reinterpret_cast<const float4 *>(q_all + ((size_t)(l0 + sb) * B + b) * U)[u4]
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (!act) continue;
#pragma unroll
    for (int sb = 0; sb < PSB; ++sb) {
    if (l0 + sb >= steps) break;
    const float4 qq = qs[sb];
    const float *ds_s_ = ds_s + sb * DSP, *cf_s_ = cf_s + sb * CFP;
#pragma unroll
    for (int i = 0; i < PFR; ++i) {
      const int t = lo + fg + NG * i;
      if (t < hi) {
        const float g = ds_s_[t - lo];
        float x[4] = {kx[i].x + qq.x, kx[i].y + qq.y, kx[i].z + qq.z, kx[i].w + qq.w};
        if (KIND) {
#pragma unroll
          for (int f = 0; f < PNF; ++f)
            if (f < F) {
              const float c = cf_s_[(t - lo) * F + f];
              x[0] = fmaf(c, wfr[f].x, x[0]); x[1] = fmaf(c, wfr[f].y, x[1]); x[2] = fmaf(c, wfr[f].z, x[2]); x[3] = fmaf(c, wfr[f].w, x[3]);
            }
        }
        // one exp + one rcp per tanh (as the persistent kernels; relative error ~1e-7)
        const float th0 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[0])) - 1.0f, th1 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[1])) - 1.0f;
        const float th2 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[2])) - 1.0f, th3 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[3])) - 1.0f;
        const float d0 = g * vv.x * (1.f - th0 * th0), d1 = g * vv.y * (1.f - th1 * th1);
        const float d2 = g * vv.z * (1.f - th2 * th2), d3 = g * vv.w * (1.f - th3 * th3);
        dk[i].x += d0; dk[i].y += d1; dk[i].z += d2; dk[i].w += d3;
        dv.x = fmaf(g, th0, dv.x); dv.y = fmaf(g, th1, dv.y); dv.z = fmaf(g, th2, dv.z); dv.w = fmaf(g, th3, dv.w);
        if (KIND) {
#pragma unroll
          for (int f = 0; f < PNF; ++f)
            if (f < F) {
              const float c = cf_s_[(t - lo) * F + f];
              dwf[f].x = fmaf(c, d0, dwf[f].x); dwf[f].y = fmaf(c, d1, dwf[f].y);
              dwf[f].z = fmaf(c, d2, dwf[f].z); dwf[f].w = fmaf(c, d3, dwf[f].w);
            }
        }
      }
    }
    }
  }
  // d keys: my frames (frames >= enc_len keep the caller's zeros)
  if (act) {
#pragma unroll
    for (int i = 0; i < PFR; ++i) {
      const int t = lo + fg + NG * i;
      if (t < hi) reinterpret_cast<float4 *>(dkeys + ((size_t)b * Te + t) * U)[u4] = dk[i];
    }
  }
  // d v and d conv_proj: add the frame groups (fixed order) -> one partial row per (utterance, slice)
  __syncthreads();
  if (act) *reinterpret_cast<float4 *>(red + (size_t)fg * U + 4 * u4) = dv;
  __syncthreads();
  for (int u = tid; u < U; u += PT) {
    float sm = 0.f;
    for (int g = 0; g < NG; ++g) sm += red[(size_t)g * U + u];
    dv_part[((size_t)b * S + sl) * U + u] = sm;
  }
  if (KIND) {
#pragma unroll
    for (int f = 0; f < PNF; ++f) {
      if (f < F) {
        __syncthreads();
        if (act) *reinterpret_cast<float4 *>(red + (size_t)fg * U + 4 * u4) = dwf[f];
        __syncthreads();
        for (int u = tid; u < U; u += PT) {
          float sm = 0.f;
          for (int g = 0; g < NG; ++g) sm += red[(size_t)g * U + u];
          dwf_part[(((size_t)b * S + sl) * F + f) * U + u] = sm;
        }
      }
    }
  }
}

// attn_param_grads_kernel<true, .> with its two small dense products on the matrix pipe (exact fp32,
// v_mfma_f32_16x16x4_f32), U <= 512, F <= 12, <= 16 * FTN frames per slice.  Orientation: x[16 frames x 16 units], so
// that accumulator register c of lane (kq, fl) = frame 4 kq + c, unit 16 T + fl — keys, d keys (registers for the whole
// launch) and the stores run along the units, and d IS the A operand of
//   d conv_proj^T[16 units x 16 filters] += d^T[16 units x 16 frames] . features[16 frames x 16 filters]
// (k = frame 4 kq + c on both sides).  A wave owns the unit tiles T = w + 8 j; nothing crosses waves.
template <int FTN>
__global__ __launch_bounds__(PT) void attn_param_grads_mfma_kernel(int B, int Te, int U, int F, int L, const int32_t *dec_len,
                                                                   const int32_t *enc_len, const float *keys,
                                                                   const float *q_all, const float *v, const float *wf,
                                                                   const float *ds_all, const float *cf_all, float *dkeys,
                                                                   float *dv_part, float *dwf_part) {
  extern __shared__ __attribute__((aligned(16))) float psm[];
  constexpr int UT = 4, NWV = PT / 64;
  const int b = blockIdx.x, sl = blockIdx.y, S = gridDim.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fl = lane & 15, kq = lane >> 4;
  const int n = min(max(enc_len[b], 0), Te);
  const int per = (Te + S - 1) / S, lo = min(sl * per, n), hi = min(lo + per, n), nf = hi - lo;
  const int DSP = 16 * FTN, CFP = (16 * FTN * F + 3) & ~3;
  float *ds_s = psm;                      // [PSB][16 FTN] d scores of the steps in flight (zero past my frames)
  float *cf_s = ds_s + PSB * DSP;         // [PSB][16 FTN][F]
  const int tl = max(hi - 1, 0);
  f32x4_ kx[FTN][UT], dk[FTN][UT], dwf[UT];
  float b1[UT][3], vv[UT], dv[UT];
  const f32x4_ zv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const int u = min(16 * (w + NWV * j), U - 16) + fl;
    vv[j] = v[u];
    dv[j] = 0.f;
    dwf[j] = zv;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) b1[j][ks] = wf[(size_t)min(4 * ks + kq, F - 1) * U + u];
#pragma unroll
    for (int ft = 0; ft < FTN; ++ft) {
      dk[ft][j] = zv;
#pragma unroll
      for (int c = 0; c < 4; ++c) kx[ft][j][c] = keys[((size_t)b * Te + min(lo + 16 * ft + 4 * kq + c, tl)) * U + u];
    }
  }
  const int steps = min(max(dec_len[b], 0), L);
  for (int l0 = 0; l0 < steps; l0 += PSB) {
    __syncthreads();
    for (int i = tid; i < PSB * DSP; i += PT) {
      const int sb = i / DSP, t = i % DSP;
      ds_s[i] = (l0 + sb < steps && t < nf) ? ds_all[((size_t)(l0 + sb) * B + b) * Te + lo + t] : 0.f;
    }
    for (int i = tid; i < PSB * DSP * F; i += PT) {
      const int sb = i / (DSP * F), r = i % (DSP * F);
      cf_s[sb * CFP + r] = (l0 + sb < steps && r < nf * F) ? cf_all[((size_t)(l0 + sb) * B + b) * Te * F + lo * F + r] : 0.f;
    }
    float qs[PSB][UT];
#pragma unroll
    for (int sb = 0; sb < PSB; ++sb)
#pragma unroll
      for (int j = 0; j < UT; ++j)
        qs[sb][j] = q_all[((size_t)min(l0 + sb, steps - 1) * B + b) * U + min(16 * (w + NWV * j), U - 16) + fl];
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < PSB; ++sb) {
      if (l0 + sb >= steps) break;
      const float *dss = ds_s + sb * DSP, *cfs = cf_s + sb * CFP;
      // operands of the step shared by my unit tiles: features as A (frame fl, filter 4 ks + kq) and as B (frame
      // 4 kq + c, filter fl); d scores of frames 4 kq + c
      float a1[FTN][3];
      f32x4_ b3[FTN], g[FTN];
#pragma unroll
      for (int ft = 0; ft < FTN; ++ft) {
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) a1[ft][ks] = 4 * ks + kq < F ? cfs[(16 * ft + fl) * F + min(4 * ks + kq, F - 1)] : 0.f;
        g[ft] = *reinterpret_cast<const f32x4_ *>(dss + 16 * ft + 4 * kq);
#pragma unroll
        for (int c = 0; c < 4; ++c) b3[ft][c] = fl < F ? cfs[(16 * ft + 4 * kq + c) * F + min(fl, F - 1)] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < UT; ++j) {
        if (16 * (w + NWV * j) >= U) break;
        const float qq = qs[sb][j];
#pragma unroll
        for (int ft = 0; ft < FTN; ++ft) {
          f32x4_ x = kx[ft][j] + qq;
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) x = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ft][ks], b1[j][ks], x, 0, 0, 0);
          f32x4_ d;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x[c])) - 1.0f;
            d[c] = g[ft][c] * vv[j] * (1.f - th * th);
            dv[j] = fmaf(g[ft][c], th, dv[j]);
          }
          dk[ft][j] += d;
#pragma unroll
          for (int c = 0; c < 4; ++c) dwf[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(d[c], b3[ft][c], dwf[j], 0, 0, 0);
        }
      }
    }
  }
  // d keys of my frames (frames >= enc_len keep the caller's zeros); d v: my frames' sum, the four k groups added;
  // d conv_proj^T: accumulator register r = unit 4 kq + r of the tile, lane fl = filter
#pragma unroll
  for (int j = 0; j < UT; ++j) {
    const int u0 = 16 * (w + NWV * j);
    if (u0 >= U) break;
#pragma unroll
    for (int ft = 0; ft < FTN; ++ft)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = lo + 16 * ft + 4 * kq + c;
        if (t < hi) dkeys[((size_t)b * Te + t) * U + u0 + fl] = dk[ft][j][c];
      }
    float sv = dv[j];
    sv += __shfl_xor(sv, 16);
    sv += __shfl_xor(sv, 32);
    if (kq == 0) dv_part[((size_t)b * S + sl) * U + u0 + fl] = sv;
    if (fl < F)
#pragma unroll
      for (int r = 0; r < 4; ++r) dwf_part[(((size_t)b * S + sl) * F + fl) * U + u0 + 4 * kq + r] = dwf[j][r];
  }
}

// What needs every frame of an utterance: dq = sum of the slices' partials; for location-aware
// attention the gradient w.r.t. the previous alignments and the conv kernel, from the d location
// features the slices left in HBM.  grid (B, 2) for location-aware attention (y = 0: dq and d previous
// alignment, y = 1: d conv kernel), else (B, 1); FT threads; previous alignment, d features and the
// conv kernel are staged in LDS: alp[Te], dcf[Te*F], ck[K*F].  Both sums are LDS-latency bound when
// written as one dependent load + fma per iteration, so the work is spread over 1024 threads (four
// threads per output frame, one per conv-kernel entry) with independent accumulator chains.
constexpr int FT = 1024;
constexpr int FPA = 4, FPK = 2;      // location-aware: workgroups per utterance for d previous alignment / d conv kernel
__global__ __launch_bounds__(FT) void attn_bwd_finish_kernel(AttnArgs p, int S) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // grid (B, 1) or, location-aware, (B, FPA + FPK): y < FPA: a quarter of the output frames of d previous alignment
  // (y = 0 also sums dq), y >= FPA: half of the conv kernel's entries.  Both sums are LDS-latency bound chains of one
  // load pair + fma, so they are spread wide: 16 threads per output frame, one per conv-kernel entry.
  const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, NT = blockDim.x;
  const int Te = p.Te, U = p.U, F = p.F;
  const bool loc = p.kind == 1;
  float *dq = p.dq + (size_t)b * U;
  float *dal_out = p.dalign_out ? p.dalign_out + (size_t)b * Te : nullptr;
  const int tper = loc ? (Te + FPA - 1) / FPA : Te;             // output frames of a part
  const int ta = loc ? min(part * tper, Te) : 0, tb = loc ? min(ta + tper, Te) : Te;
  if (p.step >= p.dec_len[b]) {            // finished row: no gradient of its own, pass dalign through
    if (part == 0)
      for (int u = tid; u < U; u += NT) dq[u] = 0.f;
    if (dal_out && (!loc || part < FPA))
      for (int t = ta + tid; t < tb; t += NT) dal_out[t] = p.dalign_in ? p.dalign_in[(size_t)b * Te + t] : 0.f;
    return;
  }
  if (part == 0)
    for (int u = tid; u < U; u += NT) {
      float s = 0.f;
      for (int i = 0; i < S; ++i) s += p.dq_part[((size_t)b * S + i) * U + u];
      dq[u] = s;
    }
  if (!loc) {
    if (dal_out)
      for (int t = tid; t < Te; t += NT) dal_out[t] = 0.f;
    return;
  }
  const int n = min(max(p.enc_len[b], 0), Te);
  float *alp = smem, *dcf = alp + Te, *ck = dcf + Te * F;
  for (int t = tid; t < Te; t += NT) alp[t] = p.align_prev[(size_t)b * Te + t];
  // frames past the length carry no gradient
  for (int i = tid; i < Te * F; i += NT) dcf[i] = i < n * F ? p.dcf_g[(size_t)b * Te * F + i] : 0.f;
  for (int i = tid; i < p.K * F; i += NT) ck[i] = p.ck[i];
  __syncthreads();
  const int pb = (p.K - 1) / 2;
  if (part < FPA) {
    // out frame to = t - d + pb receives a[t] * ck[d,f]; thread (t, q) takes the taps d = d0 + q, + 16, ...
    for (int t0 = ta; t0 < tb; t0 += NT / 16) {
      const int t = t0 + (tid >> 4), q = tid & 15;
      float s0 = 0.f, s1 = 0.f;
      if (t < tb) {
        const int d0 = max(0, t + pb - (n - 1)), d1 = min(p.K, t + pb + 1);
        int d = d0 + q;
        for (; d + 16 < d1; d += 32) {
          const float *g = dcf + (t - d + pb) * F, *c = ck + d * F;
          for (int f = 0; f < F; ++f) {
            s0 = fmaf(g[f], c[f], s0);
            s1 = fmaf(g[f - 16 * F], c[f + 16 * F], s1);
          }
        }
        for (; d < d1; d += 16) {
          const float *g = dcf + (t - d + pb) * F, *c = ck + d * F;
          for (int f = 0; f < F; ++f) s0 = fmaf(g[f], c[f], s0);
        }
      }
      float s = s0 + s1;
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      if (t < tb && q == 0) dal_out[t] = s;
    }
  } else {
    const int KF = p.K * F, half = (KF + FPK - 1) / FPK, i0 = (part - FPA) * half, i1 = min(i0 + half, KF);
    for (int i = i0 + tid; i < i1; i += NT) {
      const int d = i / F, f = i % F;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      const int to0 = max(0, pb - d), to1 = min(n, Te + pb - d);
      const float *a = alp + d - pb, *g = dcf + f;
      int to = to0;
      for (; to + 3 < to1; to += 4) {
        s0 = fmaf(a[to], g[to * F], s0);
        s1 = fmaf(a[to + 1], g[(to + 1) * F], s1);
        s2 = fmaf(a[to + 2], g[(to + 2) * F], s2);
        s3 = fmaf(a[to + 3], g[(to + 3) * F], s3);
      }
      for (; to < to1; ++to) s0 = fmaf(a[to], g[to * F], s0);
      p.dck_part[(size_t)b * p.K * F + i] += (s0 + s1) + (s2 + s3);
    }
  }
}

// ---------------------------------------------------------------------------
// helpers
__global__ __launch_bounds__(256) void mask_time_kernel(int B, int L, int F, float *__restrict__ x,
                                                        const int32_t *__restrict__ len) {
  const size_t total = (size_t)B * L * F;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t bt = i / F;
    const int b = (int)(bt / L), t = (int)(bt % L);
    if (t >= len[b]) x[i] = 0.f;
  }
}

// y[b,l,f] = x[l,b,f]  (time-major <-> batch-major)
__global__ __launch_bounds__(256) void swap01_kernel(int L, int B, int F, const float *__restrict__ x,
                                                     float *__restrict__ y) {
  const size_t total = (size_t)B * L * F;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t f = i % F, bl = i / F;
    const size_t l = bl % L, b = bl / L;
    y[i] = x[(l * B + b) * F + f];
  }
}

// dK[c,:] = sum over (l,b) with ids[l,b] == c of dz[l,b,:]  (gradient of the one-hot rows).
// Workgroup (column block, class c): the ids are scanned 256 at a time through LDS (one coalesced load
// per chunk instead of a dependent global load per row), matching rows are added in increasing row
// order (deterministic).
// Two passes per segment of SEG rows: (1) the segment's matching row numbers into LDS (ballots, no loads of dz),
// (2) the rows added eight loads at a time.  (The first version added each chunk's ~5 hits as they were found: one
// dependent global load per hit behind four barriers per 256 ids — 0.62 ms for cfg5's [10 176 x 2048].)
constexpr int SCATTER_SEG = 8192;
__global__ __launch_bounds__(256) void scatter_rows_kernel(int C, int N, int W, const int32_t *__restrict__ ids,
                                                           const float *__restrict__ dz, float *__restrict__ dK) {
  __shared__ int hit[SCATTER_SEG];
  __shared__ int wcount[2][4];      // double-buffered by chunk parity: ONE barrier per chunk, no re-read behind it
  const int c = blockIdx.y, tid = threadIdx.x;
  const int col = min(blockIdx.x * 256 + tid, W - 1);
  float s = 0.f;
  for (int seg = 0; seg < N; seg += SCATTER_SEG) {
    const int send = min(seg + SCATTER_SEG, N);
    // every thread keeps the running number of hits itself (the same sum in every thread): no shared counter that a
    // fast wave could overwrite while a slow one still reads it
    int nhit = 0, par = 0;
    for (int base = seg; base < send; base += 256, par ^= 1) {
      const int i = base + tid;
      const bool m = i < send && ids[i] == c;
      // order-preserving compaction: matches of wave w go after those of the waves before it
      const unsigned long long bal = __ballot(m);
      if ((tid & 63) == 0) wcount[par][tid >> 6] = __popcll(bal);
      __syncthreads();
      const int c0 = wcount[par][0], c1 = wcount[par][1], c2 = wcount[par][2], c3 = wcount[par][3];
      const int wv = tid >> 6;
      const int off = nhit + (wv > 0 ? c0 : 0) + (wv > 1 ? c1 : 0) + (wv > 2 ? c2 : 0);
      if (m) hit[off + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = i;
      nhit += c0 + c1 + c2 + c3;
      // (the next chunk writes wcount[par ^ 1]; wcount[par] is rewritten two chunks on, behind the next barrier)
    }
    __syncthreads();
    const int n = nhit;
    int j = 0;
    for (; j + 8 <= n; j += 8) {          // increasing row order, eight loads in flight
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = dz[(size_t)hit[j + k] * W + col];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; j < n; ++j) s += dz[(size_t)hit[j] * W + col];
    __syncthreads();
  }
  if (blockIdx.x * 256 + tid < W) dK[(size_t)c * W + col] = s;
}

// out[c][r] = in[r][c] (32x32 LDS tiles); used once per backward pass to turn the per-step
// dz·W^T products into row-major products the skinny GEMM kernel takes
__global__ __launch_bounds__(256) void transpose_kernel(int R, int C, const float *__restrict__ in, int ldin,
                                                        float *__restrict__ out) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < R && c0 + tx < C) tile[i][tx] = in[(size_t)(r0 + i) * ldin + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < C && r0 + tx < R) out[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

static int transpose(int R, int C, const float *in, int ldin, float *out, hipStream_t s) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, R, C, in, ldin, out);
  NABU_LAUNCH_CHECK();
  return 0;
}

// x[r, 0] = 1 for every row: WindowedAttention.initial_alignments (attention.py:352-359)
__global__ __launch_bounds__(256) void first_col_one_kernel(int rows, int ld, float *__restrict__ x) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < rows) x[(size_t)r * ld] = 1.f;
}
int first_col_one(int rows, int ld, float *x, hipStream_t s) {
  hipLaunchKernelGGL(first_col_one_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, rows, ld, x);
  NABU_LAUNCH_CHECK();
  return 0;
}

// out[r][4u+g] = in[r][gU+u]: the gate-interleaved copy of a cell kernel's dense rows, so that a
// 32-column slice of the step product holds all four gates of 8 units (LSTM-cell epilogue, gemm_skinny.hip)
__global__ __launch_bounds__(256) void permute_gates_kernel(int R, int U, const float *__restrict__ in,
                                                           float *__restrict__ out) {
  const size_t n = (size_t)R * 4 * U;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / (4 * U);
    const int c = (int)(i % (4 * U)), u = c >> 2, g = c & 3;
    out[i] = in[r * 4 * U + (size_t)g * U + u];
  }
}
// dst[r][c] += src[r][c] for row-strided operands
__global__ __launch_bounds__(256) void add_rows_kernel(int R, int Cn, const float *__restrict__ src, int lds_,
                                                      float *__restrict__ dst, int ldd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= R * Cn) return;
  const int r = i / Cn, c = i % Cn;
  dst[(size_t)r * ldd + c] += src[(size_t)r * lds_ + c];
}

static int grid1(size_t n) {
  size_t b = (n + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

static size_t attn_lds(const nabu_attn_desc *d, bool bwd) {
  size_t f = 2 * (size_t)d->Te + (d->kind == 1 ? (size_t)d->Te * d->F * (bwd ? 2 : 1) : 0);
  f += bwd ? 4 + (size_t)(AT / 64) * d->U : 64 + 4 + 4 * (size_t)AT;
  if (d->kind == 1) f += ((size_t)d->K * d->F + 3) & ~(size_t)3;     // the staged conv kernel
  return f * sizeof(float);
}

// Backward: frame slices per utterance so that the launch has ~256 workgroups (one per CU); a slice
// keeps at least 16 encoder frames (8 waves x 2 frames in flight)
static int attn_bwd_nslices(const nabu_attn_desc *d) {
  int S = (512 + d->B - 1) / d->B;      // two 512-thread workgroups per CU: four waves per SIMD hide the frame latency
  const int cap = (d->Te + 15) / 16;
  if (S > cap) S = cap;
  if (S > 8) S = 8;
  return S < 1 ? 1 : S;
}

static int check_attn(const nabu_attn_desc *d) {
  if (!d || d->size != sizeof(nabu_attn_desc)) return fail(NABU_EINVAL, "attention: bad descriptor size");
  if (d->B <= 0 || d->Te <= 0 || d->E <= 0 || d->U <= 0) return fail(NABU_EINVAL, "attention: bad dimensions");
  if (d->U > 1024) return fail(NABU_EUNSUP, "attention: num_units > 1024");
  if (d->U % 4 || d->E % 4) return fail(NABU_EUNSUP, "attention: num_units and encoder dim must be multiples of 4");
  if (d->kind < 0 || d->kind > 2) return fail(NABU_EINVAL, "attention: unknown kind");
  if (d->prob_fn < 0 || d->prob_fn > 2) return fail(NABU_EINVAL, "attention: unknown probability_fn");
  if (d->kind == 1 && (d->K <= 0 || d->F <= 0 || d->F > 16)) return fail(NABU_EUNSUP, "attention: numfilt must be 1..16");
  if (d->kind == 2 && (d->K < 0 || d->F < 1)) return fail(NABU_EINVAL, "attention: windowed needs left_window_width >= 0, right_window_width >= 1");
  if (attn_lds(d, true) > 150 * 1024) return fail(NABU_EUNSUP, "attention: encoder length too large for LDS");
  return 0;
}

}  // namespace nabu

using namespace nabu;

extern "C" int nabu_lstm_cell_fwd(int B, int U, int step, const int32_t *seq_len, const float *z,
                                  const float *bias, const float *emb_rows, const int32_t *ids,
                                  const float *c_prev, const float *h_prev, float *acts, float *c_new,
                                  float *h_new, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && U > 0 && seq_len && z && bias && c_prev && h_prev && acts && c_new && h_new,
                 "lstm_cell_fwd: bad argument");
  NABU_CHECK_ARG((emb_rows == nullptr) == (ids == nullptr), "lstm_cell_fwd: emb_rows and ids go together");
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((B * U + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, U, step, seq_len, z, bias, emb_rows, ids, c_prev,
                     h_prev, acts, c_new, h_new);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_lstm_cell_bwd(int B, int U, int step, const int32_t *seq_len, const float *acts,
                                  const float *c_new, const float *c_prev, const float *dh,
                                  const float *dh2, const float *dc_in, float *dz, float *dc_out,
                                  nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && U > 0 && seq_len && acts && c_new && c_prev && dh && dc_in && dz && dc_out,
                 "lstm_cell_bwd: bad argument");
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((B * U + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, U, step, seq_len, acts, c_new, c_prev, dh, dh2,
                     dc_in, dz, dc_out);
  NABU_LAUNCH_CHECK();
  return 0;
}

// tickets: B zeroed counters (left zero) -> the finish steps run inside the attention launches
static int attn_fwd_impl(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                         const int32_t *enc_len, const float *keys, const float *values,
                         const float *q, const float *v, const float *conv_kernel,
                         const float *conv_proj, const float *align_prev, const float *ctx_prev,
                         float *align, float *ctx, float *znorm, void *ws, size_t ws_bytes,
                         nabu_stream_t stream, unsigned *tickets) {
  if (int e = check_attn(d)) return e;
  NABU_CHECK_ARG(dec_len && enc_len && keys && values && q && v && align_prev && ctx_prev && align && ctx,
                 "attn_fwd: null pointer");
  const int S = attn_bwd_nslices(d);     // frame slices per utterance (1 for batches that fill the chip)
  if (S > 1 && (!ws || ws_bytes < nabu_attn_fwd_ws_bytes(d))) return fail(NABU_EWS, "attn_fwd: workspace too small");
  NABU_CHECK_ARG(d->kind != 1 || (conv_kernel && conv_proj), "attn_fwd: location-aware attention needs its kernels");
  AttnArgs p = {};
  p.B = d->B; p.Te = d->Te; p.E = d->E; p.U = d->U; p.kind = d->kind; p.K = d->K; p.F = d->F; p.step = step;
  p.dec_len = dec_len; p.enc_len = enc_len; p.keys = keys; p.values = values; p.q = q; p.v = v;
  p.ck = conv_kernel; p.wf = conv_proj; p.align_prev = align_prev; p.ctx_prev = ctx_prev;
  p.align = align; p.ctx = ctx; p.prob_fn = d->prob_fn; p.znorm = znorm;
  p.fwd_part = static_cast<float *>(ws);
  p.tickets = tickets;
  const size_t shm = attn_lds(d, false);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool reg = d->kind == 1 && d->U <= 256 * RJ && d->F <= RF;
  auto kern = d->kind != 1 ? attn_fwd_kernel<0> : reg ? attn_fwd_kernel<2> : attn_fwd_kernel<1>;
  // the sliced location-aware softmax form of the step chain: the matrix-pipe kernel wherever its geometry holds
  // (attn_fwd_kernel<2> stays the general kernel: more frames per slice, U not a multiple of 16, sigmoid probabilities)
  if (d->kind == 1 && reg && d->prob_fn == 0 && S > 1 && tickets && (d->Te + S - 1) / S <= 32 &&
      d->U % 16 == 0 && d->E / 4 <= AT / 2 && d->Te <= 1024 && d->K * d->F <= 8 * AT) {
    const int UT = (d->U / 16 + AT / 64 - 1) / (AT / 64);
    kern = UT <= 1 ? attn_fwd_loc_mfma_kernel<1> : UT == 2 ? attn_fwd_loc_mfma_kernel<2> : UT == 3 ? attn_fwd_loc_mfma_kernel<3>
                                                                                                   : attn_fwd_loc_mfma_kernel<4>;
  }
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(kern, dim3(d->B, S), dim3(AT), shm, s, p);
  NABU_LAUNCH_CHECK();
  if (S > 1 && !tickets) {
    hipLaunchKernelGGL(attn_fwd_finish_kernel, dim3(d->B), dim3(256), 0, s, p, S);
    NABU_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int nabu_attn_fwd(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                             const int32_t *enc_len, const float *keys, const float *values,
                             const float *q, const float *v, const float *conv_kernel,
                             const float *conv_proj, const float *align_prev, const float *ctx_prev,
                             float *align, float *ctx, float *znorm, void *ws, size_t ws_bytes,
                             nabu_stream_t stream) {
  return attn_fwd_impl(d, step, dec_len, enc_len, keys, values, q, v, conv_kernel, conv_proj, align_prev, ctx_prev,
                       align, ctx, znorm, ws, ws_bytes, stream, nullptr);
}

extern "C" size_t nabu_attn_fwd_ws_bytes(const nabu_attn_desc *d) {
  if (check_attn(d)) return 0;
  const size_t S = attn_bwd_nslices(d);
  return S > 1 ? (size_t)d->B * S * ((size_t)d->E + 4) * sizeof(float) : 0;
}

extern "C" int nabu_attn_bwd_slices(const nabu_attn_desc *d) {
  if (check_attn(d)) return 0;
  return attn_bwd_nslices(d);
}

extern "C" size_t nabu_attn_bwd_ws_bytes(const nabu_attn_desc *d) {
  if (check_attn(d)) return 0;
  const size_t S = attn_bwd_nslices(d);
  return ((size_t)d->B * S * d->U + (d->kind == 1 ? (size_t)d->B * d->Te * d->F : 0) + 4) * sizeof(float);
}

static int attn_bwd_impl(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                         const int32_t *enc_len, const float *keys, const float *values,
                         const float *q, const float *v, const float *conv_kernel,
                         const float *conv_proj, const float *align_prev, const float *align,
                         const float *ctx, const float *dctx, const float *dalign_in, float *dq,
                         float *dkeys, float *dv_part, float *dconv_proj_part,
                         float *dconv_kernel_part, float *dalign_out, const float *znorm, void *ws,
                         size_t ws_bytes, nabu_stream_t stream, unsigned *tickets, float *ds_out = nullptr,
                         float *cf_out = nullptr) {
  if (int e = check_attn(d)) return e;
  NABU_CHECK_ARG(dec_len && enc_len && keys && values && q && v && align && ctx && dctx && dq && dkeys && dv_part && ws,
                 "attn_bwd: null pointer");
  NABU_CHECK_ARG(d->kind != 1 || (conv_kernel && conv_proj && align_prev && dconv_proj_part &&
                              dconv_kernel_part && dalign_out),
                 "attn_bwd: location-aware attention needs its kernels and gradient buffers");
  if (ws_bytes < nabu_attn_bwd_ws_bytes(d)) return fail(NABU_EWS, "attn_bwd: workspace too small");
  const int S = attn_bwd_nslices(d);
  AttnArgs p = {};
  p.B = d->B; p.Te = d->Te; p.E = d->E; p.U = d->U; p.kind = d->kind; p.K = d->K; p.F = d->F; p.step = step;
  p.dec_len = dec_len; p.enc_len = enc_len; p.keys = keys; p.values = values; p.q = q; p.v = v;
  p.ck = conv_kernel; p.wf = conv_proj; p.align_prev = align_prev;
  p.align = const_cast<float *>(align);
  p.ctx = const_cast<float *>(ctx);
  p.dctx = dctx; p.dalign_in = dalign_in; p.dq = dq; p.dkeys = dkeys; p.dv_part = dv_part;
  p.dwf_part = dconv_proj_part; p.dck_part = dconv_kernel_part; p.dalign_out = dalign_out;
  p.prob_fn = d->prob_fn; p.znorm = const_cast<float *>(znorm);
  p.dq_part = static_cast<float *>(ws);
  p.dcf_g = p.dq_part + (size_t)d->B * S * d->U;
  p.tickets = d->kind != 1 ? tickets : nullptr;   // location-aware: the finish is a wide kernel of its own
  p.ds_out = ds_out; p.cf_out = cf_out;
  const bool defer = ds_out != nullptr;
  NABU_CHECK_ARG(d->prob_fn != 2 || znorm, "attn_bwd: normalized_sigmoid needs the normalisers of the forward pass");
  size_t shm = attn_lds(d, true);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool reg = d->kind == 1 && d->U <= 256 * RJ && d->F <= RF;
  auto kern = defer ? (d->kind != 1 ? attn_bwd_kernel<0, true> : reg ? attn_bwd_kernel<2, true> : attn_bwd_kernel<1, true>)
                    : (d->kind != 1 ? attn_bwd_kernel<0, false> : reg ? attn_bwd_kernel<2, false> : attn_bwd_kernel<1, false>);
  // the matrix-pipe kernel of the deferred location-aware chain wherever its geometry holds (attn_bwd_kernel<2, true>
  // stays the general kernel)
  if (defer && reg && cf_out && d->U % 16 == 0 && (d->Te + S - 1) / S <= 32 && d->Te <= 1024 && d->K * d->F <= 8 * AT &&
      d->E <= 16 * AT) {
    const int UT = (d->U / 16 + AT / 64 - 1) / (AT / 64);
    kern = UT <= 1 ? attn_bwd_loc_mfma_kernel<1> : UT == 2 ? attn_bwd_loc_mfma_kernel<2> : UT == 3 ? attn_bwd_loc_mfma_kernel<3>
                                                                                                   : attn_bwd_loc_mfma_kernel<4>;
    // its LDS: conv kernel, previous alignment, d scores, features, the waves' d-feature tiles [NW][2][16][16]
    const size_t need = ((((size_t)d->K * d->F + 3) & ~(size_t)3) + 2 * (size_t)d->Te + (size_t)d->Te * d->F + 4 + (AT / 64) * 512) * sizeof(float);
    if (need > shm) shm = need;
  }
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(kern, dim3(d->B, S), dim3(AT), shm, s, p);
  NABU_LAUNCH_CHECK();
#ifdef ATTN_STAMPS
  {
    static int calls = 0;
    if (++calls == 2000) {
      unsigned long long st[16];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn_stamps), sizeof(st));
      fprintf(stderr, "attn_bwd stamps (us):");
      for (int i = 1; i <= 6; ++i) fprintf(stderr, " %.2f", (double)(st[i] - st[i - 1]) / 100.0);
      fprintf(stderr, "  total %.2f\n", (double)(st[6] - st[0]) / 100.0);
    }
  }
#endif
  if (p.tickets) return 0;
  const size_t shm2 = ((size_t)d->Te + (d->kind == 1 ? (size_t)d->Te * d->F + (size_t)d->K * d->F : 0) + 4) * sizeof(float);
  if (shm2 > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(attn_bwd_finish_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2));
  hipLaunchKernelGGL(attn_bwd_finish_kernel, dim3(d->B, d->kind == 1 ? FPA + FPK : 1), dim3(d->kind == 1 ? FT : 256), shm2, s, p, S);
  NABU_LAUNCH_CHECK();
  return 0;
}
extern "C" int nabu_attn_bwd(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                             const int32_t *enc_len, const float *keys, const float *values,
                             const float *q, const float *v, const float *conv_kernel,
                             const float *conv_proj, const float *align_prev, const float *align,
                             const float *ctx, const float *dctx, const float *dalign_in, float *dq,
                             float *dkeys, float *dv_part, float *dconv_proj_part,
                             float *dconv_kernel_part, float *dalign_out, const float *znorm, void *ws,
                             size_t ws_bytes, nabu_stream_t stream) {
  return attn_bwd_impl(d, step, dec_len, enc_len, keys, values, q, v, conv_kernel, conv_proj, align_prev, align, ctx,
                       dctx, dalign_in, dq, dkeys, dv_part, dconv_proj_part, dconv_kernel_part, dalign_out, znorm, ws,
                       ws_bytes, stream, nullptr);
}

// attn_param_grads_kernel has its own frame partition: enough slices that a thread keeps at most 4 frames (more, smaller
// workgroups balance utterances of different lengths better), at most 16
static int attn_defer_slices(const nabu_attn_desc *d) {
  if (d->U % 4 || d->U / 4 > PT || PT % (d->U / 4)) return 0;
  if (d->kind == 1 && d->F > PNF) return 0;
  const int NG = PT / (d->U / 4);
  int S = (d->Te + 4 * NG - 1) / (4 * NG);
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  const int per = (d->Te + S - 1) / S;
  return (per + NG - 1) / NG <= 8 ? S : 0;
}
static int attn_param_grads(const nabu_attn_desc *d, int S, int L, const int32_t *dec_len, const int32_t *enc_len,
                            const float *keys, const float *q_all, const float *v, const float *wf, const float *ds_all,
                            const float *cf_all, float *dkeys, float *dv_part, float *dwf_part, hipStream_t s) {
  const int per = (d->Te + S - 1) / S, NG = PT / (d->U / 4);
  size_t shm = (PSB * (((size_t)per + 3) / 4 * 4 + ((size_t)per * (d->kind == 1 ? d->F : 0) + 3) / 4 * 4) + (size_t)NG * d->U + 4) * sizeof(float);
  const bool four = (per + NG - 1) / NG <= 4;
  auto kern = d->kind == 1 ? (four ? attn_param_grads_kernel<true, 4> : attn_param_grads_kernel<true, 8>)
                           : (four ? attn_param_grads_kernel<false, 4> : attn_param_grads_kernel<false, 8>);
  // location-aware, U <= 512, <= 32 frames per slice: the matrix-pipe kernel (the vector kernel stays the general one)
  if (d->kind == 1 && d->U % 16 == 0 && d->U <= 512 && d->F <= 12 && per <= 32) {
    kern = per <= 16 ? attn_param_grads_mfma_kernel<1> : attn_param_grads_mfma_kernel<2>;
    const size_t need = (size_t)PSB * (32 + 32 * d->F + 4) * sizeof(float);
    if (need > shm) shm = need;
  }
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(kern, dim3(d->B, S), dim3(PT), shm, s, d->B, d->Te, d->U, d->kind == 1 ? d->F : 0, L, dec_len, enc_len,
                     keys, q_all, v, wf, ds_all, cf_all, dkeys, dv_part, dwf_part);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_mask_time_f32(int B, int L, int F, float *x, const int32_t *len, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && L > 0 && F > 0 && x && len, "mask_time: bad argument");
  hipLaunchKernelGGL(mask_time_kernel, dim3(grid1((size_t)B * L * F)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), B, L, F, x, len);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_swap01_f32(int L, int B, int F, const float *x, float *y, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && L > 0 && F > 0 && x && y, "swap01: bad argument");
  hipLaunchKernelGGL(swap01_kernel, dim3(grid1((size_t)B * L * F)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), L, B, F, x, y);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_scatter_rows_f32(int C, int N, int W, const int32_t *ids, const float *dz, float *dK,
                                     nabu_stream_t stream) {
  NABU_CHECK_ARG(C > 0 && N > 0 && W > 0 && ids && dz && dK, "scatter_rows: bad argument");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((W + 255) / 256, C), dim3(256), 0,
                     static_cast<hipStream_t>(stream), C, N, W, ids, dz, dK);
  NABU_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// Whole-sequence decoder driver: the per-step launch sequence of
// RNNDecoder._decode runs here, in C++, so that a decoder step costs its kernel
// launches only (a Python/ctypes loop spent ~15 us of host time per launch).
namespace nabu {

// experiment / test switches, read at every call (a decoder call is milliseconds)
static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

struct SpLayout {
  size_t H[NABU_SPELLER_MAX_LAYERS], Cs[NABU_SPELLER_MAX_LAYERS], Ho[NABU_SPELLER_MAX_LAYERS],
      acts[NABU_SPELLER_MAX_LAYERS];
  size_t ctx, align, q, keys, logits_tm, ids, znorm, total;   // offsets in floats (ids: [L,B] int32)
  size_t dscale, sdraw;    // persistent decoder (speller_persist.h): dropout scale factors [L,B,U], sampling draws [L,B,2]
};

static SpLayout sp_layout(const nabu_speller_desc *d) {
  SpLayout s;
  const size_t B = d->B, L = d->L, U = d->U, E = d->E, Te = d->Te, C = d->C;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  for (int n = 0; n < d->num_layers; ++n) {
    s.H[n] = take((L + 1) * B * U);
    s.Cs[n] = take((L + 1) * B * U);
    s.Ho[n] = d->keep_prob < 1.f ? take((L + 1) * B * U) : s.H[n];
    s.acts[n] = take(L * B * 4 * U);
  }
  s.ctx = take((L + 1) * B * E);
  s.align = take((L + 1) * B * Te);
  s.q = take(L * B * U);
  s.keys = take(B * Te * U);
  s.logits_tm = take(L * B * C);
  s.ids = take(L * B);
  s.znorm = take(L * B);
  s.dscale = d->keep_prob < 1.f ? take(L * B * U) : 0;
  s.sdraw = d->sample_prob > 0.f ? take(2 * L * B) : 0;
  s.total = o;
  return s;
}

struct SpWs {
  size_t z, dl, dH, dCtx, dkeys, dv, dwf, dck, attn, dq, dz[NABU_SPELLER_MAX_LAYERS], dh[2][NABU_SPELLER_MAX_LAYERS],
      dc[2][NABU_SPELLER_MAX_LAYERS], dctx[2], dal[2], dx, tmp, gemm, gemm_bytes, total;
  size_t wqT, kxT[NABU_SPELLER_MAX_LAYERS], khT[NABU_SPELLER_MAX_LAYERS];   // transposed weights (backward)
  size_t kperm[NABU_SPELLER_MAX_LAYERS];   // gate-interleaved copies of the cell kernels' dense rows (forward)
  size_t kxhT, dxh[2];     // [4U, E+U] transposed rows of layer 0's kernel; [B, E+U] carries d(context | h) of a step
  size_t wq_sw, kxh_sw;    // the same two weights re-blocked for rows16_kernel (gemm_skinny.hip): [U, U], [E+U, 4U]
  size_t tickets, fpart;   // fused skinny products: per-column-slice tickets (zeroed per call), partial tiles
  size_t status, persist, persist_bytes;   // persistent decoder kernel: status word (ws[0]), XCC table + exchange rings
  size_t dv8;                              // its d attention_v partial rows [B*8, U]
  size_t dck8;                             // ... location-aware: conv kernel gradient partial rows [B*8, K*F]
  size_t ds_all, cf_all;                   // deferred attention gradients: d scores [L,B,Te], location features [L,B,Te,F]
  size_t dv16, dwf16;                      // ... and the partial rows of attn_param_grads_kernel [B*Sp, U], [B*Sp, F*U]
  // the decoder steps run as NS independent sub-batches on NS streams: per sub-batch slices of
  // the scratch that a step's kernels share
  int NS, S;               // sub-batches; attention-backward slices per utterance (of a sub-batch)
  size_t attn_each, gemm_each, fpart_each, z_each;
};

// The L decoder steps are a chain of small dependent kernels (each ~5 us of launch + memory latency,
// whatever its size).  Utterances are independent of each other until the weight gradients are summed, so
// the batch is cut into NS sub-batches whose chains run concurrently on NS streams (forked from / joined
// to the caller's stream by events); what one chain leaves idle the others use.  NABU_SPELLER_STREAMS=n
// overrides (1 = off).
static int sp_nsub(const nabu_speller_desc *d) {
  const int env = env_int("NABU_SPELLER_STREAMS", 0);
  int want = env > 0 ? env : 4;
  while (want > 1 && (d->B % want != 0 || d->B / want < (env > 0 ? 1 : 16))) want /= 2;
  return want < 1 ? 1 : want;
}

struct SubStreams {
  int n;
  hipStream_t st[8];
  hipEvent_t fork, done[8];
};
static int sub_streams(int n, hipStream_t main, SubStreams *out) {
  static thread_local hipStream_t side[8] = {nullptr};
  static thread_local hipEvent_t ev[9] = {nullptr};
  out->n = n;
  out->st[0] = main;
  for (int i = 1; i < n; ++i) {
    if (!side[i]) NABU_HIP(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    out->st[i] = side[i];
  }
  for (int i = 0; i <= n && i < 9; ++i)
    if (!ev[i]) NABU_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
  out->fork = ev[0];
  for (int i = 1; i < n; ++i) out->done[i] = ev[i];
  return 0;
}
static int sub_fork(const SubStreams &ss) {
  if (ss.n == 1) return 0;
  NABU_HIP(hipEventRecord(ss.fork, ss.st[0]));
  for (int i = 1; i < ss.n; ++i) NABU_HIP(hipStreamWaitEvent(ss.st[i], ss.fork, 0));
  return 0;
}
static int sub_join(const SubStreams &ss) {
  for (int i = 1; i < ss.n; ++i) {
    NABU_HIP(hipEventRecord(ss.done[i], ss.st[i]));
    NABU_HIP(hipStreamWaitEvent(ss.st[0], ss.done[i], 0));
  }
  return 0;
}
// Enqueue the sub-batch chains from one host thread each: a chain is thousands of launches, and ONE thread
// feeding four streams is about as fast as the GPU drains them (2.5 us per launch against ~10 us kernels, four at a
// time) — measured: with a single enqueuing thread every queue sat idle ~45% of the time waiting for its next
// step.  body(sub) enqueues ALL steps of one sub-batch on its stream and returns a NABU_E* / hipError_t code.
template <typename F>
static int run_subs(int NS, F body) {
  if (NS == 1) return body(0);
  const int threads_env = env_int("NABU_SPELLER_THREADS", 0);
  int codes[8] = {0};
  std::string texts[8];
  if (!threads_env) {
    int first = 0;                       // every chain is enqueued even after a failure: the caller joins the streams
    for (int i = 0; i < NS; ++i) {
      const int e = body(i);
      if (e && !first) first = e;
    }
    return first;
  }
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  std::thread th[8];
  for (int i = 1; i < NS; ++i)
    th[i] = std::thread([&, i]() {
      if (hipSetDevice(dev) != hipSuccess) { codes[i] = (int)hipErrorInvalidDevice; texts[i] = "hipSetDevice failed in a decoder enqueue thread"; return; }
      codes[i] = body(i);
      if (codes[i]) texts[i] = err_buf();        // the error text is thread-local: hand it to the caller's thread
    });
  codes[0] = body(0);
  for (int i = 1; i < NS; ++i) th[i].join();
  if (codes[0]) return codes[0];
  for (int i = 1; i < NS; ++i)
    if (codes[i]) return fail(codes[i], "%s", texts[i].c_str());
  return 0;
}

static nabu_attn_desc sub_attn_desc(const nabu_speller_desc *d, int Bn) {
  nabu_attn_desc a = {sizeof(nabu_attn_desc), Bn, d->Te, d->E, d->U, d->kind, d->K, d->F, d->prob_fn};
  return a;
}

static SpWs sp_ws(const nabu_speller_desc *d) {
  SpWs s;
  const size_t B = d->B, L = d->L, U = d->U, E = d->E, Te = d->Te, C = d->C, F = d->F, K = d->K;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  s.NS = sp_nsub(d);
  const size_t NS = s.NS, Bn = B / NS;
  // ws[0]: status word of the persistent decoder kernel (0 = ok; sticky, the caller provides the workspace
  // zero-initialised once, like the recurrent layers' workspace), then its XCC table and exchange rings
  s.status = take(64);
  {
    SpPersistDesc pd = {(int)B, (int)L, (int)U, (int)E, (int)Te, (int)C};
    pd.kind = d->kind; pd.K = d->K; pd.F = d->F;
    s.persist_bytes = speller_persist_ws_bytes(pd);
    if (speller_persist_bwd_ws_bytes(pd) > s.persist_bytes) s.persist_bytes = speller_persist_bwd_ws_bytes(pd);
    s.persist = take(s.persist_bytes / 4 + 4);
    s.dv8 = take(speller_persist_bwd_ws_bytes(pd) ? B * 8 * U : 0);
    s.dck8 = take((speller_persist_bwd_ws_bytes(pd) && d->kind == 1) ? B * 8 * K * F : 0);
    s.ds_all = take(L * B * Te);
    s.cf_all = take(d->kind == 1 ? L * B * Te * F : 0);
    s.dv16 = take(B * 16 * U);
    s.dwf16 = take(d->kind == 1 ? B * 16 * F * U : 0);
  }
  s.z_each = Bn * 4 * U;
  s.z = take(B * 4 * U);
  s.dl = take(L * B * C);
  s.dH = take(L * B * U);
  s.dCtx = take(L * B * E);
  s.dkeys = take(B * Te * U);
  const nabu_attn_desc adesc = sub_attn_desc(d, (int)Bn);
  const size_t S = attn_bwd_nslices(&adesc);
  s.S = (int)S;
  s.dv = take(B * S * U);
  s.dwf = take(B * S * F * U + 4);
  s.attn_each = ((nabu_attn_bwd_ws_bytes(&adesc) > nabu_attn_fwd_ws_bytes(&adesc) ? nabu_attn_bwd_ws_bytes(&adesc)
                                                                                    : nabu_attn_fwd_ws_bytes(&adesc)) / 4 + 4 + 3) / 4 * 4;
  s.attn = take(NS * s.attn_each);
  s.dck = take(B * K * F + 4);
  s.dq = take(L * B * U);
  for (int n = 0; n < d->num_layers; ++n) {
    s.dz[n] = take(L * B * 4 * U);
    for (int i = 0; i < 2; ++i) { s.dh[i][n] = take(B * U); s.dc[i][n] = take(B * U); }
  }
  for (int i = 0; i < 2; ++i) { s.dctx[i] = take(B * E); s.dal[i] = take(B * Te); }
  s.dx = take(B * U);
  s.tmp = take(B * U);
  s.wqT = take(U * U);
  for (int n = 0; n < d->num_layers; ++n) {
    s.kxT[n] = take(4 * U * (n == 0 ? E : U));
    s.khT[n] = take(4 * U * U);
  }
  for (int n = 0; n < d->num_layers; ++n) s.kperm[n] = take((n == 0 ? E + U : 2 * U) * 4 * U);
  s.kxhT = take(4 * U * (E + U));
  s.wq_sw = take(U * U);
  s.kxh_sw = take(4 * U * (E + U));
  for (int i = 0; i < 2; ++i) s.dxh[i] = take(B * (E + U));
  s.tickets = take(NS * 1024 + B + 4);   // + one counter per utterance for the attention launches
  {
    size_t kmax = E + U > 4 * U ? E + U : 4 * U, nmax = 4 * U > E ? 4 * U : E;
    s.fpart_each = ((kmax / 64 + 1) * Bn * nmax + 3) / 4 * 4;
    s.fpart = take(NS * s.fpart_each);
  }
  size_t g = 0;
  auto mx = [&](size_t v) { if (v > g) g = v; };
  const int BL = (int)(B * L), BT = (int)(B * Te);
  mx(nabu_gemm_ws_bytes((int)B, (int)(4 * U), (int)E)); mx(nabu_gemm_ws_bytes((int)B, (int)(4 * U), (int)U));
  mx(nabu_gemm_ws_bytes((int)B, (int)U, (int)U)); mx(nabu_gemm_ws_bytes((int)B, (int)E, (int)(4 * U)));
  mx(nabu_gemm_ws_bytes((int)B, (int)U, (int)(4 * U)));
  mx(nabu_gemm_ws_bytes((int)B, (int)C, (int)U)); mx(nabu_gemm_ws_bytes((int)B, (int)C, (int)E));
  mx(nabu_gemm_ws_bytes(BT, (int)U, (int)E)); mx(nabu_gemm_ws_bytes(BT, (int)E, (int)U));
  mx(nabu_gemm_ws_bytes((int)E, (int)U, BT));
  mx(nabu_gemm_ws_bytes(BL, (int)C, (int)U)); mx(nabu_gemm_ws_bytes(BL, (int)C, (int)E));
  mx(nabu_gemm_ws_bytes((int)U, (int)C, BL)); mx(nabu_gemm_ws_bytes((int)E, (int)C, BL));
  mx(nabu_gemm_ws_bytes(BL, (int)U, (int)C)); mx(nabu_gemm_ws_bytes(BL, (int)E, (int)C));
  mx(nabu_gemm_ws_bytes((int)U, (int)U, BL)); mx(nabu_gemm_ws_bytes((int)E, (int)(4 * U), BL));
  mx(nabu_gemm_ws_bytes((int)U, (int)(4 * U), BL)); mx(nabu_gemm_ws_bytes((int)Te, (int)E, (int)L));
  mx(nabu_colsum_ws_bytes(BL, (int)(4 * U))); mx(nabu_colsum_ws_bytes((int)(B * 16), (int)(F * U + K * F + U)));
  s.gemm_bytes = (g + 255) / 256 * 256;
  s.gemm_each = s.gemm_bytes / 4 + 4;
  s.gemm = take(NS * s.gemm_each);
  s.total = o;
  return s;
}

static int check_sp(const nabu_speller_desc *d) {
  if (!d || d->size != sizeof(nabu_speller_desc)) return fail(NABU_EINVAL, "speller: bad descriptor size");
  if (d->B <= 0 || d->Te <= 0 || d->E <= 0 || d->U <= 0 || d->C <= 1 || d->L <= 0)
    return fail(NABU_EINVAL, "speller: bad dimensions");
  if (d->num_layers < 1 || d->num_layers > NABU_SPELLER_MAX_LAYERS) return fail(NABU_EUNSUP, "speller: 1..%d layers", NABU_SPELLER_MAX_LAYERS);
  if (d->U % 4 || d->E % 4) return fail(NABU_EUNSUP, "speller: num_units and encoder dim must be multiples of 4");
  if (!(d->keep_prob > 0.f && d->keep_prob <= 1.f)) return fail(NABU_EINVAL, "speller: keep_prob out of (0,1]");
  if (!(d->sample_prob >= 0.f && d->sample_prob <= 1.f)) return fail(NABU_EINVAL, "speller: sample_prob out of [0,1]");
  nabu_attn_desc a = {sizeof(nabu_attn_desc), d->B, d->Te, d->E, d->U, d->kind, d->K, d->F, d->prob_fn};
  return check_attn(&a);
}

// C = A·B (+ beta*C) on row-major contiguous operands
static int mm(bool ta, bool tb, int M, int N, int K, const float *A, int lda, const float *Bm, int ldb,
              float beta, float *C, int ldc, const float *bias, float *ws, size_t wsb, nabu_stream_t st) {
  return nabu_gemm_f32(ta, tb, M, N, K, 1.f, A, lda, Bm, ldb, beta, C, ldc, bias, 0, 0, 0, ws, wsb, st);
}
#define SP_TRY(call) do { int e_ = (call); if (e_) return e_; } while (0)

// C[M,N] = A·B + A2·B2 (+ beta*C): ONE launch with the split-K reduction inside it when the shape
// allows (gemm_skinny.hip), else two plain products
static bool fused_ok(int M, int N, int K1, int lda, int K2, int lda2) {
  return env_int("NABU_SPELLER_FUSED", 1) && M <= 64 && N % 32 == 0 && K1 > 0 && K1 % 64 == 0 && K2 % 64 == 0 && lda % 4 == 0 && (K2 == 0 || lda2 % 4 == 0);
}
static int mm2(int M, int N, int K1, const float *A, int lda, const float *Bm, int ldb, int K2, const float *A2, int lda2,
               const float *B2, int ldb2, float beta, float *C, int ldc, float *w, const SpWs &W, int sub, float *gw,
               size_t gwb, nabu_stream_t st) {
  if (fused_ok(M, N, K1, lda, K2, lda2))
    return gemm_skinny_fused(M, N, K1, A, lda, Bm, ldb, K2, A2, lda2, B2, ldb2, beta, C, ldc, nullptr,
                             w + W.fpart + (size_t)sub * W.fpart_each,
                             reinterpret_cast<unsigned *>(w + W.tickets) + (size_t)sub * 1024, static_cast<hipStream_t>(st));
  if (int e = mm(false, false, M, N, K1, A, lda, Bm, ldb, beta, C, ldc, nullptr, gw, gwb, st)) return e;
  if (K2 > 0) return mm(false, false, M, N, K2, A2, lda2, B2, ldb2, 1.f, C, ldc, nullptr, gw, gwb, st);
  return 0;
}

}  // namespace nabu

extern "C" size_t nabu_speller_reserve_bytes(const nabu_speller_desc *d) {
  if (check_sp(d)) return 0;
  return sp_layout(d).total * sizeof(float);
}
extern "C" int nabu_speller_decoder_inputs(const nabu_speller_desc *d, const void *reserve, int32_t *out_ids,
                                           nabu_stream_t stream) {
  if (int e = check_sp(d)) return e;
  NABU_CHECK_ARG(reserve && out_ids, "speller_decoder_inputs: null pointer");
  const SpLayout R = sp_layout(d);
  NABU_HIP(hipMemcpyAsync(out_ids, static_cast<const float *>(reserve) + R.ids, (size_t)d->L * d->B * 4,
                          hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return 0;
}
extern "C" size_t nabu_speller_ws_bytes(const nabu_speller_desc *d) {
  if (check_sp(d)) return 0;
  return sp_ws(d).total * sizeof(float);
}

// Which decoder steps run as ONE persistent launch (speller_persist.hip) instead of the step chain; the same
// predicates drive nabu_speller_fwd / _bwd and the query nabu_speller_uses_persistent.
static bool fwd_takes_persistent(const nabu_speller_desc *d, const SpWs &W) {
  const int B = d->B, U = d->U, E = d->E, Bn = B / W.NS;
  const bool cell_epi0 = env_int("NABU_SPELLER_EPILOGUE", 1) && fused_ok(Bn, 4 * U, E, E, U, U);
  SpPersistDesc pd = {B, d->L, U, E, d->Te, d->C};
  pd.kind = d->kind; pd.K = d->K; pd.F = d->F;
  pd.sample_prob = d->sample_prob;
  if (!(d->num_layers == 1 && (d->kind == 0 || d->kind == 1) && d->prob_fn == 0 && cell_epi0 && W.persist_bytes > 0 &&
        speller_persist_ok(pd)))
    return false;
  // Location-aware attention, more than one launch of 32 utterances, values streamed from L2 (cfg5's geometry): the step
  // chain on sub-batches of 16 with its round-5 kernels (rows16_kernel, attn_fwd_loc_mfma_kernel) is faster than two
  // persistent launches (cfg5: 41.0 against 42.7 ms per training step).  NABU_SPELLER_PERSIST=2: the persistent kernel anyway.
  const char *env = getenv("NABU_SPELLER_PERSIST");
  const bool chain_fast = d->kind == 1 && B > 32 && Bn <= 64 && E % 16 == 0 && (d->sample_prob == 0.f || sample_step_ok(d->C)) && rows16_ok(Bn, 4 * U, E + U, E) &&
                          rows16_ok(Bn, U, U, U) && env_int("NABU_SPELLER_ROWS16", 1) && speller_persist_streams_values(pd);
  return !(chain_fast && !(env && atoi(env) == 2));
}
static bool bwd_takes_persistent(const nabu_speller_desc *d, const SpWs &W) {
  const int B = d->B, U = d->U, E = d->E, Bn = B / W.NS;
  const bool fuse_shapes = env_int("NABU_SPELLER_EPILOGUE", 1) && d->num_layers == 1 && fused_ok(Bn, U, U, U, 0, 0) &&
                           fused_ok(Bn, E + U, 4 * U, 4 * U, 0, 0) && (E + U) / 32 <= 1024;
  SpPersistDesc pd = {B, d->L, U, E, d->Te, d->C};
  pd.kind = d->kind; pd.K = d->K; pd.F = d->F;
  // (location-aware attention: the kernel leaves d keys / d attention_v / d conv_proj to attn_param_grads_kernel)
  if (d->kind == 1) {
    const nabu_attn_desc adb = sub_attn_desc(d, B);
    if (!env_int("NABU_SPELLER_DEFER", 1) || attn_defer_slices(&adb) <= 0) return false;
  }
  return fuse_shapes && (d->kind == 0 || d->kind == 1) && d->prob_fn == 0 && W.persist_bytes > 0 && speller_persist_bwd_ok(pd);
}
extern "C" int nabu_speller_uses_persistent(const nabu_speller_desc *d, int backward) {
  if (check_sp(d)) return 0;
  const SpWs W = sp_ws(d);
  return (backward ? bwd_takes_persistent(d, W) : fwd_takes_persistent(d, W)) ? 1 : 0;
}

extern "C" int nabu_speller_fwd(const nabu_speller_desc *d, const float *values, const int32_t *enc_len,
                                const int32_t *ids, const int32_t *dec_len, const nabu_speller_params *p,
                                float *logits, void *reserve, void *ws, size_t ws_bytes,
                                nabu_stream_t stream) {
  if (int e = check_sp(d)) return e;
  NABU_CHECK_ARG(values && enc_len && ids && dec_len && p && logits && reserve && ws, "speller_fwd: null pointer");
  const SpLayout R = sp_layout(d);
  const SpWs W = sp_ws(d);
  if (ws_bytes < W.total * sizeof(float)) return fail(NABU_EWS, "speller_fwd: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *r = static_cast<float *>(reserve), *w = static_cast<float *>(ws);
  const int B = d->B, L = d->L, U = d->U, E = d->E, Te = d->Te, C = d->C, nl = d->num_layers;
  float *gw = w + W.gemm;
  const size_t gwb = W.gemm_bytes;
  const bool drop = d->keep_prob < 1.f;
  NABU_HIP(hipMemsetAsync(w + W.tickets, 0, ((size_t)W.NS * 1024 + B + 4) * 4, s));
  // zero initial state (index 0 of every time-major array)
  for (int n = 0; n < nl; ++n) {
    NABU_HIP(hipMemsetAsync(r + R.H[n], 0, (size_t)B * U * 4, s));
    NABU_HIP(hipMemsetAsync(r + R.Cs[n], 0, (size_t)B * U * 4, s));
    if (drop) NABU_HIP(hipMemsetAsync(r + R.Ho[n], 0, (size_t)B * U * 4, s));
  }
  NABU_HIP(hipMemsetAsync(r + R.ctx, 0, (size_t)B * E * 4, s));
  NABU_HIP(hipMemsetAsync(r + R.align, 0, (size_t)B * Te * 4, s));
  if (d->kind == 2) SP_TRY(first_col_one(B, Te, r + R.align, s));
  // decoder inputs actually used (scheduled sampling replaces entries of rows 1..L-1 below)
  int32_t *ids_used = reinterpret_cast<int32_t *>(r + R.ids);
  NABU_HIP(hipMemcpyAsync(ids_used, ids, (size_t)L * B * 4, hipMemcpyDeviceToDevice, s));
  const bool sampling = d->sample_prob > 0.f;
  // keys = memory_layer(values)
  SP_TRY(mm(false, false, B * Te, U, E, values, E, p->memory_kernel, U, 0.f, r + R.keys, U, nullptr, gw, gwb, stream));
  const int NS = W.NS, Bn = B / NS;
  const nabu_attn_desc adn = sub_attn_desc(d, Bn);
  const size_t attn_fwd_wsb_n = nabu_attn_fwd_ws_bytes(&adn);
  // LSTM cell folded into the step product's last workgroup (gemm_skinny.hip) when the shapes allow:
  // the product then runs against gate-interleaved copies of the kernels' dense rows
  const int epi_env = env_int("NABU_SPELLER_EPILOGUE", 1);
  bool cell_epi[NABU_SPELLER_MAX_LAYERS];
  for (int n = 0; n < nl; ++n) {
    const int K1 = n == 0 ? E : U;
    cell_epi[n] = epi_env && fused_ok(Bn, 4 * U, K1, K1, U, U);
    if (cell_epi[n]) {
      const int rows = K1 + U;
      const float *src = p->lstm_kernel[n] + (n == 0 ? (size_t)C * 4 * U : 0);
      hipLaunchKernelGGL(permute_gates_kernel, dim3(grid1((size_t)rows * 4 * U)), dim3(256), 0, s, rows, U, src,
                         w + W.kperm[n]);
      NABU_LAUNCH_CHECK();
    }
  }
  // the whole step loop as ONE persistent launch (speller_persist.hip) where the geometry allows:
  // one LSTM layer, vanilla softmax attention, teacher forcing, no dropout, B = 32 (cfg3)
  SpPersistDesc pd = {B, L, U, E, Te, C};
  pd.kind = d->kind; pd.K = d->K; pd.F = d->F;
  pd.keep_prob = d->keep_prob; pd.seed = d->seed; pd.seed_offset = d->seed_offset;     // nl == 1: offset + t*nl + n = offset + t
  pd.sample_prob = d->sample_prob; pd.sample_seed = d->sample_seed; pd.sample_offset = d->sample_offset;
  pd.drop_scale = drop ? r + R.dscale : nullptr;
  pd.sample_draws = d->sample_prob > 0.f ? reinterpret_cast<unsigned *>(r + R.sdraw) : nullptr;
  const bool persist = fwd_takes_persistent(d, W);
  if (persist)
    SP_TRY(speller_persist_fwd(pd, dec_len, enc_len, ids_used, w + W.kperm[0], p->lstm_bias[0], p->lstm_kernel[0],
                               p->query_kernel, p->attention_v, r + R.keys, values, p->conv_kernel, p->conv_proj, r + R.H[0],
                               drop ? r + R.Ho[0] : nullptr, r + R.Cs[0], r + R.acts[0], r + R.q, r + R.ctx, r + R.align,
                               reinterpret_cast<int *>(w + W.status), w + W.persist, W.persist_bytes, s, p->out_kernel,
                               p->out_bias, ids_used));
  unsigned *atk = env_int("NABU_SPELLER_ATTN_FUSED", 1) ? reinterpret_cast<unsigned *>(w + W.tickets) + (size_t)NS * 1024 : nullptr;
  SubStreams ss;
  // sub-batches of <= 16 utterances: the cell's product ([context | h] . kernel with the cell as epilogue) and the query
  // by rows16_kernel (gemm_skinny.hip) over weights re-blocked once per pass; NABU_SPELLER_ROWS16=0: gemm_skinny_fused
  const bool r16 = !persist && nl == 1 && cell_epi[0] && env_int("NABU_SPELLER_ROWS16", 1) && E % 16 == 0 &&
                   rows16_ok(Bn, 4 * U, E + U, E) && rows16_ok(Bn, U, U, U);
  if (r16) {
    SP_TRY(rows16_swizzle_kn(4 * U, E + U, p->lstm_kernel[0] + (size_t)C * 4 * U, 4 * U, w + W.kxh_sw, U, s));
    SP_TRY(rows16_swizzle_kn(U, U, p->query_kernel, U, w + W.wq_sw, 0, s));
  }
  if (!persist) {
  SP_TRY(sub_streams(NS, s, &ss));
  SP_TRY(sub_fork(ss));
  }
  auto fwd_chain = [&](int sub) -> int {
    for (int t = 0; t < L; ++t) {
      const int b0 = sub * Bn;
      nabu_stream_t st = static_cast<nabu_stream_t>(ss.st[sub]);
      float *gws = gw + (size_t)sub * W.gemm_each;
      float *z = w + W.z + (size_t)b0 * 4 * U;
      const int32_t *dlen = dec_len + b0;
      for (int n = 0; n < nl; ++n) {
        const float *Kn = p->lstm_kernel[n];
        float *Hn = r + R.H[n] + (size_t)b0 * U, *Cn = r + R.Cs[n] + (size_t)b0 * U;
        const size_t cur = (size_t)t * B * U, nxt = (size_t)(t + 1) * B * U;
        if (cell_epi[n]) {   // product + cell in one launch
          SkinnyEpilogue ep = {};
          ep.kind = 1; ep.U = U; ep.step = t; ep.seq_len = dlen;
          ep.bias = p->lstm_bias[n];
          ep.emb = n == 0 ? Kn : nullptr;
          ep.ids = n == 0 ? ids_used + (size_t)t * B + b0 : nullptr;
          ep.c_prev = Cn + cur; ep.h_prev = Hn + cur;
          ep.acts = r + R.acts[n] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U;
          ep.c_new = Cn + nxt; ep.h_new = Hn + nxt;
          const int K1 = n == 0 ? E : U;
          const float *x1 = n == 0 ? r + R.ctx + (size_t)t * B * E + (size_t)b0 * E : r + R.Ho[n - 1] + nxt + (size_t)b0 * U;
          const float *Kp = w + W.kperm[n];
          if (r16) {
            if (drop) {      // the cell's output dropout in the same launch (the mask of dropout_rows below)
              ep.ho_new = r + R.Ho[n] + nxt + (size_t)b0 * U;
              ep.keep = d->keep_prob; ep.seed = d->seed; ep.seed_offset = d->seed_offset + (unsigned long long)t * nl + n;
              ep.row0 = b0;
            }
            SP_TRY(rows16(Bn, 4 * U, E + U, x1, E, w + W.kxh_sw, 0.f, nullptr, 0, ss.st[sub], &ep, nullptr, E, Hn + cur, U));
          } else
          SP_TRY(gemm_skinny_fused(Bn, 4 * U, K1, x1, K1, Kp, 4 * U, U, Hn + cur, U, Kp + (size_t)K1 * 4 * U, 4 * U, 0.f, z,
                                   4 * U, nullptr, w + W.fpart + (size_t)sub * W.fpart_each,
                                   reinterpret_cast<unsigned *>(w + W.tickets) + (size_t)sub * 1024, ss.st[sub], &ep));
        } else if (n == 0) {
          SP_TRY(mm2(Bn, 4 * U, E, r + R.ctx + (size_t)t * B * E + (size_t)b0 * E, E, Kn + (size_t)C * 4 * U, 4 * U, U,
                     Hn + cur, U, Kn + (size_t)(C + E) * 4 * U, 4 * U, 0.f, z, 4 * U, w, W, sub, gws, gwb, st));
          SP_TRY(nabu_lstm_cell_fwd(Bn, U, t, dlen, z, p->lstm_bias[0], Kn, ids_used + (size_t)t * B + b0, Cn + cur,
                                    Hn + cur, r + R.acts[0] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U, Cn + nxt,
                                    Hn + nxt, st));
        } else {
          SP_TRY(mm2(Bn, 4 * U, U, r + R.Ho[n - 1] + nxt + (size_t)b0 * U, U, Kn, 4 * U, U, Hn + cur, U,
                     Kn + (size_t)U * 4 * U, 4 * U, 0.f, z, 4 * U, w, W, sub, gws, gwb, st));
          SP_TRY(nabu_lstm_cell_fwd(Bn, U, t, dlen, z, p->lstm_bias[n], nullptr, nullptr, Cn + cur, Hn + cur,
                                    r + R.acts[n] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U, Cn + nxt, Hn + nxt, st));
        }
        if (drop && !(r16 && cell_epi[n]))
          SP_TRY(dropout_rows((size_t)Bn * U, Hn + nxt, r + R.Ho[n] + nxt + (size_t)b0 * U, d->keep_prob, d->seed,
                              d->seed_offset + (unsigned long long)t * nl + n, (size_t)b0 * U, ss.st[sub]));
      }
      const float *htop = r + R.Ho[nl - 1] + (size_t)(t + 1) * B * U + (size_t)b0 * U;
      float *qt = r + R.q + (size_t)t * B * U + (size_t)b0 * U;
      if (r16) SP_TRY(rows16(Bn, U, U, htop, U, w + W.wq_sw, 0.f, qt, U, ss.st[sub]));
      else
      SP_TRY(mm2(Bn, U, U, htop, U, p->query_kernel, U, 0, nullptr, 0, nullptr, 0, 0.f, qt, U, w, W, sub, gws, gwb, st));
      SP_TRY(attn_fwd_impl(&adn, t, dlen, enc_len + b0, r + R.keys + (size_t)b0 * Te * U, values + (size_t)b0 * Te * E, qt,
                           p->attention_v, p->conv_kernel, p->conv_proj,
                           r + R.align + (size_t)t * B * Te + (size_t)b0 * Te, r + R.ctx + (size_t)t * B * E + (size_t)b0 * E,
                           r + R.align + (size_t)(t + 1) * B * Te + (size_t)b0 * Te,
                           r + R.ctx + (size_t)(t + 1) * B * E + (size_t)b0 * E, r + R.znorm + (size_t)t * B + b0,
                           w + W.attn + (size_t)sub * W.attn_each, attn_fwd_wsb_n, st, atk ? atk + b0 : nullptr));
      if (sampling && t + 1 < L) {
        // ScheduledEmbeddingTrainingHelper: the step's logits decide the next input of selected rows
        float *lt = r + R.logits_tm + (size_t)t * B * C + (size_t)b0 * C;
        if (sample_step_ok(C)) {     // one launch, logits only for sampled rows
          SP_TRY(sample_step(Bn, C, U, E, htop, U, r + R.ctx + (size_t)(t + 1) * B * E + (size_t)b0 * E, E, p->out_kernel,
                             p->out_bias, d->sample_prob, d->sample_seed, d->sample_offset + (unsigned long long)t,
                             ids + (size_t)(t + 1) * B + b0, ids_used + (size_t)(t + 1) * B + b0, b0, ss.st[sub]));
          continue;
        }
        SP_TRY(mm(false, false, Bn, C, U, htop, U, p->out_kernel, C, 0.f, lt, C, p->out_bias, gws, gwb, st));
        SP_TRY(mm(false, false, Bn, C, E, r + R.ctx + (size_t)(t + 1) * B * E + (size_t)b0 * E, E,
                  p->out_kernel + (size_t)U * C, C, 1.f, lt, C, nullptr, gws, gwb, st));
        SP_TRY(sample_ids_rows(Bn, C, lt, d->sample_prob, d->sample_seed, d->sample_offset + (unsigned long long)t,
                               ids + (size_t)(t + 1) * B + b0, ids_used + (size_t)(t + 1) * B + b0, b0, ss.st[sub]));
      }
    }
    return 0;
  };
  if (!persist) {
  {  // join the side streams before an error is propagated: chains already enqueued must not outlive the call
    const int e_run = run_subs(NS, fwd_chain), e_join = sub_join(ss);
    if (e_run) return e_run;
    SP_TRY(e_join);
  }
  }
  // output projection of all steps: [h_t, ctx_t]·W + b, then batch-major + impute_finished
  float *ltm = r + R.logits_tm;
  SP_TRY(mm(false, false, L * B, C, U, r + R.Ho[nl - 1] + (size_t)B * U, U, p->out_kernel, C, 0.f, ltm, C, p->out_bias, gw, gwb, stream));
  SP_TRY(mm(false, false, L * B, C, E, r + R.ctx + (size_t)B * E, E, p->out_kernel + (size_t)U * C, C, 1.f, ltm, C, nullptr, gw, gwb, stream));
  SP_TRY(nabu_swap01_f32(L, B, C, ltm, logits, stream));
  SP_TRY(nabu_mask_time_f32(B, L, C, logits, dec_len, stream));
  (void)s;
  return 0;
}

extern "C" int nabu_speller_bwd(const nabu_speller_desc *d, const float *values, const int32_t *enc_len,
                                const int32_t *ids, const int32_t *dec_len, const nabu_speller_params *p,
                                const float *dlogits, void *reserve, const nabu_speller_grads *g,
                                float *dvalues, void *ws, size_t ws_bytes, nabu_stream_t stream) {
  if (int e = check_sp(d)) return e;
  NABU_CHECK_ARG(values && enc_len && ids && dec_len && p && dlogits && reserve && g && dvalues && ws,
                 "speller_bwd: null pointer");
  const SpLayout R = sp_layout(d);
  const SpWs W = sp_ws(d);
  if (ws_bytes < W.total * sizeof(float)) return fail(NABU_EWS, "speller_bwd: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *r = static_cast<float *>(reserve), *w = static_cast<float *>(ws);
  const int B = d->B, L = d->L, U = d->U, E = d->E, Te = d->Te, C = d->C, nl = d->num_layers, F = d->F, K = d->K;
  float *gw = w + W.gemm;
  const size_t gwb = W.gemm_bytes;
  const bool drop = d->keep_prob < 1.f;
  const int BL = B * L;
  float *dl = w + W.dl, *dH = w + W.dH, *dCtx = w + W.dCtx, *dkeys = w + W.dkeys, *dq = w + W.dq;
  const float *htop_all = r + R.Ho[nl - 1] + (size_t)B * U;   // h_top[t], t = 0..L-1
  const float *ctx1 = r + R.ctx + (size_t)B * E;              // ctx[t]
  // output projection
  SP_TRY(nabu_swap01_f32(B, L, C, dlogits, dl, stream));      // [B,L,C] -> [L,B,C]
  SP_TRY(mm(true, false, U, C, BL, htop_all, U, dl, C, 0.f, g->out_kernel, C, nullptr, gw, gwb, stream));
  SP_TRY(mm(true, false, E, C, BL, ctx1, E, dl, C, 0.f, g->out_kernel + (size_t)U * C, C, nullptr, gw, gwb, stream));
  SP_TRY(nabu_colsum_f32(BL, C, dl, C, 0.f, g->out_bias, gw, gwb, stream));
  SP_TRY(mm(false, true, BL, U, C, dl, C, p->out_kernel, C, 0.f, dH, U, nullptr, gw, gwb, stream));
  SP_TRY(mm(false, true, BL, E, C, dl, C, p->out_kernel + (size_t)U * C, C, 0.f, dCtx, E, nullptr, gw, gwb, stream));
  NABU_HIP(hipMemsetAsync(dkeys, 0, (size_t)B * Te * U * 4, s));
  NABU_HIP(hipMemsetAsync(w + W.tickets, 0, ((size_t)W.NS * 1024 + B + 4) * 4, s));
  const int S = W.S;          // per-slice partial rows of the attention backward (slices of a sub-batch's utterances)
  NABU_HIP(hipMemsetAsync(w + W.dv, 0, (size_t)B * S * U * 4, s));
  if (d->kind == 1) {
    NABU_HIP(hipMemsetAsync(w + W.dwf, 0, (size_t)B * S * F * U * 4, s));
    NABU_HIP(hipMemsetAsync(w + W.dck, 0, (size_t)B * K * F * 4, s));
  }
  for (int n = 0; n < nl; ++n) {
    NABU_HIP(hipMemsetAsync(w + W.dh[0][n], 0, (size_t)B * U * 4, s));
    NABU_HIP(hipMemsetAsync(w + W.dc[0][n], 0, (size_t)B * U * 4, s));
  }
  // transposed copies of the weights the per-step gradient products use: dz·W^T becomes a
  // row-major product with M = B rows, which the skinny GEMM kernel streams in a few microseconds
  SP_TRY(transpose(U, U, p->query_kernel, U, w + W.wqT, s));
  for (int n = 0; n < nl; ++n) {
    const float *Kn = p->lstm_kernel[n];
    if (n == 0) {
      SP_TRY(transpose(E, 4 * U, Kn + (size_t)C * 4 * U, 4 * U, w + W.kxT[0], s));
      SP_TRY(transpose(U, 4 * U, Kn + (size_t)(C + E) * 4 * U, 4 * U, w + W.khT[0], s));
    } else {
      SP_TRY(transpose(U, 4 * U, Kn, 4 * U, w + W.kxT[n], s));
      SP_TRY(transpose(U, 4 * U, Kn + (size_t)U * 4 * U, 4 * U, w + W.khT[n], s));
    }
  }
  const int NS = W.NS, Bn = B / NS;
  const nabu_attn_desc adn = sub_attn_desc(d, Bn);
  const size_t attn_wsb_n = nabu_attn_bwd_ws_bytes(&adn);
  SubStreams ss;
  // single-layer decoder without dropout (the BASELINE recipes): the cell's backward pass is folded into the
  // last workgroup of dq·Wq^T, and dz·[Kx^T | Kh^T] is ONE product whose [B, E+U] result carries d context and
  // d h to the next step — 4 dependent launches per step instead of 7
  const int epi_env_b = env_int("NABU_SPELLER_EPILOGUE", 1);
  const bool fuse_shapes = epi_env_b && nl == 1 && fused_ok(Bn, U, U, U, 0, 0) && fused_ok(Bn, E + U, 4 * U, 4 * U, 0, 0) &&
                           (E + U) / 32 <= 1024;
  // the whole step loop as ONE persistent launch (speller_persist.hip), as in the forward pass (output dropout is
  // applied inside it: the scale factors are drawn again from the Philox stream by a small launch in front of it)
  SpPersistDesc pd = {B, L, U, E, Te, C};
  pd.kind = d->kind; pd.K = d->K; pd.F = d->F;
  pd.keep_prob = d->keep_prob; pd.seed = d->seed; pd.seed_offset = d->seed_offset;
  pd.drop_scale = drop ? const_cast<float *>(r + R.dscale) : nullptr;
  const bool persist = bwd_takes_persistent(d, W);
  // sub-batches of <= 16 utterances: both products of a step by rows16_kernel (no split-K hand-off between workgroups:
  // 13 -> 7 us per launch; its cell epilogue applies the output dropout's mask); NABU_SPELLER_ROWS16=0: gemm_skinny_fused
  const bool r16 = fuse_shapes && !persist && E % 32 == 0 && env_int("NABU_SPELLER_SPLIT", 1) && env_int("NABU_SPELLER_ROWS16", 1) &&
                   rows16_ok(Bn, U, U, U) && rows16_ok(Bn, E + U, 4 * U, 4 * U);
  const bool fuse_b = fuse_shapes && (!drop || persist || r16);      // (gemm_skinny_fused's cell epilogue has no dropout)
  if (fuse_b) SP_TRY(transpose(E + U, 4 * U, p->lstm_kernel[0] + (size_t)C * 4 * U, 4 * U, w + W.kxhT, s));
  const bool split_b = fuse_b && E % 32 == 0 && env_int("NABU_SPELLER_SPLIT", 1);
  if (r16) {
    SP_TRY(rows16_swizzle(U, U, p->query_kernel, U, w + W.wq_sw, s));
    SP_TRY(rows16_swizzle(E + U, 4 * U, p->lstm_kernel[0] + (size_t)C * 4 * U, 4 * U, w + W.kxh_sw, s));
  }
  if (persist)
    SP_TRY(speller_persist_bwd(pd, dec_len, enc_len, w + W.kxhT, p->query_kernel, p->attention_v, r + R.keys, values,
                               r + R.acts[0], r + R.Cs[0], r + R.q, r + R.ctx, r + R.align, dH, dCtx, dq, w + W.dz[0],
                               dkeys, w + W.dv8, reinterpret_cast<int *>(w + W.status), w + W.persist, W.persist_bytes, s,
                               p->conv_kernel, p->conv_proj, w + W.ds_all, d->kind == 1 ? w + W.cf_all : nullptr,
                               d->kind == 1 ? w + W.dck8 : nullptr));
  if (!persist) {
    SP_TRY(sub_streams(NS, s, &ss));
    SP_TRY(sub_fork(ss));
  }
  // d keys / d attention_v / d conv_proj of all steps in ONE launch after the chain (attn_param_grads_kernel)
  nabu_attn_desc adb = adn;      // the whole batch
  adb.B = B;
  // (the persistent kernel accumulates them itself for vanilla attention and leaves them to that launch for
  // location-aware attention)
  const int Sp = ((!persist || d->kind == 1) && env_int("NABU_SPELLER_DEFER", 1)) ? attn_defer_slices(&adb) : 0;
  const bool defer = Sp > 0;
  unsigned *atk = env_int("NABU_SPELLER_ATTN_FUSED", 1) ? reinterpret_cast<unsigned *>(w + W.tickets) + (size_t)NS * 1024 : nullptr;
  auto bwd_chain = [&](int sub) -> int {
  int cur = 0;   // index of the carries coming from step t+1
  bool have_carry = false;
  for (int t = L - 1; t >= 0; --t) {
    {
      const int b0 = sub * Bn;
      nabu_stream_t st = static_cast<nabu_stream_t>(ss.st[sub]);
      float *gws = gw + (size_t)sub * W.gemm_each;
      const int32_t *dlen = dec_len + b0;
      float *dCt = dCtx + (size_t)t * B * E + (size_t)b0 * E;
      if (have_carry && fuse_b && !split_b) {
        hipLaunchKernelGGL(add_rows_kernel, dim3(grid1((size_t)Bn * E)), dim3(256), 0, ss.st[sub], Bn, E,
                           w + W.dxh[(t + 1) & 1] + (size_t)b0 * (E + U), E + U, dCt, E);
        NABU_LAUNCH_CHECK();
      } else if (have_carry && !fuse_b) {
        SP_TRY(nabu_axpy_f32((size_t)Bn * E, 1.f, w + W.dctx[(t + 1) & 1] + (size_t)b0 * E, dCt, st));
      }
      float *dal_out = d->kind == 1 ? w + W.dal[t & 1] + (size_t)b0 * Te : nullptr;
      const float *dal_carry = (d->kind == 1 && have_carry) ? w + W.dal[(t + 1) & 1] + (size_t)b0 * Te : nullptr;
      float *dqt = dq + (size_t)t * B * U + (size_t)b0 * U;
      SP_TRY(attn_bwd_impl(&adn, t, dlen, enc_len + b0, r + R.keys + (size_t)b0 * Te * U, values + (size_t)b0 * Te * E,
                           r + R.q + (size_t)t * B * U + (size_t)b0 * U, p->attention_v, p->conv_kernel, p->conv_proj,
                           r + R.align + (size_t)t * B * Te + (size_t)b0 * Te,
                           r + R.align + (size_t)(t + 1) * B * Te + (size_t)b0 * Te,
                           r + R.ctx + (size_t)(t + 1) * B * E + (size_t)b0 * E, dCt, dal_carry, dqt,
                           dkeys + (size_t)b0 * Te * U, w + W.dv + (size_t)b0 * S * U,
                           d->kind == 1 ? w + W.dwf + (size_t)b0 * S * F * U : nullptr,
                           d->kind == 1 ? w + W.dck + (size_t)b0 * K * F : nullptr, dal_out,
                           r + R.znorm + (size_t)t * B + b0, w + W.attn + (size_t)sub * W.attn_each, attn_wsb_n, st,
                           atk ? atk + b0 : nullptr,
                           defer ? w + W.ds_all + ((size_t)t * B + b0) * Te : nullptr,
                           (defer && d->kind == 1) ? w + W.cf_all + ((size_t)t * B + b0) * Te * F : nullptr));
      float *dHt = dH + (size_t)t * B * U + (size_t)b0 * U;
      if (fuse_b) {
        float *dzt = w + W.dz[0] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U;
        const float *Cn = r + R.Cs[0] + (size_t)b0 * U;
        SkinnyEpilogue ep = {};
        ep.kind = 2; ep.U = U; ep.step = t; ep.seq_len = dlen;
        ep.acts = r + R.acts[0] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U;
        ep.c_new = const_cast<float *>(Cn + (size_t)(t + 1) * B * U);
        ep.c_prev = Cn + (size_t)t * B * U;
        ep.dh2 = have_carry ? w + W.dxh[(t + 1) & 1] + (size_t)b0 * (E + U) + E : nullptr;
        ep.ld_dh2 = E + U;
        ep.dc_in = w + W.dc[cur][0] + (size_t)b0 * U;
        ep.dz = dzt;
        ep.dc_out = w + W.dc[cur ^ 1][0] + (size_t)b0 * U;
        ep.keep = drop ? d->keep_prob : 1.f; ep.seed = d->seed; ep.seed_offset = d->seed_offset + (unsigned long long)t;
        ep.row0 = b0;
        float *fp = w + W.fpart + (size_t)sub * W.fpart_each;
        unsigned *tk = reinterpret_cast<unsigned *>(w + W.tickets) + (size_t)sub * 1024;
        if (r16) {
          SP_TRY(rows16(Bn, U, U, dqt, U, w + W.wq_sw, 1.f, dHt, U, ss.st[sub], &ep));
          if (t > 0) {
            SkinnySplit sp = {w + W.dxh[t & 1] + (size_t)b0 * (E + U) + E, E + U, E, 0.f};
            SP_TRY(rows16(Bn, E + U, 4 * U, dzt, 4 * U, w + W.kxh_sw, 1.f, dCtx + (size_t)(t - 1) * B * E + (size_t)b0 * E, E,
                          ss.st[sub], nullptr, &sp));
          }
          have_carry = true;
          cur ^= 1;
          continue;
        }
        SP_TRY(gemm_skinny_fused(Bn, U, U, dqt, U, w + W.wqT, U, 0, nullptr, 0, nullptr, 0, 1.f, dHt, U, nullptr, fp, tk,
                                 ss.st[sub], &ep));
        if (split_b && t > 0) {
          // d context of step t-1 goes straight into that step's dCtx row block (on top of the output
          // projection's share), d h into the carry: no separate add launch in front of the next attention
          SkinnySplit sp = {w + W.dxh[t & 1] + (size_t)b0 * (E + U) + E, E + U, E, 0.f};
          SP_TRY(gemm_skinny_fused(Bn, E + U, 4 * U, dzt, 4 * U, w + W.kxhT, E + U, 0, nullptr, 0, nullptr, 0, 1.f,
                                   dCtx + (size_t)(t - 1) * B * E + (size_t)b0 * E, E, nullptr, fp, tk, ss.st[sub], nullptr,
                                   &sp));
        } else {
          SP_TRY(gemm_skinny_fused(Bn, E + U, 4 * U, dzt, 4 * U, w + W.kxhT, E + U, 0, nullptr, 0, nullptr, 0, 0.f,
                                   w + W.dxh[t & 1] + (size_t)b0 * (E + U), E + U, nullptr, fp, tk, ss.st[sub], nullptr));
        }
        have_carry = true;
        cur ^= 1;
        continue;
      }
      SP_TRY(mm2(Bn, U, U, dqt, U, w + W.wqT, U, 0, nullptr, 0, nullptr, 0, 1.f, dHt, U, w, W, sub, gws, gwb, st));
      const float *dtop = dHt;
      for (int n = nl - 1; n >= 0; --n) {
        const float *dh_in = dtop;
        if (drop) {
          SP_TRY(dropout_rows((size_t)Bn * U, dtop, w + W.tmp + (size_t)b0 * U, d->keep_prob, d->seed,
                              d->seed_offset + (unsigned long long)t * nl + n, (size_t)b0 * U, ss.st[sub]));
          dh_in = w + W.tmp + (size_t)b0 * U;
        }
        float *dzt = w + W.dz[n] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U;
        const float *Cn = r + R.Cs[n] + (size_t)b0 * U;
        SP_TRY(nabu_lstm_cell_bwd(Bn, U, t, dlen, r + R.acts[n] + (size_t)t * B * 4 * U + (size_t)b0 * 4 * U,
                                  Cn + (size_t)(t + 1) * B * U, Cn + (size_t)t * B * U, dh_in,
                                  w + W.dh[cur][n] + (size_t)b0 * U, w + W.dc[cur][n] + (size_t)b0 * U, dzt,
                                  w + W.dc[cur ^ 1][n] + (size_t)b0 * U, st));
        if (n == 0) {
          float *nx = w + W.dctx[t & 1] + (size_t)b0 * E;
          SP_TRY(mm2(Bn, E, 4 * U, dzt, 4 * U, w + W.kxT[0], E, 0, nullptr, 0, nullptr, 0, 0.f, nx, E, w, W, sub, gws, gwb, st));
          SP_TRY(mm2(Bn, U, 4 * U, dzt, 4 * U, w + W.khT[0], U, 0, nullptr, 0, nullptr, 0, 0.f,
                     w + W.dh[cur ^ 1][0] + (size_t)b0 * U, U, w, W, sub, gws, gwb, st));
        } else {
          SP_TRY(mm2(Bn, U, 4 * U, dzt, 4 * U, w + W.kxT[n], U, 0, nullptr, 0, nullptr, 0, 0.f, w + W.dx + (size_t)b0 * U, U,
                     w, W, sub, gws, gwb, st));
          SP_TRY(mm2(Bn, U, 4 * U, dzt, 4 * U, w + W.khT[n], U, 0, nullptr, 0, nullptr, 0, 0.f,
                     w + W.dh[cur ^ 1][n] + (size_t)b0 * U, U, w, W, sub, gws, gwb, st));
          dtop = w + W.dx + (size_t)b0 * U;
        }
      }
    }
    have_carry = true;
    cur ^= 1;
  }
  return 0;
  };
  if (!persist) {
  {
    const int e_run = run_subs(NS, bwd_chain), e_join = sub_join(ss);
    if (e_run) return e_run;
    SP_TRY(e_join);
  }
  }
  if (defer)
    SP_TRY(attn_param_grads(&adb, Sp, L, dec_len, enc_len, r + R.keys, r + R.q, p->attention_v, p->conv_proj, w + W.ds_all,
                            w + W.cf_all, dkeys, w + W.dv16, w + W.dwf16, s));
  // sums over steps as single GEMMs
  SP_TRY(mm(true, false, U, U, BL, htop_all, U, dq, U, 0.f, g->query_kernel, U, nullptr, gw, gwb, stream));
  for (int n = 0; n < nl; ++n) {
    const float *dzn = w + W.dz[n];
    float *gK = g->lstm_kernel[n];
    if (n == 0) {
      SP_TRY(nabu_scatter_rows_f32(C, BL, 4 * U, reinterpret_cast<const int32_t *>(r + R.ids), dzn, gK, stream));
      SP_TRY(mm(true, false, E, 4 * U, BL, r + R.ctx, E, dzn, 4 * U, 0.f, gK + (size_t)C * 4 * U, 4 * U, nullptr, gw, gwb, stream));
      SP_TRY(mm(true, false, U, 4 * U, BL, r + R.H[0], U, dzn, 4 * U, 0.f, gK + (size_t)(C + E) * 4 * U, 4 * U, nullptr, gw, gwb, stream));
    } else {
      SP_TRY(mm(true, false, U, 4 * U, BL, r + R.Ho[n - 1] + (size_t)B * U, U, dzn, 4 * U, 0.f, gK, 4 * U, nullptr, gw, gwb, stream));
      SP_TRY(mm(true, false, U, 4 * U, BL, r + R.H[n], U, dzn, 4 * U, 0.f, gK + (size_t)U * 4 * U, 4 * U, nullptr, gw, gwb, stream));
    }
    SP_TRY(nabu_colsum_f32(BL, 4 * U, dzn, 4 * U, 0.f, g->lstm_bias[n], gw, gwb, stream));
  }
  if (persist && !defer) SP_TRY(nabu_colsum_f32(B * 8, U, w + W.dv8, U, 0.f, g->attention_v, gw, gwb, stream));
  else if (defer)        SP_TRY(nabu_colsum_f32(B * Sp, U, w + W.dv16, U, 0.f, g->attention_v, gw, gwb, stream));
  else            SP_TRY(nabu_colsum_f32(B * S, U, w + W.dv, U, 0.f, g->attention_v, gw, gwb, stream));
  if (d->kind == 1) {
    if (defer) SP_TRY(nabu_colsum_f32(B * Sp, F * U, w + W.dwf16, F * U, 0.f, g->conv_proj, gw, gwb, stream));
    else       SP_TRY(nabu_colsum_f32(B * S, F * U, w + W.dwf, F * U, 0.f, g->conv_proj, gw, gwb, stream));
    if (persist) SP_TRY(nabu_colsum_f32(B * 8, K * F, w + W.dck8, K * F, 0.f, g->conv_kernel, gw, gwb, stream));
    else         SP_TRY(nabu_colsum_f32(B, K * F, w + W.dck, K * F, 0.f, g->conv_kernel, gw, gwb, stream));
  }
  // keys = values·Wmem ; context_t = align_t^T·values
  SP_TRY(mm(true, false, E, U, B * Te, values, E, dkeys, U, 0.f, g->memory_kernel, U, nullptr, gw, gwb, stream));
  SP_TRY(mm(false, true, B * Te, E, U, dkeys, U, p->memory_kernel, U, 0.f, dvalues, E, nullptr, gw, gwb, stream));
  const float *al1 = r + R.align + (size_t)B * Te;
  // dvalues[b] += align[:, b, :]^T · dCtx[:, b, :] for every utterance: one batched launch
  SP_TRY(gemm_batched_f32(true, false, Te, E, L, al1, B * Te, Te, dCtx, B * E, E, 1.f, dvalues, E, (long long)Te * E, B, s));
  return 0;
}
