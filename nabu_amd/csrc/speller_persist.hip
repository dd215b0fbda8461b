// speller_persist.hip — the L decoder steps of the Speller (reference rnn_decoder.py:59-77: dynamic_decode over
// AttentionWrapper(LSTMCell, BahdanauAttention), speller.py:30-61, attention.py:142-184) as ONE persistent
// launch.  The step chain of speller.hip costs one kernel per dependent phase (product+cell, query product,
// attention, finish: ~42 us per step at the cfg3 geometry whatever the batch); here the phases hand over through
// L2 the way the persistent recurrence does (lstm_persist.hip).
//
// DECOMPOSITION — the design of the recurrence applied to the decoder: utterances are independent, so every XCD
// (unit = 32 workgroups of 256 threads = one L2) decodes its own B/8 = 4 utterances and ALL exchange traffic of a
// step stays inside that L2.  The price is that every XCD holds the whole cell kernel: [(E+U) x 4U] fp32 =
// 12.6 MB at cfg3 — in the REGISTER FILES of its 32 CUs (394 KB of the 512 KB of a CU), loaded once.
// Workgroup j of a unit has four duties per step t:
//   A  cell:   z[4 rows, my 4U/32 gate columns] = [ctx_{t-1} | h_{t-1}] . Kp on the matrix pipe, exact fp32
//              (v_mfma_f32_4x4x1: 16 blocks = 16 groups of 4 gate columns, A = X[row][k], B = Kp[k][column] from
//              registers; the 4 waves split k), + embedding row + bias, gates, c_t, h_t  -> publishes h_t slice
//   B  query:  q_t[4 rows, my U/32 columns] = h_t . Wq (my columns of Wq in LDS)      -> publishes q_t slice
//   C  scores: workgroup (utterance i = j/8, frame slice s = j%8): v.tanh(keys + q_t) over ITS frames (keys and
//              values slices LDS-resident for the whole launch), local softmax statistics, partial context
//                                                                                      -> publishes (m, z, part[E])
//   D  combine: the 8 slices of an utterance each combine an E/8 column block of the context and normalise their
//              own alignments                                                            -> publishes ctx_t block
// EXCHANGE: four rings of 4 slots per unit (h, q, partials, ctx), "the data is the flag" exactly as in
// lstm_persist.hip: slots pre-filled with 0xFFFFFFFF, consumers re-load with L1-bypassing loads until no word is
// the sentinel, a producer hands its piece of slot t-2 back right after publishing slot t.  Why that is safe
// without timing: when a workgroup publishes into ring X at step t it has gathered (in this or the previous duty)
// pieces that EVERY reader of its X_{t-2} piece published after reading it; and a reader polls slot t+2 of X only
// after gathering something the producer published after its reset AND after a later poll loop of its own (loads
// issued behind the reset store have returned: vector-memory operations complete in issue order).  Placement: the
// units are formed at run time — a workgroup reads its XCC id and takes the next free slot of that XCD's unit (one
// fetch-add); one workgroup fills a CU, so every XCD ends up with exactly 32 whatever the dispatch order.
// Every spin is bounded (time-out -> status word -> everybody leaves).
#include "speller_persist.h"

#include <stdlib.h>

#include "lstm_persist.h"

namespace nabu {
namespace {

constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr unsigned OOB = 0xFFFFFFF0u;
#ifndef SP_NW
#define SP_NW 4
#endif
constexpr int NW = SP_NW, NT = 64 * NW;      // waves, threads per workgroup (4: ONE wave per SIMD, up to 512 registers per lane)
constexpr int NU = 8, P = 32;        // units (XCDs), workgroups per unit
constexpr int R = 4, S = 8;          // utterances per unit, frame slices per utterance
constexpr int RING = 4;
constexpr size_t TABLE_BYTES = 4096;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  int L, U, E, Te, FS;               // FS = frames per slice
  int Bt, b0;                        // rows of the whole batch (strides of the time-major tensors); my first row
  int K, F, stream_vals;             // location-aware attention: filter taps, filters; values slice read from L2 per step
  const float *ck, *wf;              // conv kernel [K][F], feature projection [F][U]
  const int32_t *dec_len, *enc_len, *ids;
  const float *kperm, *bias, *emb, *wq, *v, *keys, *values;
  float *H, *Cs, *acts, *q, *ctx, *align;
  const float *wout, *bout;          // scheduled sampling: output projection [(U+E), C], [C]
  int32_t *ids_w;                    // ... the inputs actually used (= ids), rows 1.. written here
  int C;
  float sprob;
  const unsigned *sdraw;             // ... the Philox words (x: Bernoulli, y: inverse CDF) of (seed, offset + t, row) [L, Bt, 2]
  float *Ho;                         // dropped cell outputs (keep < 1), else unused
  float keep;                        // output dropout keep probability (1 = off)
  const float *dscale;               // ... its scale factors (0 or 1 / keep) [L, Bt, U], written before the launch
  unsigned *table;
  char *xbuf;
  int *status;
  unsigned long long timeout_ticks;
  int dbg;
};

__device__ __forceinline__ bool has_sentinel(const u32x4 v) {
  return v.x == SENT || v.y == SENT || v.z == SENT || v.w == SENT;
}
// wave64 reductions on the DPP network (no LDS round trips); every lane gets the result
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dppf(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wsum(float v) {
  v += dppf<0xB1, 0xF>(0.f, v);          // quad_perm [1,0,3,2]
  v += dppf<0x4E, 0xF>(0.f, v);          // quad_perm [2,3,0,1]
  v += dppf<0x141, 0xF>(0.f, v);         // row_half_mirror
  v += dppf<0x140, 0xF>(0.f, v);         // row_mirror: every lane of a row holds the row's sum
  v += dppf<0x142, 0xA>(0.f, v);         // row_bcast15 -> rows 1, 3
  v += dppf<0x143, 0xC>(0.f, v);         // row_bcast31 -> rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over the 16 lanes of a row (every lane of the row ends with it)
__device__ __forceinline__ float rowsum(float v) {
  v += dppf<0xB1, 0xF>(0.f, v);
  v += dppf<0x4E, 0xF>(0.f, v);
  v += dppf<0x141, 0xF>(0.f, v);
  v += dppf<0x140, 0xF>(0.f, v);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, dppf<0xB1, 0xF>(v, v));
  v = fmaxf(v, dppf<0x4E, 0xF>(v, v));
  v = fmaxf(v, dppf<0x141, 0xF>(v, v));
  v = fmaxf(v, dppf<0x140, 0xF>(v, v));
  v = fmaxf(v, dppf<0x142, 0xA>(v, v));
  v = fmaxf(v, dppf<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float quad_bcast(float v, int i) {
  switch (i) {
    case 0: return dppf<0x00, 0xF>(v, v);
    case 1: return dppf<0x55, 0xF>(v, v);
    case 2: return dppf<0xAA, 0xF>(v, v);
    default: return dppf<0xFF, 0xF>(v, v);
  }
}
// gate functions as in lstm_persist_dev.h (round 5): v_rcp_f32 is not centred and 2 r - 1 amplifies its bias into a relative
// bias of tanh — one Newton step behind the reciprocal, tanh as (1 - e) / (1 + e)
__device__ __forceinline__ float rcpn(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float fsig(float x) { return rcpn(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) {
  const float e = __expf(-2.0f * __builtin_amdgcn_fmed3f(x, -30.0f, 30.0f));
  return (1.0f - e) * rcpn(1.0f + e);
}
// the tanh inside the attention score v . tanh(keys + q + ...): not part of a recurrent state (nothing accumulates its
// ~1e-7 bias over the steps), and FS x U of them per workgroup and step — the centred form above costs the forward
// kernel 14 % (cfg3) to 19 % (cfg5) of its time there
__device__ __forceinline__ float ftanh_s(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

struct Spin {
  unsigned long long t0;
  unsigned n;
  __device__ __forceinline__ void start() { n = 0; }
  __device__ __forceinline__ bool expired(const Args &p) {
    if (n == 0) t0 = wall_clock64();      // the clock is only read once a poll has failed
    if ((++n & 31u) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    return wall_clock64() - t0 > p.timeout_ticks;
  }
};

// stores of the exchange: plain when the unit shares one L2, else write-through
__device__ __forceinline__ void xst4(const u32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned off, bool coloc) {
  if (coloc) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
  else       __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}
__device__ __forceinline__ void xst1(unsigned v, __amdgpu_buffer_rsrc_t rs, unsigned off, bool coloc) {
  if (coloc) __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 0);
  else       __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, 16);
}
__device__ __forceinline__ u32x4 xld4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
}
__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }

#define SP_TIMEOUT(code)                                                                                    \
  do {                                                                                                      \
    if ((threadIdx.x & 63) == 0) {                                                                          \
      flag[0] = 1;                                                                                          \
      __hip_atomic_store(p.status, (code) + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
    }                                                                                                       \
  } while (0)

// Scheduled sampling, the RARE path (sample_prob = 0.1 in the reference's defaults): the row of this workgroup is to be
// sampled — gather [ho_t | ctx_t] from the exchange, logits = row . Wout + b, draw from softmax(logits) exactly as
// sample_ids_kernel does.  NOT inlined: inside the forward kernel these ~150 lines cost the common path 102 spilled
// registers (the sampling instantiation ran at 36 us per decoder step against 13 without sampling); as a function the
// spills around the call are paid by the one step in ten (per row) that takes it.
struct SampleRow {
  __amdgpu_buffer_rsrc_t rc, rh;
  unsigned hoff, coff;          // byte offsets of my row's h / context inside their rings (slot of this step)
  const float *wout, *bout;
  int C, U, E;
  const float *dscale;          // dropout scale factors of my row in this step [U] (nullptr: no dropout)
  float *aux;
  int *flag;
  int *status;
  unsigned long long timeout_ticks;
  unsigned ry;                  // the second Philox word of (seed, offset + t, row): the uniform of the inverse CDF
  int teacher;
};
#ifndef SAMPLE_UNROLL
// 16-byte weight loads in flight per thread.  Deliberately few: every register this function uses beyond the
// call-clobbered set is one the CALLER (512 registers live, 384 of them weights) spills in its common path — cfg3 with
// sample_prob 1e-6 over the plain step: +0.11 ms at 2, +0.4 at 8, +2.4 at 16 (79 / 117 / 228 spilled registers)
#define SAMPLE_UNROLL 2
#endif
__device__ __attribute__((noinline)) int sample_row(const SampleRow q) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int C = q.C, U = q.U, K = q.U + q.E;
  float *xrow = q.aux, *lred = q.aux + K, *e_s = lred + NW * 64;
  int *id_s = reinterpret_cast<int *>(e_s + 64);
  {
    // my row of [h_t (U) | ctx_t (E)], published in this step: K/4 pieces, <= 2 per thread
    const int NPC = K / 4;
    u32x4 v[2];
    unsigned offs[2];
    bool isc[2];
    int qq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      qq[j] = min(j * NT + tid, NPC - 1);
      isc[j] = 4 * qq[j] >= U;
      offs[j] = isc[j] ? q.coff + (unsigned)((4 * qq[j] - U) * 4) : q.hoff + (unsigned)(4 * qq[j] * 4);
    }
    unsigned long long t0 = 0;
    unsigned n = 0;
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4 a = xld4(q.rc, isc[j] ? offs[j] : OOB), b = xld4(q.rh, isc[j] ? OOB : offs[j]);
        v[j] = isc[j] ? a : b;
        ok = ok && !has_sentinel(v[j]);
      }
      if (__all(ok)) break;
      if (n == 0) t0 = wall_clock64();
      if ((++n & 31u) != 0) continue;
      __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(q.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > q.timeout_ticks) {
        if ((threadIdx.x & 63) == 0) {
          q.flag[0] = 1;
          __hip_atomic_store(q.status, 1 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j * NT + tid < NPC) {
        f32x4 f = __builtin_bit_cast(f32x4, v[j]);
        if (q.dscale && !isc[j]) {                 // the projection sees the DROPPED output
          const f32x4 sc = *reinterpret_cast<const f32x4 *>(q.dscale + 4 * qq[j]);
          f.x *= sc.x; f.y *= sc.y; f.z *= sc.z; f.w *= sc.w;
        }
        *reinterpret_cast<f32x4 *>(xrow + 4 * qq[j]) = f;
      }
  }
  __syncthreads();
  if (q.flag[0]) return q.teacher;
  if (C % 4 == 0 && C <= 4 * (NT / 16)) {
    // thread (class quad cq = tid / 16, k slot ks = tid % 16): 16-byte loads of the weight rows k = ks, ks + 16, ...,
    // eight in flight; the 16 k slots of a class quad are 16 consecutive lanes: four shuffle steps add them
    const int cq = tid >> 4, ks = tid & 15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (4 * cq < C) {
      const float *wp = q.wout + 4 * cq;
#pragma unroll SAMPLE_UNROLL
      for (int k = ks; k < K; k += 16) {
        const float x = xrow[k];
        const f32x4 wv = *reinterpret_cast<const f32x4 *>(wp + (size_t)k * C);
        acc.x = fmaf(x, wv.x, acc.x); acc.y = fmaf(x, wv.y, acc.y); acc.z = fmaf(x, wv.z, acc.z); acc.w = fmaf(x, wv.w, acc.w);
      }
    }
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) {
      acc.x += __shfl_xor(acc.x, m); acc.y += __shfl_xor(acc.y, m); acc.z += __shfl_xor(acc.z, m); acc.w += __shfl_xor(acc.w, m);
    }
    // lred as [wave 0 .. NW - 1][64]: the totals go to "wave 0"'s row, the other rows read as zero below
    for (int i = tid; i < NW * 64; i += NT) lred[i] = 0.f;
    __syncthreads();
    if (ks == 0 && 4 * cq < C) {
      lred[4 * cq] = acc.x; lred[4 * cq + 1] = acc.y; lred[4 * cq + 2] = acc.z; lred[4 * cq + 3] = acc.w;
    }
  } else {
    // lane = class, the waves split k; Wout is read from L2 (its rows are contiguous over the classes)
    float part = 0.f;
    if (lane < C) {
      const int k0 = w * (K / NW);
      const float *wp = q.wout + (size_t)k0 * C + lane;
      for (int k = 0; k < K / NW; k += 4) {
        const f32x4 x4 = *reinterpret_cast<const f32x4 *>(xrow + k0 + k);
        part = fmaf(x4.x, wp[(size_t)k * C], part);
        part = fmaf(x4.y, wp[(size_t)(k + 1) * C], part);
        part = fmaf(x4.z, wp[(size_t)(k + 2) * C], part);
        part = fmaf(x4.w, wp[(size_t)(k + 3) * C], part);
      }
    }
    lred[w * 64 + lane] = part;
  }
  __syncthreads();
  if (w == 0) {
    float l = -INFINITY;
    if (lane < C) {
      l = q.bout[lane];
      for (int ww = 0; ww < NW; ++ww) l += lred[ww * 64 + lane];
    }
    const float m = wmax(l);
    e_s[lane] = lane < C ? expf(l - m) : 0.f;
    if (lane == 0) {                             // the sums in class order, as sample_ids_kernel adds them
      float tot = 0.f;
      for (int c = 0; c < C; ++c) tot += e_s[c];
      const float target = u01(q.ry) * tot;
      float acc = 0.f;
      int pick = C - 1;
      for (int c = 0; c < C; ++c) {
        acc += e_s[c];
        if (acc > target) { pick = c; break; }
      }
      id_s[0] = pick;
    }
  }
  __syncthreads();
  return id_s[0];
}

// NABU_PERSIST_DEBUG bit 2: wall-clock stamps of the phases of step L/2 in block 0 (status[16 + i], 10 ns ticks)
#define SP_STAMP(i)                                                                \
  do {                                                                             \
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && t == L / 2)                  \
      p.status[16 + (i)] = (int)wall_clock64();                                    \
  } while (0)

// The random numbers of a regularised forward call, drawn before it:
//   dscale [L][B][U]: the output dropout's scale factors (0 or 1 / keep) — step t is the Philox stream dropout_rows draws
//                     for (seed, seed_offset + t) over [B, U] (group = 4 consecutive elements);
//   sdraw [L][B][2]:  words x, y of Philox((row, 0, offset + t), sample_seed) — sample_ids_kernel's Bernoulli and
//                     inverse-CDF uniforms of step t
__global__ __launch_bounds__(256) void speller_randoms_kernel(size_t groups, size_t draws, int B, int U, float keep,
                                                              unsigned long long seed, unsigned long long seed_offset,
                                                              float *dscale, unsigned long long sseed,
                                                              unsigned long long soff, unsigned *sdraw) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < groups) {
    const size_t per_step = (size_t)B * U / 4, t = i / per_step, g = i % per_step;
    const float4 sc = dropout_scale4(g, keep, seed, seed_offset + t);
    *reinterpret_cast<float4 *>(dscale + 4 * i) = sc;
  } else if (i < groups + draws) {
    const size_t j = i - groups, t = j / B, row = j % B;
    const unsigned long long off = soff + t;
    const uint4 rr = philox4x32_10(make_uint4((unsigned)row, 0u, (unsigned)off, (unsigned)(off >> 32)),
                                   make_uint2((unsigned)sseed, (unsigned)(sseed >> 32)));
    sdraw[2 * j] = rr.x; sdraw[2 * j + 1] = rr.y;
  }
}

// KR = weight registers per lane >= (E+U)/4
// LOC: location-aware attention (attention.py:186-292): the score also takes conv1d(previous alignments)·conv_proj;
// the normalised alignments travel in a fifth ring
// REG: the regularised training recipes — output dropout and / or scheduled sampling (their code is compiled out of the
// plain instantiations, whose register allocation is tight)
template <int KR, bool LOC, bool REG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void speller_persist_fwd_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int flag[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int unit, slot;     // decided at start-up: unit = the XCD this workgroup runs on, slot = its arrival rank there
  const int U = p.U, E = p.E, Te = p.Te, FS = p.FS, L = p.L;
  const int K = E + U, KW = K / NW;          // k range of a wave
  const int CW = 4 * U / P, UW = U / P;      // my gate columns / my units (= my q columns)
  const int CB = E / S;                      // my context columns in duty D
  const int B = p.Bt;
  const int Kc = LOC ? p.K : 0, Fc = LOC ? p.F : 0, pbc = (Kc - 1) / 2, TeP = S * FS;

  // ---- start-up: XCC ids of the unit (lstm_persist.hip: unit_handshake)
  // Which XCD a block lands on is the dispatcher's business (observed: b % 8 in one launch, pairs of consecutive
  // blocks per XCD in another).  So the units are formed at run time: a workgroup joins the unit of the XCD it runs on
  // and takes the next free slot there (one agent-scope fetch-add).  One workgroup fits a CU (one wave per SIMD, most
  // of the register file), an XCD has 32 CUs and all 256 workgroups are resident: every XCD ends up with exactly 32.
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
  if (tid == 0) {
    flag[0] = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // counters start at 0xFFFFFFFF (the workspace prefill): the first arrival reads that and takes slot 0
    const unsigned old = __hip_atomic_fetch_add(p.table + 512 + (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = (int)(old + 1u);
    if (old + 1u >= (unsigned)P || xcc >= (unsigned)NU) {       // cannot happen on a whole MI355X: give up loudly
      flag[0] = 1;
      __hip_atomic_store(p.status, 3 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (flag[0]) return;          // an earlier launch on this workspace timed out / no slot
  unit = (int)xcc;
  slot = flag[1];
  __syncthreads();
  const bool coloc = !(p.dbg & 8);      // the unit shares one L2 by construction (bit 3: force write-through)

  // ---- exchange rings of my unit: h [R][U], q [R][U], ctx [R][E], partials [R*S][E+4]
  const unsigned hb = (unsigned)(R * U * 4), cb = (unsigned)(R * E * 4), pb = (unsigned)(R * S * (E + 4) * 4);
  const unsigned lb = (unsigned)(R * TeP * 4);          // (LOC) alignments [R][S*FS]
  const size_t unit_bytes = (size_t)RING * (2 * hb + cb + pb + (LOC ? lb : 0) + 16);   // + 16: the sampled inputs [R]
  char *ub = p.xbuf + (size_t)unit * unit_bytes;
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)(RING * hb), 0x00020000);
  __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * hb, 0, (int)(RING * hb), 0x00020000);
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * 2 * hb, 0, (int)(RING * cb), 0x00020000);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (2 * hb + cb), 0, (int)(RING * pb), 0x00020000);
  __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (2 * hb + cb + pb), 0, (int)(RING * lb), 0x00020000);
  __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (2 * hb + cb + pb + (LOC ? lb : 0)), 0, RING * 16, 0x00020000);
  const bool samp = REG && p.sprob > 0.f;
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // ---- LDS: keys / values slices of my (utterance, frame slice), my columns of Wq, scratch
  float *keys_s = smem;                               // [FS][U]
  float *vals_s = keys_s + (size_t)FS * U;            // [FS][E] (absent when the values are streamed)
  float *wq_s = vals_s + (p.stream_vals ? 0 : (size_t)FS * E);              // [U][UW]
  float *v_s = wq_s + (size_t)U * UW;                 // [U] attention vector
  float *es = v_s + U;                                // [FS] exp(score - local max) of my frames
  // (LOC) feature projection [F][U], conv kernel [K][F], padded previous alignments [pb + S*FS + K], features [FS][F]
  float *wf_s = es + NW * 64;
  float *ck_s = wf_s + (size_t)Fc * U;
  float *alp_s = ck_s + (((Kc + 3) & ~3) * Fc);
  float *cf_s = alp_s + (LOC ? ((pbc + TeP + Kc + 8) & ~3) : 0);
  float *scr = cf_s + ((FS * Fc + 3) & ~3);           // scratch, sized by the host
  // scratch, re-used by the duties of a step (barriers in between):
  //   A: Xs = my wave's X[4 rows][KW]; its partial tile (256 floats) goes where its own X was (or behind all X)
  //   B: hs = h_t [R][U];  C: qs = q_t of my utterance [U], sc_s [FS] behind hs;  D: parts [S][CB], mz [S][2] over hs
  float *Xs = scr + (size_t)w * R * KR;              // rows of KR floats: columns >= KW stay zero (so do their weights)
  constexpr bool red_in_x = R * KR >= 256;
  constexpr int red_stride = red_in_x ? R * KR : 256;
  float *red0 = red_in_x ? scr : scr + NW * R * KR;   // tile of wave ww at red0 + ww * red_stride: [64 columns][4 rows]
  // (KW < KR: the staged X has zero padding that must stay zero -> the other duties' scratch lies behind it)
  float *aux = KW == KR ? scr : scr + NW * R * KR + (red_in_x ? 0 : NW * 256);
  float *hs = aux;                                    // [R][U + 4]
  float *qred = aux + R * (U + 4);                    // [NW][16 columns][4 rows]: duty B's partial sums
  float *qs = qred + NW * 64;
  float *sc_s = qs + U;
  float *parts = aux;
  float *mz = parts + S * CB;

  const int ci = slot / S, cs = slot % S;             // duties C, D: utterance of the unit, frame slice
  const int cbg = p.b0 + unit * R + ci;               // its batch row
  const int f0 = cs * FS;
  for (int i = tid; i < FS * U; i += NT) {
    const int f = f0 + i / U;
    keys_s[i] = f < Te ? p.keys[((size_t)cbg * Te + f) * U + i % U] : 0.f;
  }
  if (!p.stream_vals)
    for (int i = tid; i < FS * E; i += NT) {
      const int f = f0 + i / E;
      vals_s[i] = f < Te ? p.values[((size_t)cbg * Te + f) * E + i % E] : 0.f;
    }
  if (LOC) {
    for (int i = tid; i < Fc * U; i += NT) wf_s[i] = p.wf[i];
    for (int i = tid; i < ((Kc + 3) & ~3) * Fc; i += NT) {      // [filter][tap], taps padded to a multiple of 4 with zeros
      const int j = i / ((Kc + 3) & ~3), d = i % ((Kc + 3) & ~3);
      ck_s[i] = d < Kc ? p.ck[d * Fc + j] : 0.f;
    }
    for (int i = tid; i < ((pbc + TeP + Kc + 8) & ~3); i += NT) alp_s[i] = 0.f;       // the pads stay zero; alignments of step -1 are zero
  }
  for (int i = tid; i < U * UW; i += NT) wq_s[i] = p.wq[(size_t)(i / UW) * U + UW * slot + i % UW];   // [k][column]
  for (int i = tid; i < U; i += NT) v_s[i] = p.v[i];
  for (int i = tid; i < NW * R * KR; i += NT) scr[i] = 0.f;

  // ---- duty A identities.  Matrix product: lane = (block = group of 4 gate columns, jj); A operand row = lane & 3
  const int blk = lane >> 2, jj = lane & 3;
  float Wr[KR];
  {
    const int col = 4 * blk + jj;
#pragma unroll
    for (int kk = 0; kk < KR; ++kk)
      Wr[kk] = (kk < KW && col < CW) ? p.kperm[(size_t)(w * KW + kk) * 4 * U + CW * slot + col] : 0.f;
  }
  // gate phase: thread = (row, my column col_l = 4*unit + gate), tid < 4 * 64
  const int grow = tid >> 6, gcol = tid & 63, gu = gcol >> 2, gg = gcol & 3;
  const bool gate_thr = tid < R * 64 && gcol < CW;
  const int gb = p.b0 + unit * R + grow;                // batch row
  const int gunit = UW * slot + gu;                     // hidden unit
  const float gbias = gate_thr ? p.bias[gg * U + gunit] : 0.f;
  const int glen = gate_thr ? p.dec_len[gb] : 0;
  float c_state = 0.f, h_state = 0.f;
  // duty D state: my context column of the previous step, my frames' alignments of the previous step
  float ctx_prev = 0.f, al_prev = 0.f;
  // Saved tensors (what the backward pass reads): written one step late, right before the matrix product of the
  // next step — a poll loop waits for every older memory operation of its wave (one in-order counter), and an HBM
  // store in front of a poll put its acknowledgement (~4 us for the scattered 4-byte stores) on the critical path
  float a_last = 0.f, q_last = 0.f, ho_last = 0.f;
  const bool drop = REG && p.keep < 1.f;
  auto save_step = [&](int ts) {     // step ts is complete in my registers
    if (gate_thr) {
      p.acts[((size_t)ts * B + gb) * 4 * U + gg * U + gunit] = a_last;
      if (gg == 0) {
        p.Cs[((size_t)(ts + 1) * B + gb) * U + gunit] = c_state;
        p.H[((size_t)(ts + 1) * B + gb) * U + gunit] = h_state;
        if (drop) p.Ho[((size_t)(ts + 1) * B + gb) * U + gunit] = ho_last;
      }
    }
    if (tid < R * UW) p.q[((size_t)ts * B + p.b0 + unit * R + tid / UW) * U + UW * slot + tid % UW] = q_last;
    if (tid < CB) p.ctx[((size_t)(ts + 1) * B + cbg) * E + cs * CB + tid] = ctx_prev;
    if (tid < FS && f0 + tid < Te) p.align[((size_t)(ts + 1) * B + cbg) * Te + f0 + tid] = al_prev;
  };
  const int clen = p.dec_len[cbg], cn = min(max(p.enc_len[cbg], 0), Te);
  __syncthreads();

  for (int t = 0; t < L; ++t) {
    const unsigned so = (unsigned)(t % RING), sp = (unsigned)((t + RING - 1) % RING), sr = (unsigned)((t + RING - 2) % RING);
    SP_STAMP(0);
    // scheduled sampling: was the input of this step — of the row my wave owns in the gate phase — sampled at the end
    // of the previous one?  (the Bernoulli words were drawn before the launch: only a sampled row's wave has to wait for
    // the word its utterance's first workgroup publishes, every other reads the teacher's label)
    bool in_sampled = false;
    if (samp && t > 0) {
      const int ix = __builtin_amdgcn_readfirstlane(2 * ((t - 1) * B + p.b0 + unit * R + min(tid >> 6, R - 1)));
      in_sampled = u01(p.sdraw[ix]) < p.sprob;
    }
    // =========================== A: cell ===========================
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (t > 0) {
      // gather X = [ctx_{t-1} | h_{t-1}], the k range of my wave: KW/4 pieces of 16 bytes per row
      constexpr int NQ = (KR + 63) / 64;
      const int PR = KW / 4, NPC = R * PR;               // NPC = KW <= KR pieces
      u32x4 v[NQ];
      unsigned off[NQ];
      bool fc[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int qi = min(64 * i + lane, NPC - 1);
        const int row = qi / PR, k = w * KW + 4 * (qi % PR);
        fc[i] = k < E;
        off[i] = fc[i] ? sp * cb + (unsigned)((row * E + k) * 4) : sp * hb + (unsigned)((row * U + (k - E)) * 4);
      }
      Spin g;
      g.start();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const u32x4 a = xld4(rc, fc[i] ? off[i] : OOB), b = xld4(rh, fc[i] ? OOB : off[i]);
          v[i] = fc[i] ? a : b;
          ok = ok && !has_sentinel(v[i]);
        }
        if (__all(ok)) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int qi = 64 * i + lane;
        if (qi < NPC) *reinterpret_cast<f32x4 *>(Xs + (qi / PR) * KR + 4 * (qi % PR)) = __builtin_bit_cast(f32x4, v[i]);
      }
      SP_STAMP(1);
      save_step(t - 1);
      // product (in-order LDS: my wave reads what it wrote)
      const float *xr = Xs + (lane & 3) * KR;
      // four accumulator chains: with ONE wave per SIMD nobody else fills the result latency of a dependent product
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
      // One wave per SIMD: nobody else hides the LDS latency, and the compiler sinks operand reads to one group
      // ahead — so the reads of the NEXT chunk of 16 k are pinned in front of this chunk's 16 products
      constexpr int CH = 4, NCH = KR / 4 / CH;
      f32x4 xq[2][CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) xq[0][i] = *reinterpret_cast<const f32x4 *>(xr + 4 * i);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
          for (int i = 0; i < CH; ++i) xq[(c + 1) & 1][i] = *reinterpret_cast<const f32x4 *>(xr + 4 * ((c + 1) * CH + i));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const f32x4 x = xq[c & 1][i];
          const int k4 = c * CH + i;
          a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, Wr[4 * k4 + 0], a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, Wr[4 * k4 + 1], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, Wr[4 * k4 + 2], a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, Wr[4 * k4 + 3], a3, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      a0 += a2;
      a1 += a3;
      acc = a0 + a1;
    }
    SP_STAMP(2);
    *reinterpret_cast<f32x4 *>(red0 + (size_t)w * red_stride + lane * 4) = acc;     // [column][row]
    __syncthreads();
    if (flag[0]) return;
    if (tid < R * 64) {
      float z = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) z += red0[(size_t)ww * red_stride + gcol * 4 + grow];
      float a = 0.f;
      int my_id = 0;
      if (in_sampled) {
        // the input of this step was drawn at the end of the previous one (duty E): wave w = row w polls its word
        unsigned idw;
        Spin g;
        g.start();
        for (;;) {
          idw = __builtin_amdgcn_raw_buffer_load_b32(ri, sp * 16 + (unsigned)(grow * 4), 0, 16);
          if (__all(idw != SENT)) break;
          if (g.expired(p)) { SP_TIMEOUT(1); idw = 0; break; }
        }
        my_id = (int)idw;
      } else if (gate_thr) {
        my_id = p.ids[(size_t)t * B + gb];
      }
      if (gate_thr) {
        z += gbias + p.emb[(size_t)my_id * 4 * U + gg * U + gunit];
        a = gg == 1 ? ftanh(z) : fsig(gg == 2 ? z + 1.0f : z);
      }
      const float gi = quad_bcast(a, 0), gj = quad_bcast(a, 1), gf = quad_bcast(a, 2), go = quad_bcast(a, 3);
      const bool act = t < glen;
      if (gate_thr && act) {
        c_state = c_state * gf + gi * gj;
        h_state = ftanh(c_state) * go;
      }
      a_last = act ? a : 0.f;        // saved tensors of this step go to HBM under the NEXT step's product
      if (drop && gate_thr && gg == 0)     // what the query and the output projection see of h_t (the recurrence keeps h_t)
        ho_last = h_state * p.dscale[((size_t)t * B + gb) * U + gunit];
      const bool pub = gate_thr && gg == 0;
      xst1(fbits(h_state), rh, pub ? so * hb + (unsigned)((grow * U + gunit) * 4) : OOB, coloc);
      xst1(SENT, rh, (pub && t >= 2) ? sr * hb + (unsigned)((grow * U + gunit) * 4) : OOB, coloc);
    }
    SP_STAMP(3);
    __syncthreads();     // the partial tiles have been read: the scratch is free for h_t
    // =========================== B: query ==========================
    {
      const int NPC = R * U / 4;                       // <= 2 * NT (host check)
      const int q0 = min(tid, NPC - 1), q1 = min(NT + tid, NPC - 1);
      u32x4 v0, v1;
      // the query is taken from the DROPPED output: the scale factors of my two pieces (written before the launch:
      // the loads are on their way while the poll below waits)
      f32x4 s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0;
      if (drop) {
        const float *ds = p.dscale + ((size_t)t * B + p.b0 + unit * R) * U;
        s0 = *reinterpret_cast<const f32x4 *>(ds + (size_t)(4 * q0 / U) * U + 4 * q0 % U);
        s1 = *reinterpret_cast<const f32x4 *>(ds + (size_t)(4 * q1 / U) * U + 4 * q1 % U);
      }
      Spin g;
      g.start();
      for (;;) {
        v0 = xld4(rh, so * hb + (unsigned)(q0 * 16));
        v1 = xld4(rh, so * hb + (unsigned)(q1 * 16));
        if (__all(!has_sentinel(v0) && !has_sentinel(v1))) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
      f32x4 f0 = __builtin_bit_cast(f32x4, v0), f1 = __builtin_bit_cast(f32x4, v1);
      if (drop) {
        f0.x *= s0.x; f0.y *= s0.y; f0.z *= s0.z; f0.w *= s0.w;
        f1.x *= s1.x; f1.y *= s1.y; f1.z *= s1.z; f1.w *= s1.w;
      }
      if (tid < NPC) *reinterpret_cast<f32x4 *>(hs + (4 * q0 / U) * (U + 4) + 4 * q0 % U) = f0;
      if (NT + tid < NPC) *reinterpret_cast<f32x4 *>(hs + (4 * q1 / U) * (U + 4) + 4 * q1 % U) = f1;
    }
    SP_STAMP(4);
    __syncthreads();
    if (flag[0]) return;
    {
      // q[4 rows][my UW columns] on the matrix pipe: blocks = (column group cg, k phase ks); wave w takes the k range
      // [w U/4, (w+1) U/4), lane (cg, ks, j): A = h[row j][k + ks], B = Wq[k + ks][4 cg + j].  hs rows are padded by
      // 4 floats: the 16 (row, ks) operands of an instruction then sit in 16 different banks.
      const int bcg = lane >> 4, bks = (lane >> 2) & 3, bj = lane & 3;
      f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qb = {0.f, 0.f, 0.f, 0.f};
      {
        const float *ha = hs + bj * (U + 4) + w * (U / NW) + bks;
        const float *wb = wq_s + (size_t)(w * (U / NW) + bks) * UW + min(4 * bcg + bj, UW - 1);
        const bool colok = 4 * bcg + bj < UW;
        for (int kk = 0; kk < U / NW / 4; kk += 2) {
          const float a0 = ha[4 * kk], a1 = ha[4 * kk + 4];
          const float b0 = colok ? wb[(size_t)4 * kk * UW] : 0.f, b1 = colok ? wb[(size_t)(4 * kk + 4) * UW] : 0.f;
          qa = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, qa, 0, 0, 0);
          qb = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, qb, 0, 0, 0);
        }
      }
      f32x4 qs4 = qa + qb;       // [row i] of column 4 cg + j, k phase ks: add the four phases (lanes ^4, ^8)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        qs4[i] += __shfl_xor(qs4[i], 4);
        qs4[i] += __shfl_xor(qs4[i], 8);
      }
      if (bks == 0) *reinterpret_cast<f32x4 *>(qred + ((size_t)w * 16 + 4 * bcg + bj) * 4) = qs4;   // [wave][column][row]
      __syncthreads();
      // thread (row, column) adds the four k ranges
      const int row = tid / UW, c = tid % UW;
      float s = 0.f;
      if (tid < R * UW)
        for (int ww = 0; ww < NW; ww += 2) s += qred[(ww * 16 + c) * 4 + row] + qred[((ww + 1) * 16 + c) * 4 + row];
      const bool pub = tid < R * UW;
      q_last = s;
      xst1(fbits(s), rq, pub ? so * hb + (unsigned)((row * U + UW * slot + c) * 4) : OOB, coloc);
      xst1(SENT, rq, (pub && t >= 2) ? sr * hb + (unsigned)((row * U + UW * slot + c) * 4) : OOB, coloc);
    }
    SP_STAMP(5);
    // =========================== C: scores, partial context ========
    {
      const int NPC = U / 4;                           // <= NT (host check)
      const int qi = min(tid, NPC - 1);
      const unsigned off = so * hb + (unsigned)((ci * U + 4 * qi) * 4);
      u32x4 v;
      Spin g;
      g.start();
      for (;;) {
        v = xld4(rq, off);
        if (__all(!has_sentinel(v))) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
      if (tid < NPC) *reinterpret_cast<f32x4 *>(qs + 4 * qi) = __builtin_bit_cast(f32x4, v);
    }
    SP_STAMP(6);
    if (LOC && t > 0) {
      // previous alignments of my utterance (published in duty D of the step before: long there)
      const int NPC = TeP / 4;                         // <= NT (host check)
      const int qi = min(tid, NPC - 1);
      const unsigned off = sp * lb + (unsigned)((ci * TeP + 4 * qi) * 4);
      u32x4 v;
      Spin g;
      g.start();
      for (;;) {
        v = xld4(rl, off);
        if (__all(!has_sentinel(v))) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
      if (tid < NPC) {
        const f32x4 fv = __builtin_bit_cast(f32x4, v);
        float *d = alp_s + pbc + 4 * qi;
        d[0] = fv.x; d[1] = fv.y; d[2] = fv.z; d[3] = fv.w;
      }
    }
    __syncthreads();
    if (flag[0]) return;
    SP_STAMP(12);
    if (LOC) {
      // location features of my frames: cf[f][j] = sum_d a_prev[f0 + f + d - pb] * ck[d][j] ('same' padding)
      for (int i = tid; i < FS * Fc; i += NT) {
        const int f = i / Fc, j = i % Fc;
        const int K4 = (Kc + 3) & ~3;
        const float *a = alp_s + f0 + f;                 // (not 16-byte aligned in general: scalar reads of a, 16-byte of c)
        const f32x4 *c4 = reinterpret_cast<const f32x4 *>(ck_s + (size_t)j * K4);
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < K4; d += 4) {
          const f32x4 cc = c4[d / 4];
          acc4.x = fmaf(a[d], cc.x, acc4.x);
          acc4.y = fmaf(a[d + 1], cc.y, acc4.y);
          acc4.z = fmaf(a[d + 2], cc.z, acc4.z);
          acc4.w = fmaf(a[d + 3], cc.w, acc4.w);
        }
        cf_s[i] = (acc4.x + acc4.y) + (acc4.z + acc4.w);
      }
      __syncthreads();
    }
    SP_STAMP(13);
    const bool frozen = t >= clen;
    // location-aware scores of <= 32 frames on the matrix pipe (exact fp32, v_mfma_f32_16x16x4_f32):
    //   x[16 frames x 16 units] = features[16 x 12] . conv_proj[12 x 16] + (keys + q),  score[frame] = sum_u v[u] tanh(x)
    // accumulator register c of lane (kq, fl) = frame 4 kq + c of the tile, unit fl: keys are read along the units
    // (no bank conflict), a wave owns every NW-th unit tile, the 16 unit lanes are added by DPP, the waves in LDS.
    // (the vector form below spends 10 filters x (1 + 4) LDS reads and 16 fma per 16 scores: 9.7 us of a 28 us step)
    if (LOC && FS <= 32 && (U & 15) == 0 && Fc <= 12) {
      const int fl = lane & 15, kq = lane >> 4;
      float a1[2][3];
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
          const int fr = 16 * ft + fl, fi = 4 * ks + kq;
          a1[ft][ks] = (fr < FS && fi < Fc) ? cf_s[fr * Fc + fi] : 0.f;
        }
      f32x4 sacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      // (filters >= F: the features' operand is zero there, so whatever finite row the clamped read brings is harmless)
      const int r0 = min(kq, Fc - 1) * U, r1 = min(4 + kq, Fc - 1) * U, r2 = min(8 + kq, Fc - 1) * U;
#pragma unroll 2
      for (int T = w; 16 * T < U; T += NW) {
        const int u = 16 * T + fl;
        const float qq = qs[u], vv = v_s[u];
        const float b1[3] = {wf_s[r0 + u], wf_s[r1 + u], wf_s[r2 + u]};
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          f32x4 x;
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] = keys_s[(size_t)min(16 * ft + 4 * kq + c, FS - 1) * U + u] + qq;
#pragma unroll
          for (int ks = 0; ks < 3; ++ks) x = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ft][ks], b1[ks], x, 0, 0, 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) sacc[ft][c] = fmaf(vv, ftanh_s(x[c]), sacc[ft][c]);
        }
      }
#pragma unroll
      for (int ft = 0; ft < 2; ++ft)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float tot = rowsum(sacc[ft][c]);
          if (fl == 0) es[w * 64 + 16 * ft + 4 * kq + c] = tot;
        }
      __syncthreads();
      if (tid < FS) {
        float tot = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tot += es[ww * 64 + tid];
        sc_s[tid] = (f0 + tid < cn) ? tot : -INFINITY;
      }
    } else
    for (int fg = 0; fg < FS; fg += 4 * NW) {      // four frames of a wave at a time: independent chains
      float sacc[4] = {0.f, 0.f, 0.f, 0.f};
      const f32x4 *q4 = reinterpret_cast<const f32x4 *>(qs), *v4 = reinterpret_cast<const f32x4 *>(v_s);
      for (int u4 = lane; u4 < U / 4; u4 += 64) {
        const f32x4 qq = q4[u4], vv = v4[u4];
        f32x4 kx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) kx[i] = reinterpret_cast<const f32x4 *>(keys_s + (size_t)min(fg + w + NW * i, FS - 1) * U)[u4];
        if (LOC)
          for (int j = 0; j < Fc; ++j) {       // one read of the projection row per filter, four frames on it
            const f32x4 wj = reinterpret_cast<const f32x4 *>(wf_s + (size_t)j * U)[u4];
#pragma unroll
            for (int i = 0; i < 4; ++i) kx[i] += cf_s[min(fg + w + NW * i, FS - 1) * Fc + j] * wj;
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4 kk = kx[i];
          sacc[i] = fmaf(vv.x, ftanh_s(kk.x + qq.x), sacc[i]);
          sacc[i] = fmaf(vv.y, ftanh_s(kk.y + qq.y), sacc[i]);
          sacc[i] = fmaf(vv.z, ftanh_s(kk.z + qq.z), sacc[i]);
          sacc[i] = fmaf(vv.w, ftanh_s(kk.w + qq.w), sacc[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int f = fg + w + NW * i;
        const float tot = wsum(sacc[i]);
        if (lane == 0 && f < FS) sc_s[f] = (f0 + f < cn) ? tot : -INFINITY;
      }
    }
    __syncthreads();
    SP_STAMP(7);
    // local softmax statistics, by every wave for itself (FS <= 64): no further barrier
    float m_loc, z_loc;
    float *es_w = es + w * 64;
    {
      const float sc = lane < FS ? sc_s[lane] : -INFINITY;
      const float m = wmax(fmaxf(sc, -3.0e38f));
      const float e = lane < FS ? __expf(sc - m) : 0.f;
      es_w[lane] = e;
      m_loc = m;
      z_loc = wsum(e);
    }
    {
      // partial context: thread = 4 columns
      for (int c4 = tid; c4 < E / 4; c4 += NT) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (!frozen && !p.stream_vals) {
          for (int f = 0; f < FS; ++f) a += es_w[f] * *reinterpret_cast<const f32x4 *>(vals_s + (size_t)f * E + 4 * c4);
        } else if (!frozen) {
          // the values of my frames come from the XCD's L2 (they do not fit the LDS next to the keys): 5 loads in flight
          const float *vp = p.values + ((size_t)cbg * Te + f0) * E + 4 * c4;
          const int nf = min(FS, max(cn - f0, 0));        // masked frames carry weight 0: not read
          for (int fb = 0; fb < nf; fb += 5) {
            f32x4 vv[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) vv[i] = *reinterpret_cast<const f32x4 *>(vp + (size_t)min(fb + i, nf - 1) * E);
#pragma unroll
            for (int i = 0; i < 5; ++i)
              if (fb + i < nf) a += es_w[fb + i] * vv[i];
          }
        }
        xst4(__builtin_bit_cast(u32x4, a), rp, so * pb + (unsigned)(((ci * S + cs) * (E + 4) + 4 * c4) * 4), coloc);
        xst4(sent4, rp, t >= 2 ? sr * pb + (unsigned)(((ci * S + cs) * (E + 4) + 4 * c4) * 4) : OOB, coloc);
      }
      if (tid == 0) {
        const f32x4 ms = {frozen ? -3.0e38f : m_loc, frozen ? 0.f : z_loc, 0.f, 0.f};
        xst4(__builtin_bit_cast(u32x4, ms), rp, so * pb + (unsigned)(((ci * S + cs) * (E + 4) + E) * 4), coloc);
        xst4(sent4, rp, t >= 2 ? sr * pb + (unsigned)(((ci * S + cs) * (E + 4) + E) * 4) : OOB, coloc);
      }
    }
    SP_STAMP(8);
    // =========================== D: combine ========================
    {
      // pieces: S x CB/4 of my column block, then S (m, z) records
      const int PC = CB / 4, NPC = S * PC + S;         // <= 2 * NT (host check)
      u32x4 v[2];
      unsigned off[2];
      int ii[2], c4[2];
      bool rec[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int qi = min(j * NT + tid, NPC - 1);
        rec[j] = qi >= S * PC;
        ii[j] = rec[j] ? qi - S * PC : qi / PC;
        c4[j] = rec[j] ? 0 : qi % PC;
        off[j] = so * pb + (unsigned)(((ci * S + ii[j]) * (E + 4) + (rec[j] ? E : cs * CB + 4 * c4[j])) * 4);
      }
      Spin g;
      g.start();
      for (;;) {
        v[0] = xld4(rp, off[0]);
        v[1] = xld4(rp, off[1]);
        if (__all(!has_sentinel(v[0]) && !has_sentinel(v[1]))) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (j * NT + tid < NPC) {
          const f32x4 fv = __builtin_bit_cast(f32x4, v[j]);
          if (rec[j]) { mz[2 * ii[j]] = fv.x; mz[2 * ii[j] + 1] = fv.y; }
          else        *reinterpret_cast<f32x4 *>(parts + ii[j] * CB + 4 * c4[j]) = fv;
        }
    }
    SP_STAMP(9);
    __syncthreads();
    if (flag[0]) return;
    {
      float M = -3.0e38f, Z = 0.f, fac[S];
#pragma unroll
      for (int i = 0; i < S; ++i) M = fmaxf(M, mz[2 * i]);
#pragma unroll
      for (int i = 0; i < S; ++i) {
        fac[i] = __expf(mz[2 * i] - M);
        Z += fac[i] * mz[2 * i + 1];
      }
      const float inv = __builtin_amdgcn_rcpf(Z);
      if (tid < CB) {
        float c = 0.f;
#pragma unroll
        for (int i = 0; i < S; ++i) c = fmaf(fac[i] * inv, parts[i * CB + tid], c);
        if (frozen) c = ctx_prev;
        ctx_prev = c;
      }
      xst1(fbits(ctx_prev), rc, tid < CB ? so * cb + (unsigned)((ci * E + cs * CB + tid) * 4) : OOB, coloc);
      xst1(SENT, rc, (tid < CB && t >= 2) ? sr * cb + (unsigned)((ci * E + cs * CB + tid) * 4) : OOB, coloc);
      if (tid < FS) {
        float a = frozen ? al_prev : es[tid] * __expf(m_loc - M) * inv;     // (tid < FS <= 64: wave 0's copy)
        if (!frozen && f0 + tid >= cn) a = 0.f;
        al_prev = a;
      }
      if (LOC) {
        xst1(fbits(al_prev), rl, tid < FS ? so * lb + (unsigned)((ci * TeP + f0 + tid) * 4) : OOB, coloc);
        xst1(SENT, rl, (tid < FS && t >= 2) ? sr * lb + (unsigned)((ci * TeP + f0 + tid) * 4) : OOB, coloc);
      }
    }
    SP_STAMP(10);
    // =========================== E: scheduled sampling of the next input ===========================
    // ScheduledEmbeddingTrainingHelper (rnn_decoder.py:59-66): with probability sprob the next input of a row is a
    // sample of softmax(logits_t), logits_t = [ho_t | ctx_t] . Wout + b — exactly the draws of sample_ids_kernel
    // (elementwise.hip) for (seed, offset + t, batch row).  The first slice's workgroup of an utterance decides for
    // its row: it gathers the row of [ho | ctx] only when the row was selected, every workgroup of the unit learns the
    // four inputs through a 16-byte ring at the next step's gate phase.
    if (samp && t + 1 < L) {
      __syncthreads();                                   // parts / mz have been read: aux is free
      if (cs == 0) {
        // (the two Philox words of (seed, offset + t, row) were drawn before the launch: sdraw)
        const uint2 rr = *reinterpret_cast<const uint2 *>(p.sdraw + 2 * ((size_t)t * B + cbg));
        int id = p.ids[(size_t)(t + 1) * B + cbg];       // teacher forcing
        if (u01(rr.x) < p.sprob) {
          SampleRow q;
          q.rc = rc; q.rh = rh;
          q.hoff = so * hb + (unsigned)(ci * U * 4); q.coff = so * cb + (unsigned)(ci * E * 4);
          q.wout = p.wout; q.bout = p.bout; q.C = p.C; q.U = U; q.E = E;
          q.dscale = drop ? p.dscale + ((size_t)t * B + cbg) * U : nullptr;
          q.aux = aux; q.flag = flag; q.status = p.status; q.timeout_ticks = p.timeout_ticks; q.ry = rr.y; q.teacher = id;
          id = sample_row(q);
          if (flag[0]) return;
        }
        if (tid == 0) {
          p.ids_w[(size_t)(t + 1) * B + cbg] = id;
          xst1((unsigned)id, ri, so * 16 + (unsigned)(ci * 4), coloc);
          xst1(SENT, ri, t >= 2 ? sr * 16 + (unsigned)(ci * 4) : OOB, coloc);
        }
      }
    }
    __syncthreads();     // the scratch is re-used by the next step's gather
  }
  save_step(L - 1);
}


// ===========================================================================================================
// Backward pass of the step loop, same decomposition (one XCD per 4 utterances, steps t = L-1 .. 0):
//   D1  attention backward of (utterance i, frame slice s): d context = dCtx[t] (output projection) + the carry of
//       step t+1 -> d alignment of its frames -> softmax backward (sum_f a da = dctx . context_t) -> through
//       v.tanh(keys + q): its part of dq -> publishes dq_part[U]; d keys of ITS frames accumulate in registers for the
//       whole launch, d v in registers
//   D1b the 8 slices of an utterance each add a U/8 block of dq                       -> publishes dq block
//   D2  workgroup j: dh of its 16 units = dH[t] + carry + dq . Wq^T (matrix pipe), LSTM cell backward
//                                                                                       -> publishes dz (gate-major)
//   D3  [d ctx_{t-1} | d h_{t-1}] = dz . [Kx^T | Kh^T]: its (E+U)/32 output columns, the transposed cell kernel in
//       REGISTERS (the same 12.6 MB per XCD as the forward pass), k phases in the blocks of v_mfma_f32_4x4x1
//                                                                                       -> publishes the carry
// Rings and hand-back rule as in the forward kernel (every publisher has gathered, one duty earlier at the latest,
// pieces that all 32 workgroups published after reading what it now resets).
struct BArgs {
  int L, U, E, Te, FS;
  int Bt, b0;              // rows of the whole batch (strides of the time-major tensors); first row of this launch
  // location-aware attention (LOCB): filter taps, filters; values slice read from L2 per step; conv kernel [K][F],
  // feature projection [F][U]; outputs for attn_param_grads_kernel (d scores [L][B][Te], location features
  // [L][B][Te][F]) and the conv kernel's gradient, one partial row per (utterance, slice) [B*S][K*F]
  int K, F, stream_vals;
  const float *ck, *wf;
  float *ds_all, *cf_all, *dck_part;
  const int32_t *dec_len, *enc_len;
  const float *kxhT;       // [4U][E+U] (k = gate-major column of the cell kernel)
  const float *wq, *v, *keys, *values;
  const float *acts, *Cs, *q, *ctx, *align;     // saved by the forward pass (time-major)
  const float *dH;         // [L][B][U] d h_t of the output projection
  float keep;              // output dropout of the cell (1 = off): d h through the output = mask / keep * (dH + dq . Wq^T)
  const float *dscale;     // ... mask / keep [L][B][U], drawn in front of the launch (speller_randoms_kernel)
  float *dCtx;             // [L][B][E] in: the output projection's share; out: the whole d context_t
  float *dq, *dz;          // [L][B][U], [L][B][4U] (gate-major): inputs of the weight-gradient products
  float *dkeys;            // [B][Te][U]
  float *dv_part;          // [B*S][U]
  unsigned *table;
  char *xbuf;
  int *status;
  unsigned long long timeout_ticks;
  int dbg;
};

#define SPB_TIMEOUT()                                                                                       \
  do {                                                                                                      \
    if ((threadIdx.x & 63) == 0) {                                                                          \
      flag[0] = 1;                                                                                          \
      __hip_atomic_store(p.status, 2 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     \
    }                                                                                                       \
  } while (0)

// NABU_PERSIST_DEBUG bit 2: wall-clock stamps of the phases of step L/2 in block 0 (status[48 + i], 10 ns ticks)
#define SPB_STAMP(i)                                                               \
  do {                                                                             \
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && n == L / 2)                  \
      p.status[48 + (i)] = (int)wall_clock64();                                    \
  } while (0)

struct SpinB {
  unsigned long long t0;
  unsigned n;
  __device__ __forceinline__ void start() { n = 0; }
  __device__ __forceinline__ bool expired(const BArgs &p) {
    if (n == 0) t0 = wall_clock64();
    if ((++n & 31u) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    return wall_clock64() - t0 > p.timeout_ticks;
  }
};

// NSETC sets of NKQC instructions per wave (weight registers per lane: NSETC * NKQC); KS = k phases in the 16 blocks
// of an instruction (4: a set is 16 columns, 16: a set is 4 columns); DKR = frames per thread in D1
// LOCB: location-aware attention (attention.py:186-292).  The score takes conv1d(a_{t-1})·conv_proj, so D1 also
//   * recomputes the location features of its frames from the saved alignments of step t-1,
//   * produces d features[f, c] = sum_u d[f, u] conv_proj[c, u] and hands them to the other slices of its utterance
//     through a fifth ring: step t-1 needs d a_{t-1}[j] += sum_{f,c} d features_t[f, c] ck[j - f + pb, c] for ITS
//     frames j (the filters reach (K-1)/2 frames into the neighbouring slices), and the softmax backward of step t-1
//     needs sum_j a_{t-1}[j] (that carry)[j] over ALL frames = sum_{f,c} d features_t[f, c] features_t[f, c]: every
//     slice appends its part of that sum to its ring piece, nobody needs the other slices' carries;
//   * accumulates the conv kernel's gradient of its frames in registers (K F / 256 per thread) for the whole launch;
//   * leaves d keys / d attention_v / d conv_proj to attn_param_grads_kernel (speller.hip), for which it saves the
//     step's d scores and location features as the step chain does — no per-frame accumulators here.
//   Thread map of the score backward: waves over frames, lanes over 16-byte unit groups (U <= 512), d features by
//   wave reductions.
template <int NSETC, int NKQC, int KS, int DKR, bool DROP, bool LOCB>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) void speller_persist_bwd_kernel(BArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int flag[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int unit, slot;     // decided at start-up: unit = the XCD this workgroup runs on, slot = its arrival rank there
  const int U = p.U, E = p.E, Te = p.Te, FS = p.FS, L = p.L;
  const int KB = 4 * U, KBW = KB / NW;          // reduction index of D3 (gate columns), range of a wave
  const int NC = (E + U) / P;                  // my output columns of D3
  const int UW = U / P, UB = U / S;            // my units (D2); my dq block (D1b)
  const int B = p.Bt;
  constexpr int CGS = 16 / KS;                 // column groups of 4 per instruction
  const int Kc = LOCB ? p.K : 0, Fc = LOCB ? p.F : 0, pbc = (Kc - 1) / 2, TeP = S * FS, K4 = (Kc + 3) & ~3;
  const int FF = FS * Fc, PF = (FF + 4) & ~3;  // d features of a slice; its ring piece (+ the partial sum, padded)
  constexpr int KR = NSETC * NKQC;
  const int NSET = NC / (4 * CGS);             // <= NSETC

  // Which XCD a block lands on is the dispatcher's business (observed: b % 8 in one launch, pairs of consecutive
  // blocks per XCD in another).  So the units are formed at run time: a workgroup joins the unit of the XCD it runs on
  // and takes the next free slot there (one agent-scope fetch-add).  One workgroup fits a CU (one wave per SIMD, most
  // of the register file), an XCD has 32 CUs and all 256 workgroups are resident: every XCD ends up with exactly 32.
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
  if (tid == 0) {
    flag[0] = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // counters start at 0xFFFFFFFF (the workspace prefill): the first arrival reads that and takes slot 0
    const unsigned old = __hip_atomic_fetch_add(p.table + 512 + (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = (int)(old + 1u);
    if (old + 1u >= (unsigned)P || xcc >= (unsigned)NU) {       // cannot happen on a whole MI355X: give up loudly
      flag[0] = 1;
      __hip_atomic_store(p.status, 3 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (flag[0]) return;          // an earlier launch on this workspace timed out / no slot
  unit = (int)xcc;
  slot = flag[1];
  __syncthreads();
  const bool coloc = !(p.dbg & 8);      // the unit shares one L2 by construction (bit 3: force write-through)

  // rings: carry [R][E+U], dq partials [R*S][U], dq [R][U], dz [R][4U]
  const unsigned kb = (unsigned)(R * (E + U) * 4), ab = (unsigned)(R * S * U * 4), qb = (unsigned)(R * U * 4), zb = (unsigned)(R * 4 * U * 4);
  const unsigned fbz = LOCB ? (unsigned)(R * S * PF * 4) : 0u;      // (LOCB) d features [R][S][PF]
  const size_t unit_bytes = (size_t)RING * (kb + ab + qb + zb + fbz);
  char *ub = p.xbuf + (size_t)unit * unit_bytes;
  __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)(RING * kb), 0x00020000);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * kb, 0, (int)(RING * ab), 0x00020000);
  __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (kb + ab), 0, (int)(RING * qb), 0x00020000);
  __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (kb + ab + qb), 0, (int)(RING * zb), 0x00020000);
  __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (kb + ab + qb + zb), 0, (int)(RING * fbz), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // LDS
  const bool svals = LOCB && p.stream_vals;           // the values slice does not fit next to the keys: read from L2
  float *keys_s = smem;                               // [FS][U]
  float *vals_s = keys_s + (size_t)FS * U;            // [FS][E] (absent when streamed)
  float *wqr_s = vals_s + (svals ? 0 : (size_t)FS * E);   // [UW][U + 4]: Wq rows of my units
  float *v_s = wqr_s + (size_t)UW * (U + 4);          // [U]
  // (LOCB) feature projection [F][U], conv kernel [F][K4], padded alignments of step t-1 [pb + S*FS + K + 8], features
  // and d features of my frames [FS][F], my frames' carry per filter [FS][F], every slice's d features [S*FS][F] and
  // partial sums [S]
  float *wf_s = v_s + U;
  float *ck_s = wf_s + (size_t)Fc * U;
  float *alp_s = ck_s + (size_t)K4 * Fc;
  float *cf_s = alp_s + (LOCB ? ((pbc + TeP + Kc + 8) & ~3) : 0);
  float *dcf_s = cf_s + ((FF + 3) & ~3);
  float *cpart_s = dcf_s + ((FF + 3) & ~3);
  float *dcfa_s = cpart_s + ((FF + 3) & ~3);
  float *rpart_s = dcfa_s + (size_t)TeP * Fc;
  float *scr = rpart_s + (LOCB ? 2 * S : 0);          // scratch
  // D3: my wave stages dz[4 rows][half of its k range] (two halves);  D1: dcx [E], red [NW] + da/ds [FS], dqh [2][U]
  // D1b: blk [S][UB];  D2: dqs [R][U + 4], chs [R][UW], qred [NW][64][4]
  const int HK = KBW / 2;
  float *Zs = scr + (size_t)w * R * HK;
  float *dcx = scr;
  float *redw = scr + E;                              // [NW] + [FS] + [FS]
  float *dqh = redw + 64 + 2 * 64;                    // [2][U] ([NW][U]: LOCB)
  float *blk = scr;
  float *dqs = scr;
  float *chs = dqs + R * (U + 4);
  float *qred = chs + 64;
  float *ored = scr + (size_t)NW * R * HK;            // D3 output tiles [NW][NC][4] (behind the staging)

  const int ci = slot / S, cs = slot % S, cbg = p.b0 + unit * R + ci, f0 = cs * FS;
  for (int i = tid; i < FS * U; i += NT) {
    const int f = f0 + i / U;
    keys_s[i] = f < Te ? p.keys[((size_t)cbg * Te + f) * U + i % U] : 0.f;
  }
  if (!svals)
    for (int i = tid; i < FS * E; i += NT) {
      const int f = f0 + i / E;
      vals_s[i] = f < Te ? p.values[((size_t)cbg * Te + f) * E + i % E] : 0.f;
    }
  if (LOCB) {
    for (int i = tid; i < Fc * U; i += NT) wf_s[i] = p.wf[i];
    for (int i = tid; i < K4 * Fc; i += NT) {            // [filter][tap], taps padded with zeros
      const int j = i / K4, d = i % K4;
      ck_s[i] = d < Kc ? p.ck[d * Fc + j] : 0.f;
    }
    for (int i = tid; i < ((pbc + TeP + Kc + 8) & ~3); i += NT) alp_s[i] = 0.f;       // the pads stay zero
  }
  for (int i = tid; i < UW * U; i += NT) wqr_s[(i / U) * (U + 4) + i % U] = p.wq[(size_t)(UW * slot + i / U) * U + i % U];
  for (int i = tid; i < U; i += NT) v_s[i] = p.v[i];

  // D3 weights: lane (cg, ks, j) of set st, instruction kq of my wave: kxhT[k = w*KBW + KS*kq + ks][NC*slot + 4*CGS*st + 4*cg + j]
  const int mcg = lane / (4 * KS), mks = (lane >> 2) % KS, mj = lane & 3;
  const int NKQ = KBW / KS;                       // instructions per set and wave (<= NKQC, even)
  float Wr[KR];
#pragma unroll
  for (int r = 0; r < KR; ++r) {
    const int st = r / NKQC, kq = r % NKQC;
    const int k = w * KBW + KS * kq + mks, col = 4 * CGS * st + 4 * mcg + mj;
    Wr[r] = (st < NSET && kq < NKQ && col < NC) ? p.kxhT[(size_t)k * (E + U) + NC * slot + col] : 0.f;
  }

  // D2 identities: gate thread (row, col = 4*unit + gate)
  const int grow = tid >> 6, gcol = tid & 63, gu = gcol >> 2, gg = gcol & 3;
  const bool gate_thr = gcol < 4 * UW;
  const int gb = p.b0 + unit * R + grow, gunit = UW * slot + gu;
  const int glen = p.dec_len[gb];
  float dc_state = 0.f;
  // D1 identities: thread = (4 units uq, frame half fh)
  const int U4 = U / 4;
  const int uq = tid % U4, fh = tid / U4;            // U4 <= NT; threads with fh >= NFH idle
  const int NFH = NT / U4 > 0 ? (NT / U4 < FS ? NT / U4 : FS) : 1;
  const int FPH = (FS + NFH - 1) / NFH;              // frames per thread
  const int clen = p.dec_len[cbg], cn = min(max(p.enc_len[cbg], 0), Te);
  f32x4 dk[DKR];                                      // d keys of my frames, my 4 units
#pragma unroll
  for (int i = 0; i < DKR; ++i) dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 dvacc = {0.f, 0.f, 0.f, 0.f};
  float dck_acc[4] = {0.f, 0.f, 0.f, 0.f};           // (LOCB) conv kernel gradient: elements tid + NT * m of [K][F]
  float dq_last = 0.f, dz_last = 0.f;
  f32x4 dcx_last = {0.f, 0.f, 0.f, 0.f};
  auto save_step = [&](int ts) {
    if (tid < UB) p.dq[((size_t)ts * B + cbg) * U + cs * UB + tid] = dq_last;
    if (gate_thr) p.dz[((size_t)ts * B + gb) * 4 * U + gg * U + gunit] = dz_last;
    if (cs == 0 && tid < E / 4) *reinterpret_cast<f32x4 *>(p.dCtx + ((size_t)ts * B + cbg) * E + 4 * tid) = dcx_last;
  };
  __syncthreads();

  for (int n = 0; n < L; ++n) {
    const int t = L - 1 - n;
    const unsigned so = (unsigned)(n % RING), sp = (unsigned)((n + RING - 1) % RING), sr = (unsigned)((n + RING - 2) % RING);
    const bool frozen = t >= clen;
    SPB_STAMP(0);
    // =========================== D1: attention backward ===========================
    {
      // d context of (step t, my utterance) and its product with the context (softmax backward's sum)
      f32x4 dc4 = {0.f, 0.f, 0.f, 0.f}, cx4 = {0.f, 0.f, 0.f, 0.f};
      if (tid < E / 4) {
        dc4 = *reinterpret_cast<const f32x4 *>(p.dCtx + ((size_t)t * B + cbg) * E + 4 * tid);
        cx4 = *reinterpret_cast<const f32x4 *>(p.ctx + ((size_t)(t + 1) * B + cbg) * E + 4 * tid);
      }
      if (n > 0) {
        const unsigned off = tid < E / 4 ? sp * kb + (unsigned)((ci * (E + U) + 4 * tid) * 4) : OOB;
        u32x4 v;
        SpinB g;
        g.start();
        for (;;) {
          v = xld4(rk, off);
          if (__all(off == OOB || !has_sentinel(v))) break;
          if (g.expired(p)) { SPB_TIMEOUT(); break; }
        }
        if (tid < E / 4) dc4 += __builtin_bit_cast(f32x4, v);
      }
      if (n > 0) save_step(t + 1);
      dcx_last = dc4;
      if (tid < E / 4) *reinterpret_cast<f32x4 *>(dcx + 4 * tid) = dc4;
      float rp = dc4.x * cx4.x + dc4.y * cx4.y + dc4.z * cx4.z + dc4.w * cx4.w;
      rp = wsum(rp);
      if (lane == 0) redw[w] = rp;
      if (LOCB) {
        if (n > 0) {
          // d features of step t+1, every slice of my utterance (+ their partial sums): S * PF / 4 pieces, <= 2 per thread
          const int NPC = S * PF / 4;
          u32x4 v[2];
          unsigned off[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) off[j] = sp * fbz + (unsigned)((ci * S * PF + 4 * min(j * NT + tid, NPC - 1)) * 4);
          SpinB g;
          g.start();
          for (;;) {
            v[0] = xld4(rf, off[0]);
            v[1] = xld4(rf, off[1]);
            if (__all(!has_sentinel(v[0]) && !has_sentinel(v[1]))) break;
            if (g.expired(p)) { SPB_TIMEOUT(); break; }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
            if (j * NT + tid < NPC) {
              const int qi = j * NT + tid, ii = 4 * qi / PF, e0 = 4 * qi % PF;     // slice, element inside its piece
              const f32x4 fv = __builtin_bit_cast(f32x4, v[j]);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const int e = e0 + k;
                if (e < FF) dcfa_s[ii * FF + e] = fv[k];
                else if (e == FF) rpart_s[ii] = fv[k];
              }
            }
        }
        // alignments of step t-1 (what this step's location features were computed from; zeros for t = 0)
        if (tid < TeP) alp_s[pbc + tid] = (tid < Te) ? p.align[((size_t)t * B + cbg) * Te + tid] : 0.f;
      }
    }
    SPB_STAMP(1);
    __syncthreads();
    if (flag[0]) return;
    if constexpr (LOCB) {
      // location features of my frames (as in the forward kernel) and, per filter, my frames' share of the carry
      // d a_t[j] = sum_{f', c} d features_{t+1}[f', c] ck[j - f' + pb][c]
      for (int i = tid; i < FF; i += NT) {
        const int f = i / Fc, c = i % Fc;
        const float *a = alp_s + f0 + f;
        const f32x4 *c4 = reinterpret_cast<const f32x4 *>(ck_s + (size_t)c * K4);
        f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < K4; d += 4) {
          const f32x4 cc = c4[d / 4];
          acc4.x = fmaf(a[d], cc.x, acc4.x);
          acc4.y = fmaf(a[d + 1], cc.y, acc4.y);
          acc4.z = fmaf(a[d + 2], cc.z, acc4.z);
          acc4.w = fmaf(a[d + 3], cc.w, acc4.w);
        }
        const float feat = (acc4.x + acc4.y) + (acc4.z + acc4.w);
        cf_s[i] = feat;
        if (f0 + f < Te) p.cf_all[(((size_t)t * B + cbg) * Te + f0 + f) * Fc + c] = feat;
        float cp = 0.f;
        if (n > 0) {
          const int j = f0 + f + pbc;                    // f' = j - d
          const int dlo = max(0, j - (TeP - 1)), dhi = min(Kc - 1, j);
          const float *cw = ck_s + (size_t)c * K4;
          for (int d = dlo; d <= dhi; ++d) cp = fmaf(dcfa_s[(j - d) * Fc + c], cw[d], cp);
        }
        cpart_s[i] = cp;
      }
      SPB_STAMP(2);
      __syncthreads();
      {
        float r = (redw[0] + redw[1]) + (redw[2] + redw[3]);
        if (n > 0)
#pragma unroll
          for (int i = 0; i < S; ++i) r += rpart_s[i];
        // d alignment of my live frames (masked frames have a = 0): thread = 16 bytes of the encoder dimension, five
        // frames' loads in flight (the values slice may come from L2), one wave reduction per frame
        const int nf = frozen ? 0 : min(FS, max(cn - f0, 0));
        float *wpart = dqh;                          // [NW][64]
        {
          constexpr int FB = 5;
          const bool mine = tid < E / 4;
          const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
          const f32x4 dcv = mine ? *reinterpret_cast<const f32x4 *>(dcx + 4 * tid) : z4;
          const float *vb = svals ? p.values + ((size_t)cbg * Te + f0) * E : vals_s;
          for (int fb = 0; fb < nf; fb += FB) {
            f32x4 vv[FB];
#pragma unroll
            for (int i = 0; i < FB; ++i) vv[i] = mine ? *reinterpret_cast<const f32x4 *>(vb + (size_t)min(fb + i, nf - 1) * E + 4 * tid) : z4;
#pragma unroll
            for (int i = 0; i < FB; ++i) {
              const f32x4 m4 = dcv * vv[i];
              const float tot = wsum((m4.x + m4.y) + (m4.z + m4.w));
              if (lane == 0 && fb + i < nf) wpart[w * 64 + fb + i] = tot;
            }
          }
        }
        __syncthreads();
        if (tid < FS) {
          const int f = tid;
          const bool live = f < nf;
          float da = 0.f;
          if (live)
            for (int ww = 0; ww < NW; ++ww) da += wpart[ww * 64 + f];
          for (int c = 0; c < Fc; ++c) da += cpart_s[f * Fc + c];
          const float a = live ? p.align[((size_t)(t + 1) * B + cbg) * Te + f0 + f] : 0.f;
          const float g = a * (da - r);            // d score
          redw[64 + f] = g;
          if (f0 + f < Te) p.ds_all[((size_t)t * B + cbg) * Te + f0 + f] = g;
        }
      }
      SPB_STAMP(3);
      __syncthreads();
      {
        // through v . tanh(keys + q + features . conv_proj): waves over frames, lanes over 16-byte unit groups
        constexpr int MJ = 2, MF = 12;             // U / 4 <= 128 groups; filters (host check)
        f32x4 dq_l[MJ], qq[MJ], vv[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int u4 = lane + 64 * j;
          dq_l[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          qq[j] = u4 < U4 ? *reinterpret_cast<const f32x4 *>(p.q + ((size_t)t * B + cbg) * U + 4 * u4) : dq_l[j];
          vv[j] = u4 < U4 ? *reinterpret_cast<const f32x4 *>(v_s + 4 * u4) : dq_l[j];
        }
        // two frames of a wave at a time: every projection row read serves both
        for (int f = w; f < FS; f += 2 * NW) {
          const int fB = min(f + NW, FS - 1);
          const bool hasB = f + NW < FS;
          const float gA = redw[64 + f], gB = hasB ? redw[64 + fB] : 0.f;
          float dcfA[MF], dcfB[MF];
#pragma unroll
          for (int c = 0; c < MF; ++c) dcfA[c] = dcfB[c] = 0.f;
#pragma unroll
          for (int j = 0; j < MJ; ++j) {
            const int u4 = lane + 64 * j;
            if (u4 < U4) {
              f32x4 xA = *reinterpret_cast<const f32x4 *>(keys_s + (size_t)f * U + 4 * u4) + qq[j];
              f32x4 xB = *reinterpret_cast<const f32x4 *>(keys_s + (size_t)fB * U + 4 * u4) + qq[j];
#pragma unroll
              for (int c = 0; c < MF; ++c)          // (unrolled: the LDS reads of all filters are in flight together)
                if (c < Fc) {
                  const f32x4 wc = *reinterpret_cast<const f32x4 *>(wf_s + (size_t)c * U + 4 * u4);
                  xA += cf_s[f * Fc + c] * wc;
                  xB += cf_s[fB * Fc + c] * wc;
                }
              f32x4 th, ddA, ddB;
              th.x = ftanh_s(xA.x); th.y = ftanh_s(xA.y); th.z = ftanh_s(xA.z); th.w = ftanh_s(xA.w);
              ddA.x = gA * vv[j].x * (1.f - th.x * th.x); ddA.y = gA * vv[j].y * (1.f - th.y * th.y);
              ddA.z = gA * vv[j].z * (1.f - th.z * th.z); ddA.w = gA * vv[j].w * (1.f - th.w * th.w);
              th.x = ftanh_s(xB.x); th.y = ftanh_s(xB.y); th.z = ftanh_s(xB.z); th.w = ftanh_s(xB.w);
              ddB.x = gB * vv[j].x * (1.f - th.x * th.x); ddB.y = gB * vv[j].y * (1.f - th.y * th.y);
              ddB.z = gB * vv[j].z * (1.f - th.z * th.z); ddB.w = gB * vv[j].w * (1.f - th.w * th.w);
              dq_l[j] += ddA + ddB;
#pragma unroll
              for (int c = 0; c < MF; ++c)
                if (c < Fc) {
                  const f32x4 wc = *reinterpret_cast<const f32x4 *>(wf_s + (size_t)c * U + 4 * u4);
                  dcfA[c] = fmaf(ddA.x, wc.x, fmaf(ddA.y, wc.y, fmaf(ddA.z, wc.z, fmaf(ddA.w, wc.w, dcfA[c]))));
                  dcfB[c] = fmaf(ddB.x, wc.x, fmaf(ddB.y, wc.y, fmaf(ddB.z, wc.z, fmaf(ddB.w, wc.w, dcfB[c]))));
                }
            }
          }
#pragma unroll
          for (int c = 0; c < MF; ++c)
            if (c < Fc) {
              const float tA = wsum(dcfA[c]), tB = wsum(dcfB[c]);
              if (lane == 0) {
                dcf_s[f * Fc + c] = tA;
                if (hasB) dcf_s[fB * Fc + c] = tB;
              }
            }
        }
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int u4 = lane + 64 * j;
          if (u4 < U4) *reinterpret_cast<f32x4 *>(dqh + (size_t)w * U + 4 * u4) = dq_l[j];
        }
      }
      SPB_STAMP(4);
      __syncthreads();
      if (w == 0) {
        // my ring piece: d features of my frames + my part of sum_{f,c} d features . features
        float rp = 0.f;
        for (int i = lane; i < FF; i += 64) rp = fmaf(dcf_s[i], cf_s[i], rp);
        rp = wsum(rp);
        for (int q4 = lane; q4 < PF / 4; q4 += 64) {
          f32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int e = 4 * q4 + k;
            o[k] = e < FF ? dcf_s[e] : (e == FF ? rp : 0.f);
          }
          xst4(__builtin_bit_cast(u32x4, o), rf, so * fbz + (unsigned)(((ci * S + cs) * PF + 4 * q4) * 4), coloc);
          xst4(sent4, rf, n >= 2 ? sr * fbz + (unsigned)(((ci * S + cs) * PF + 4 * q4) * 4) : OOB, coloc);
        }
      }
      // conv kernel's gradient of my frames: d ck[d][c] += sum_f d features[f][c] a_{t-1}[f0 + f + d - pb]
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i = tid + NT * m;
        if (i < Kc * Fc) {
          const int d = i / Fc, c = i % Fc;
          float acc = dck_acc[m];
          for (int f = 0; f < FS; ++f) acc = fmaf(dcf_s[f * Fc + c], alp_s[f0 + f + d], acc);
          dck_acc[m] = acc;
        }
      }
    } else {
    {
      const float r = (redw[0] + redw[1]) + (redw[2] + redw[3]);
      // d alignment of my frames: wave w takes frames w, w + NW, ...
      for (int f = w; f < FS; f += NW) {
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
        for (int e4 = lane; e4 < E / 4; e4 += 64)
          a4 += *reinterpret_cast<const f32x4 *>(dcx + 4 * e4) * *reinterpret_cast<const f32x4 *>(vals_s + (size_t)f * E + 4 * e4);
        const float da = wsum((a4.x + a4.y) + (a4.z + a4.w));
        if (lane == 0) {
          const float a = (f0 + f < cn && !frozen) ? p.align[((size_t)(t + 1) * B + cbg) * Te + f0 + f] : 0.f;
          redw[64 + f] = a * (da - r);          // d score
        }
      }
    }
    __syncthreads();
    {
      // through v . tanh(keys + q): my 4 units, my half of the frames
      f32x4 dq4 = {0.f, 0.f, 0.f, 0.f};
      if (fh < NFH) {
        const f32x4 qq = *reinterpret_cast<const f32x4 *>(p.q + ((size_t)t * B + cbg) * U + 4 * uq);
        const f32x4 vv = *reinterpret_cast<const f32x4 *>(v_s + 4 * uq);
#pragma unroll
        for (int i = 0; i < DKR; ++i) {
          const int f = fh * FPH + i;
          if (i < FPH && f < FS) {
            const float g = redw[64 + f];
            const f32x4 kk = *reinterpret_cast<const f32x4 *>(keys_s + (size_t)f * U + 4 * uq);
            f32x4 th, dd;
            th.x = ftanh_s(kk.x + qq.x); th.y = ftanh_s(kk.y + qq.y); th.z = ftanh_s(kk.z + qq.z); th.w = ftanh_s(kk.w + qq.w);
            dd.x = g * vv.x * (1.f - th.x * th.x); dd.y = g * vv.y * (1.f - th.y * th.y);
            dd.z = g * vv.z * (1.f - th.z * th.z); dd.w = g * vv.w * (1.f - th.w * th.w);
            dq4 += dd;
            dk[i] += dd;
            dvacc += g * th;
          }
        }
        *reinterpret_cast<f32x4 *>(dqh + (size_t)fh * U + 4 * uq) = dq4;
      }
    }
    __syncthreads();
    }
    if (tid < U4) {
      const int NH = LOCB ? NW : NFH;
      f32x4 s4 = *reinterpret_cast<const f32x4 *>(dqh + 4 * tid);
      for (int h = 1; h < NH; ++h) s4 += *reinterpret_cast<const f32x4 *>(dqh + (size_t)h * U + 4 * tid);
      xst4(__builtin_bit_cast(u32x4, s4), ra, so * ab + (unsigned)(((ci * S + cs) * U + 4 * tid) * 4), coloc);
      xst4(sent4, ra, n >= 2 ? sr * ab + (unsigned)(((ci * S + cs) * U + 4 * tid) * 4) : OOB, coloc);
    }
    SPB_STAMP(5);
    // =========================== D1b: my block of dq ===========================
    {
      const int PC = UB / 4, NPC = S * PC;            // <= NT (host check)
      const int qi = min(tid, NPC - 1), ii = qi / PC, c4 = qi % PC;
      const unsigned off = so * ab + (unsigned)(((ci * S + ii) * U + cs * UB + 4 * c4) * 4);
      u32x4 v;
      SpinB g;
      g.start();
      for (;;) {
        v = xld4(ra, off);
        if (__all(!has_sentinel(v))) break;
        if (g.expired(p)) { SPB_TIMEOUT(); break; }
      }
      __syncthreads();        // dqh has been read by everybody: blk may overwrite the scratch
      if (tid < NPC) *reinterpret_cast<f32x4 *>(blk + ii * UB + 4 * c4) = __builtin_bit_cast(f32x4, v);
    }
    __syncthreads();
    if (flag[0]) return;
    {
      float s = 0.f;
      if (tid < UB)
        for (int i = 0; i < S; ++i) s += blk[i * UB + tid];
      dq_last = s;
      xst1(fbits(s), rq, tid < UB ? so * qb + (unsigned)((ci * U + cs * UB + tid) * 4) : OOB, coloc);
      xst1(SENT, rq, (tid < UB && n >= 2) ? sr * qb + (unsigned)((ci * U + cs * UB + tid) * 4) : OOB, coloc);
    }
    __syncthreads();          // blk read: the scratch is free for dq / carry
    SPB_STAMP(6);
    // =========================== D2: dq . Wq^T, cell backward ===================
    {
      const int NPC = R * U / 4;                       // <= 2 * NT
      const int q0 = min(tid, NPC - 1), q1 = min(NT + tid, NPC - 1);
      // my units' carry of d h: [row][E + UW*slot .. + UW): UW/4 pieces per row (UW % 4 == 0) or single words
      const int CP = (UW + 3) / 4;
      const bool cthr = tid < R * CP && n > 0;
      const int crow = tid / CP, cpc = tid % CP;
      const unsigned coff = cthr ? sp * kb + (unsigned)((crow * (E + U) + E + UW * slot + 4 * cpc) * 4) : OOB;
      u32x4 v0, v1, vc;
      SpinB g;
      g.start();
      for (;;) {
        v0 = xld4(rq, so * qb + (unsigned)(q0 * 16));
        v1 = xld4(rq, so * qb + (unsigned)(q1 * 16));
        vc = xld4(rk, coff);
        bool ok = !has_sentinel(v0) && !has_sentinel(v1);
        if (cthr) {      // only the words of my units count (UW may not fill the last piece)
          const int nw = min(4, UW - 4 * cpc);
          ok = ok && vc.x != SENT && (nw < 2 || vc.y != SENT) && (nw < 3 || vc.z != SENT) && (nw < 4 || vc.w != SENT);
        }
        if (__all(ok)) break;
        if (g.expired(p)) { SPB_TIMEOUT(); break; }
      }
      if (tid < NPC) *reinterpret_cast<f32x4 *>(dqs + (4 * q0 / U) * (U + 4) + 4 * q0 % U) = __builtin_bit_cast(f32x4, v0);
      if (NT + tid < NPC) *reinterpret_cast<f32x4 *>(dqs + (4 * q1 / U) * (U + 4) + 4 * q1 % U) = __builtin_bit_cast(f32x4, v1);
      if (tid < R * CP) {
        const f32x4 fc = cthr ? __builtin_bit_cast(f32x4, vc) : (f32x4){0.f, 0.f, 0.f, 0.f};
        float *d = chs + crow * 16 + 4 * cpc;
        d[0] = fc.x; d[1] = fc.y; d[2] = fc.z; d[3] = fc.w;
      }
    }
    SPB_STAMP(7);
    __syncthreads();
    if (flag[0]) return;
    {
      // dhq[4 rows][my UW units] = dq . Wq^T on the matrix pipe: blocks (unit group cg, k phase ks) as in the forward duty B
      const int bcg = lane >> 4, bks = (lane >> 2) & 3, bj = lane & 3;
      f32x4 qa = {0.f, 0.f, 0.f, 0.f}, qbb = {0.f, 0.f, 0.f, 0.f};
      {
        const float *ha = dqs + bj * (U + 4) + w * (U / 4) + bks;
        const bool colok = 4 * bcg + bj < UW;
        const float *wb = wqr_s + (size_t)min(4 * bcg + bj, UW - 1) * (U + 4) + w * (U / 4) + bks;
        for (int kk = 0; kk < U / 16; kk += 2) {
          const float a0 = ha[4 * kk], a1 = ha[4 * kk + 4];
          const float b0 = colok ? wb[4 * kk] : 0.f, b1 = colok ? wb[4 * kk + 4] : 0.f;
          qa = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, qa, 0, 0, 0);
          qbb = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, qbb, 0, 0, 0);
        }
      }
      f32x4 q4 = qa + qbb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        q4[i] += __shfl_xor(q4[i], 4);
        q4[i] += __shfl_xor(q4[i], 8);
      }
      if (bks == 0) *reinterpret_cast<f32x4 *>(qred + ((size_t)w * 16 + 4 * bcg + bj) * 4) = q4;   // [wave][unit][row]
    }
    __syncthreads();
    {
      float dzv = 0.f;
      if (gate_thr) {
        const float dhq = (qred[(0 * 16 + gu) * 4 + grow] + qred[(1 * 16 + gu) * 4 + grow]) +
                          (qred[(2 * 16 + gu) * 4 + grow] + qred[(3 * 16 + gu) * 4 + grow]);
        const size_t sidx = ((size_t)t * B + gb) * U + gunit;
        const float a = p.acts[((size_t)t * B + gb) * 4 * U + gg * U + gunit];
        const float gi = quad_bcast(a, 0), gj = quad_bcast(a, 1), gf = quad_bcast(a, 2), go = quad_bcast(a, 3);
        if (t < glen) {
          float dho = p.dH[sidx] + dhq;          // gradient of the cell OUTPUT (projection + query): through the dropout mask
          if (DROP && p.keep < 1.f) dho *= p.dscale[sidx];      // (the forward call's scale factors, [L, B, U])
          const float dh = dho + chs[grow * 16 + gu];          // + the recurrent carry (not dropped)
          const float cnew = p.Cs[sidx + (size_t)B * U], cprev = p.Cs[sidx];
          const float tc = ftanh(cnew);
          const float dct = dc_state + dh * go * (1.f - tc * tc);
          dzv = gg == 0 ? dct * gj * gi * (1.f - gi)
              : gg == 1 ? dct * gi * (1.f - gj * gj)
              : gg == 2 ? dct * cprev * gf * (1.f - gf)
                        : dh * tc * go * (1.f - go);
          dc_state = dct * gf;
        }
      }
      dz_last = dzv;
      xst1(fbits(dzv), rz, gate_thr ? so * zb + (unsigned)((grow * 4 * U + gg * U + gunit) * 4) : OOB, coloc);
      xst1(SENT, rz, (gate_thr && n >= 2) ? sr * zb + (unsigned)((grow * 4 * U + gg * U + gunit) * 4) : OOB, coloc);
    }
    __syncthreads();          // dqs / qred read: the scratch is free for the staged dz
    SPB_STAMP(8);
    // =========================== D3: dz . [Kx^T | Kh^T] ==========================
    f32x4 acc[NSETC];
#pragma unroll
    for (int st = 0; st < NSETC; ++st) acc[st] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t > 0) {      // (the carry of step 0 has no consumer)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // gather dz[4 rows][my k range, this half]: R * HK / 4 pieces, NQ per lane
        constexpr int NQ = (R * KS * NKQC / 2 / 4 + 63) / 64;      // pieces per lane: R * HK / 4 <= 64 * NQ
        const int PR = HK / 4, NPC = R * PR;
        u32x4 v[NQ];
        unsigned off[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int qi = min(64 * i + lane, NPC - 1);
          off[i] = so * zb + (unsigned)(((qi / PR) * KB + w * KBW + half * HK + 4 * (qi % PR)) * 4);
        }
        SpinB g;
        g.start();
        for (;;) {
          bool ok = true;
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            v[i] = xld4(rz, off[i]);
            ok = ok && !has_sentinel(v[i]);
          }
          if (__all(ok)) break;
          if (g.expired(p)) { SPB_TIMEOUT(); break; }
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int qi = 64 * i + lane;
          if (qi < NPC) *reinterpret_cast<f32x4 *>(Zs + (qi / PR) * HK + 4 * (qi % PR)) = __builtin_bit_cast(f32x4, v[i]);
        }
        // products of this half: instruction kq covers k = KS*kq + ks; my A operand: dz[row j][k]
        const float *za = Zs + mj * HK + mks;
        const int kq0 = half * (NKQ / 2);
        if (NKQ == NKQC) {          // full-size shape: straight line
#pragma unroll
          for (int kk = 0; kk < NKQC / 2; ++kk) {
            const float a = za[KS * kk];
#pragma unroll
            for (int st = 0; st < NSETC; ++st)
              acc[st] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, Wr[st * NKQC + half * (NKQC / 2) + kk], acc[st], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int kq = 0; kq < NKQC; ++kq)
            if (kq >= kq0 && kq < kq0 + NKQ / 2) {
              const float a = za[KS * (kq - kq0)];
#pragma unroll
              for (int st = 0; st < NSETC; ++st) acc[st] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, Wr[st * NKQC + kq], acc[st], 0, 0, 0);
            }
        }
      }
      // add the k phases (lanes that differ in ks), then the waves
#pragma unroll
      for (int st = 0; st < NSETC; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 4; m < 4 * KS; m <<= 1) acc[st][i] += __shfl_xor(acc[st][i], m);
      if (mks == 0) {
#pragma unroll
        for (int st = 0; st < NSETC; ++st) {
          const int col = 4 * CGS * st + 4 * mcg + mj;
          if (st < NSET && col < NC) *reinterpret_cast<f32x4 *>(ored + ((size_t)w * NC + col) * 4) = acc[st];
        }
      }
    }
    SPB_STAMP(9);
    __syncthreads();
    if (flag[0]) return;
    if (t > 0) {
      // thread (row, column) adds the four waves and publishes the carry
      const int col = tid % NC, row = tid / NC;
      float s = 0.f;
      const bool pub = tid < R * NC;
      if (pub) s = (ored[((size_t)0 * NC + col) * 4 + row] + ored[((size_t)1 * NC + col) * 4 + row]) +
                   (ored[((size_t)2 * NC + col) * 4 + row] + ored[((size_t)3 * NC + col) * 4 + row]);
      xst1(fbits(s), rk, pub ? so * kb + (unsigned)((row * (E + U) + NC * slot + col) * 4) : OOB, coloc);
      xst1(SENT, rk, (pub && n >= 2) ? sr * kb + (unsigned)((row * (E + U) + NC * slot + col) * 4) : OOB, coloc);
    }
    SPB_STAMP(10);
    __syncthreads();
  }
  save_step(0);
  if constexpr (LOCB) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int i = tid + NT * m;
      if (i < Kc * Fc) p.dck_part[((size_t)cbg * S + cs) * Kc * Fc + i] = dck_acc[m];
    }
    return;
  }
  // d keys of my frames, d v partial row of (utterance, slice)
  if (fh < NFH) {
#pragma unroll
    for (int i = 0; i < DKR; ++i) {
      const int f = f0 + fh * FPH + i;
      if (i < FPH && fh * FPH + i < FS && f < Te) *reinterpret_cast<f32x4 *>(p.dkeys + ((size_t)cbg * Te + f) * U + 4 * uq) = dk[i];
    }
    *reinterpret_cast<f32x4 *>(dqh + (size_t)fh * U + 4 * uq) = dvacc;
  }
  __syncthreads();
  if (tid < U4) {
    f32x4 s4 = *reinterpret_cast<const f32x4 *>(dqh + 4 * tid);
    for (int h = 1; h < NFH; ++h) s4 += *reinterpret_cast<const f32x4 *>(dqh + (size_t)h * U + 4 * tid);
    *reinterpret_cast<f32x4 *>(p.dv_part + ((size_t)cbg * S + cs) * U + 4 * tid) = s4;
  }
}

int kr_for(int KW) { return KW <= 64 ? 64 : KW <= 192 ? 192 : 384; }
size_t lds_floats(const SpPersistDesc &d, int FS, bool stream);
// the values slice stays in LDS when it fits, else it is read from L2 every step
bool stream_values(const SpPersistDesc &d, int FS) {
  if (const char *e = getenv("NABU_SPELLER_STREAM_VALUES")) return atoi(e) != 0;     // (tests: force the streamed path)
  return lds_floats(d, FS, false) * 4 > 160 * 1024 - 512;
}
size_t lds_floats(const SpPersistDesc &d, int FS, bool stream) {
  const size_t K = d.E + d.U, UW = d.U / P;
  const size_t KR = kr_for((int)(K / NW));
  size_t scr = NW * R * KR;
  if (R * KR < 256) scr += NW * 256;
  const size_t need_c = (size_t)R * (d.U + 4) + NW * 64 + d.U + FS + 64, need_d = (size_t)S * (d.E / S) + 2 * S + 64;
  size_t aux = need_c > need_d ? need_c : need_d;
  const size_t need_e = d.sample_prob > 0.f ? K + NW * 64 + 64 + 16 : 0;     // duty E: a row of [h | ctx], partial logits, exps
  if (need_e > aux) aux = need_e;
  if (K / NW == KR) scr = scr > aux ? scr : aux;   // aliased
  else scr += aux;
  size_t loc = 0;
  if (d.kind == 1) {
    const size_t pb = (d.K - 1) / 2, TeP = (size_t)S * FS;
    loc = (size_t)d.F * d.U + (size_t)((d.K + 3) & ~3) * d.F + ((pb + TeP + d.K + 8) & ~3) + ((FS * d.F + 3) & ~3);
  }
  return (size_t)FS * d.U + (stream ? 0 : (size_t)FS * d.E) + (size_t)d.U * UW + d.U + NW * 64 + loc + scr + 64;
}
int frames_per_slice(const SpPersistDesc &d) { return (d.Te + S - 1) / S; }
size_t ring_bytes(const SpPersistDesc &d) {
  const size_t hb = (size_t)R * d.U * 4, cb = (size_t)R * d.E * 4, pb = (size_t)R * S * (d.E + 4) * 4;
  const size_t lb = d.kind == 1 ? (size_t)R * S * frames_per_slice(d) * 4 : 0;
  return (size_t)NU * RING * (2 * hb + cb + pb + lb + 16);
}

}  // namespace

// The persistent decoder is sized for the whole MI355X: NU units of P workgroups, one per CU, formed per XCC, each
// with up to 160 KiB of LDS.  On a partitioned (CPX/DPX) or smaller device the grid cannot be co-resident (every
// call would spin to the time-out), so the geometry is checked against the CURRENT device and the step chain of
// speller.hip is taken when it does not fit.
static bool device_fits() {
  static thread_local int cached_dev = -1;
  static thread_local bool cached = false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (dev != cached_dev) {
    int cus = 0, xcc = 0, lds = 0;
    bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= NU * P;
    ok = ok && hipDeviceGetAttribute(&xcc, hipDeviceAttributeNumberOfXccs, dev) == hipSuccess && xcc == NU;
    // the runtime reports the LDS a block may use under one of these names depending on its version
    const hipDeviceAttribute_t names[3] = {hipDeviceAttributeMaxSharedMemoryPerBlock, hipDeviceAttributeSharedMemPerBlockOptin,
                                           hipDeviceAttributeMaxSharedMemoryPerMultiprocessor};
    for (hipDeviceAttribute_t n : names) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, n, dev) == hipSuccess && v > lds) lds = v;
    }
    ok = ok && lds >= 160 * 1024 - 512;
    (void)hipGetLastError();
    if (getenv("NABU_PERSIST_DEBUG")) fprintf(stderr, "nabu: persistent decoder device check: %d CUs, %d XCCs, %d B LDS -> %s\n", cus, xcc, lds, ok ? "fits" : "step chain");
    cached = ok;
    cached_dev = dev;
  }
  return cached;
}

static bool shape_ok(const SpPersistDesc &d) {
  if ((d.B != NU * R && d.B != 2 * NU * R) || d.U % 32 || d.E % 32 || d.U < 32 || d.E < 32) return false;
  if (d.kind != 0 && d.kind != 1) return false;
  if (d.sample_prob > 0.f && (d.C > 64 || d.C < 1 || (d.E + d.U) / 4 > 2 * NT)) return false;   // duty E: one lane per class
  if (d.kind == 1 && (d.K < 1 || d.F < 1 || S * frames_per_slice(d) / 4 > NT)) return false;
  const int K = d.E + d.U;
  if (K % (NW * 4) || K / NW > 384) return false;      // k range of a wave: whole 16-byte pieces, <= 384 weight registers
  if (R * d.U / 4 > 2 * NT || d.U / 4 > NT || S * (d.E / S / 4) + S > 2 * NT) return false;   // gathers: <= 2 pieces per thread
  if (4 * d.U / P > 64 || (4 * d.U / P) % 4) return false;
  if ((d.E / S) % 4 || d.E / S > NT) return false;
  const int FS = frames_per_slice(d);
  if (FS > 64 || FS < 1) return false;
  if (d.U / P > 16 || d.U % 32) return false;          // duty B: 16 columns per workgroup at most
  return lds_floats(d, FS, stream_values(d, FS)) * 4 <= 160 * 1024 - 512;
}

// ---- backward
static bool bwd_ks4(const SpPersistDesc &d) { return ((d.E + d.U) / P) % 16 == 0; }
static int bwd_frames_per_thread(const SpPersistDesc &d) {
  const int FS = frames_per_slice(d), U4 = d.U / 4;
  int nfh = NT / U4;
  if (nfh > FS) nfh = FS;
  if (nfh < 1) nfh = 1;
  return (FS + nfh - 1) / nfh;
}
static size_t bwd_loc_floats(const SpPersistDesc &d, int FS) {      // (LOCB) the arrays between v_s and the scratch
  if (d.kind != 1) return 0;
  const size_t pb = (d.K - 1) / 2, TeP = (size_t)S * FS, FF = (size_t)FS * d.F, FFp = (FF + 3) & ~(size_t)3;
  return (size_t)d.F * d.U + (size_t)((d.K + 3) & ~3) * d.F + ((pb + TeP + d.K + 8) & ~(size_t)3) + 3 * FFp + TeP * d.F + 2 * S;
}
static size_t bwd_lds_floats(const SpPersistDesc &d, bool stream) {
  const int FS = frames_per_slice(d), UW = d.U / P, NC = (d.E + d.U) / P, KBW = 4 * d.U / NW;
  int nfh = NT / (d.U / 4);
  if (nfh > FS) nfh = FS;
  if (nfh < 1) nfh = 1;
  if (d.kind == 1) nfh = NW;                       // the score backward's partial dq rows: one per wave
  size_t scr = (size_t)NW * R * (KBW / 2) + (size_t)NW * NC * 4;
  const size_t d1 = (size_t)d.E + 192 + (size_t)nfh * d.U, d1b = d.U, d2 = (size_t)R * (d.U + 4) + 64 + NW * 64;
  if (scr < d1) scr = d1;
  if (scr < d1b) scr = d1b;
  if (scr < d2) scr = d2;
  return (size_t)FS * d.U + (stream ? 0 : (size_t)FS * d.E) + (size_t)UW * (d.U + 4) + d.U + bwd_loc_floats(d, FS) + scr + 64;
}
// location-aware attention: the values slice is read from L2 in every step when it does not fit next to the keys
static bool bwd_stream_values(const SpPersistDesc &d) {
  if (d.kind != 1) return false;
  if (const char *e = getenv("NABU_SPELLER_STREAM_VALUES")) return atoi(e) != 0;     // (tests: force the streamed path)
  return bwd_lds_floats(d, false) * 4 > 160 * 1024 - 512;
}
static bool bwd_shape_ok(const SpPersistDesc &d) {
  if (!shape_ok(d)) return false;
  const int NC = (d.E + d.U) / P, KBW = 4 * d.U / NW;
  if ((d.E + d.U) % P || NC % 4 || R * NC > NT) return false;
  if (bwd_ks4(d)) {
    if (NC / 16 > 3 || KBW / 4 > 128 || (KBW / 4) % 2) return false;
  } else {
    if (NC / 4 > 3 || KBW % 32 || KBW / 16 > 8) return false;
  }
  if (d.E / 4 > NT || d.U / 4 > NT || (d.U / S) % 4) return false;
  if (d.kind == 0 && bwd_frames_per_thread(d) > 8) return false;
  if (d.kind == 1) {
    const int FS = frames_per_slice(d), PF = (FS * d.F + 4) & ~3;
    // unit groups of a lane, filters in registers, conv-kernel elements per thread, ring pieces per thread
    if (d.U / 4 > 128 || d.F > 12 || d.K * d.F > 4 * NT || S * PF / 4 > 2 * NT || S * FS > NT) return false;
  }
  return bwd_lds_floats(d, bwd_stream_values(d)) * 4 <= 160 * 1024 - 512;
}
static size_t bwd_ring_bytes(const SpPersistDesc &d) {
  const size_t kb = (size_t)R * (d.E + d.U) * 4, ab = (size_t)R * S * d.U * 4, qb = (size_t)R * d.U * 4, zb = (size_t)R * 4 * d.U * 4;
  const size_t fb = d.kind == 1 ? (size_t)R * S * ((frames_per_slice(d) * d.F + 4) & ~3) * 4 : 0;
  return (size_t)NU * RING * (kb + ab + qb + zb + fb);
}
bool speller_persist_bwd_ok(const SpPersistDesc &d) {
  const char *env = getenv("NABU_SPELLER_PERSIST_BWD");
  if (env && !atoi(env)) return false;
  if (d.kind == 1) {
    // NABU_SPELLER_PERSIST_BWD_LOC: 0 = step chain, 1 (default) = persistent when the values slice is LDS-resident,
    // 2 = also when it has to be read from L2 in every step.  Measured: cfg3's geometry with 10 filters of 101 taps
    // 31.3 ms per training step against 32.3 on the chain; cfg5's geometry (25 frames per slice, values streamed)
    // 65 us per decoder step and 32 rows = 72.4 ms per training step against 67.4 on the four-stream chain.
    const char *e2 = getenv("NABU_SPELLER_PERSIST_BWD_LOC");
    const int mode = e2 ? atoi(e2) : 1;
    if (mode <= 0 || !bwd_shape_ok(d) || (mode == 1 && bwd_stream_values(d))) return false;
  }
  return bwd_shape_ok(d) && device_fits();
}
size_t speller_persist_bwd_ws_bytes(const SpPersistDesc &d) { return bwd_shape_ok(d) ? TABLE_BYTES + bwd_ring_bytes(d) : 0; }

int speller_persist_bwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const float *kxhT,
                        const float *wq, const float *v, const float *keys, const float *values, const float *acts,
                        const float *Cs, const float *q, const float *ctx, const float *align, const float *dH, float *dCtx,
                        float *dq, float *dz, float *dkeys, float *dv_part, int *status, void *ws, size_t ws_bytes,
                        hipStream_t stream, const float *conv_kernel, const float *conv_proj, float *ds_all, float *cf_all,
                        float *dck_part) {
  if (!speller_persist_bwd_ok(d)) return fail(NABU_EUNSUP, "persistent decoder (backward): unsupported shape");
  if (ws_bytes < speller_persist_bwd_ws_bytes(d)) return fail(NABU_EWS, "persistent decoder (backward): workspace too small");
  const bool loc = d.kind == 1;
  if (loc && !(conv_kernel && conv_proj && ds_all && cf_all && dck_part))
    return fail(NABU_EINVAL, "persistent decoder (backward): location-aware attention needs its kernels and output arrays");
  BArgs a;
  a.L = d.L; a.U = d.U; a.E = d.E; a.Te = d.Te; a.FS = frames_per_slice(d);
  a.Bt = d.B; a.b0 = 0; a.K = d.K; a.F = d.F; a.stream_vals = bwd_stream_values(d) ? 1 : 0;
  a.ck = conv_kernel; a.wf = conv_proj; a.ds_all = ds_all; a.cf_all = cf_all; a.dck_part = dck_part;
  a.dec_len = dec_len; a.enc_len = enc_len; a.kxhT = kxhT; a.wq = wq; a.v = v; a.keys = keys; a.values = values;
  a.acts = acts; a.Cs = Cs; a.q = q; a.ctx = ctx; a.align = align; a.dH = dH; a.dCtx = dCtx; a.dq = dq; a.dz = dz;
  a.dkeys = dkeys; a.dv_part = dv_part;
  a.keep = d.keep_prob; a.dscale = d.drop_scale;
  if (d.keep_prob < 1.f && !d.drop_scale) return fail(NABU_EINVAL, "persistent decoder (backward): dropout needs the drop_scale array");
  if (d.keep_prob < 1.f) {     // the same draws as the forward call's (whatever kernels ran it)
    const size_t groups = (size_t)d.L * d.B * d.U / 4;
    hipLaunchKernelGGL(speller_randoms_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, stream, groups, (size_t)0,
                       d.B, d.U, d.keep_prob, d.seed, d.seed_offset, d.drop_scale, 0ull, 0ull, static_cast<unsigned *>(nullptr));
    NABU_LAUNCH_CHECK();
  }
  a.table = static_cast<unsigned *>(ws);
  a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
  a.status = status;
  a.timeout_ticks = lstm_persist_timeout_ticks();
  const char *e = getenv("NABU_PERSIST_DEBUG");
  a.dbg = e ? atoi(e) : 0;
  const size_t lds = bwd_lds_floats(d, a.stream_vals != 0) * 4;
  const bool ks4 = bwd_ks4(d), drop = d.keep_prob < 1.f;
  auto kern = ks4 ? speller_persist_bwd_kernel<3, 128, 4, 8, false, false> : speller_persist_bwd_kernel<3, 8, 16, 8, false, false>;
  if (drop && !loc) kern = ks4 ? speller_persist_bwd_kernel<3, 128, 4, 8, true, false> : speller_persist_bwd_kernel<3, 8, 16, 8, true, false>;
  if (loc) {
    kern = ks4 ? speller_persist_bwd_kernel<3, 128, 4, 1, false, true> : speller_persist_bwd_kernel<3, 8, 16, 1, false, true>;
    if (drop) kern = ks4 ? speller_persist_bwd_kernel<3, 128, 4, 1, true, true> : speller_persist_bwd_kernel<3, 8, 16, 1, true, true>;
  }
  // per call: the attribute is per device, a cache keyed by the function alone would miss a second device
  NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  // 32 utterances per launch (4 per XCD): a batch of 64 runs as two launches on the stream
  for (int b0 = 0; b0 < d.B; b0 += NU * R) {
    a.b0 = b0;
    NABU_HIP(hipMemsetAsync(ws, 0xFF, TABLE_BYTES + bwd_ring_bytes(d), stream));
    hipLaunchKernelGGL(kern, dim3(NU * P), dim3(NT), lds, stream, a);
    NABU_LAUNCH_CHECK();
  }
  return 0;
}

bool speller_persist_ok(const SpPersistDesc &d) {
  const char *env = getenv("NABU_SPELLER_PERSIST");     // 0 = the step chain of speller.hip
  if (env && !atoi(env)) return false;
  return shape_ok(d) && device_fits();
}

// would the forward kernel read its values slice from L2 in every step (the slice does not fit the LDS next to the keys)?
bool speller_persist_streams_values(const SpPersistDesc &d) { return shape_ok(d) && stream_values(d, frames_per_slice(d)); }

size_t speller_persist_ws_bytes(const SpPersistDesc &d) {   // (independent of the switch: the workspace layout is)
  if (!shape_ok(d)) return 0;
  return TABLE_BYTES + ring_bytes(d);
}

int speller_persist_fwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const int32_t *ids,
                        const float *kperm, const float *bias, const float *emb, const float *wq, const float *v,
                        const float *keys, const float *values, const float *conv_kernel, const float *conv_proj, float *H,
                        float *Ho, float *Cs, float *acts, float *q, float *ctx, float *align, int *status, void *ws,
                        size_t ws_bytes, hipStream_t stream, const float *out_kernel, const float *out_bias,
                        int32_t *ids_used) {
  if (!speller_persist_ok(d)) return fail(NABU_EUNSUP, "persistent decoder: unsupported shape");
  if (ws_bytes < speller_persist_ws_bytes(d)) return fail(NABU_EWS, "persistent decoder: workspace too small");
  if (d.kind == 1 && !(conv_kernel && conv_proj)) return fail(NABU_EINVAL, "persistent decoder: location-aware attention needs its kernels");
  Args a;
  a.L = d.L; a.U = d.U; a.E = d.E; a.Te = d.Te; a.FS = frames_per_slice(d);
  a.Bt = d.B; a.K = d.K; a.F = d.F; a.stream_vals = stream_values(d, a.FS) ? 1 : 0;
  a.ck = conv_kernel; a.wf = conv_proj;
  a.dec_len = dec_len; a.enc_len = enc_len; a.ids = ids;
  a.kperm = kperm; a.bias = bias; a.emb = emb; a.wq = wq; a.v = v; a.keys = keys; a.values = values;
  a.H = H; a.Cs = Cs; a.acts = acts; a.q = q; a.ctx = ctx; a.align = align;
  a.Ho = Ho; a.keep = d.keep_prob; a.dscale = d.drop_scale;
  if (d.keep_prob < 1.f && !(Ho && d.drop_scale)) return fail(NABU_EINVAL, "persistent decoder: dropout needs the Ho and drop_scale arrays");
  a.wout = out_kernel; a.bout = out_bias; a.ids_w = ids_used; a.C = d.C;
  a.sprob = d.sample_prob; a.sdraw = d.sample_draws;
  if (d.sample_prob > 0.f && !d.sample_draws) return fail(NABU_EINVAL, "persistent decoder: scheduled sampling needs the sample_draws array");
  if (d.sample_prob > 0.f && !(out_kernel && out_bias && ids_used == ids))
    return fail(NABU_EINVAL, "persistent decoder: scheduled sampling needs the output projection and a writable ids array");
  a.table = static_cast<unsigned *>(ws);
  a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
  a.status = status;
  a.timeout_ticks = lstm_persist_timeout_ticks();
  const char *e = getenv("NABU_PERSIST_DEBUG");
  a.dbg = e ? atoi(e) : 0;
  const size_t lds = lds_floats(d, a.FS, a.stream_vals != 0) * 4;
  const int KW = (d.E + d.U) / NW;
  const bool loc = d.kind == 1;
  const bool reg = d.keep_prob < 1.f || d.sample_prob > 0.f;
  auto kern = loc ? (KW <= 64 ? speller_persist_fwd_kernel<64, true, false> : KW <= 192 ? speller_persist_fwd_kernel<192, true, false> : speller_persist_fwd_kernel<384, true, false>)
                  : (KW <= 64 ? speller_persist_fwd_kernel<64, false, false> : KW <= 192 ? speller_persist_fwd_kernel<192, false, false> : speller_persist_fwd_kernel<384, false, false>);
  if (reg)
    kern = loc ? (KW <= 64 ? speller_persist_fwd_kernel<64, true, true> : KW <= 192 ? speller_persist_fwd_kernel<192, true, true> : speller_persist_fwd_kernel<384, true, true>)
               : (KW <= 64 ? speller_persist_fwd_kernel<64, false, true> : KW <= 192 ? speller_persist_fwd_kernel<192, false, true> : speller_persist_fwd_kernel<384, false, true>);
  NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
  if (reg) {
    // every random number of the call in one small launch in front of it (the Philox rounds cost the decoder kernels
    // registers they do not have: 114 spilled in the hot loops with the generator inside)
    const size_t groups = d.keep_prob < 1.f ? (size_t)d.L * d.B * d.U / 4 : 0, draws = d.sample_prob > 0.f ? (size_t)d.L * d.B : 0;
    hipLaunchKernelGGL(speller_randoms_kernel, dim3((unsigned)((groups + draws + 255) / 256)), dim3(256), 0, stream, groups, draws,
                       d.B, d.U, d.keep_prob, d.seed, d.seed_offset, d.drop_scale, d.sample_seed, d.sample_offset, d.sample_draws);
    NABU_LAUNCH_CHECK();
  }
  // 32 utterances per launch (4 per XCD): a batch of 64 runs as two launches on the stream
  for (int b0 = 0; b0 < d.B; b0 += NU * R) {
    a.b0 = b0;
    NABU_HIP(hipMemsetAsync(ws, 0xFF, TABLE_BYTES + ring_bytes(d), stream));
    hipLaunchKernelGGL(kern, dim3(NU * P), dim3(NT), lds, stream, a);
    NABU_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace nabu
