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
// issued behind the reset store have returned: vector-memory operations complete in issue order).  Placement:
// block b -> unit b % 8; the XCC ids are compared at start-up; a unit that is not on one XCD publishes
// write-through (correct, slower).  Every spin is bounded (time-out -> status word -> everybody leaves).
#include "speller_persist.h"

#include <stdlib.h>

#include "lstm_persist.h"

namespace nabu {
namespace {

constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr unsigned OOB = 0xFFFFFFF0u;
constexpr int NT = 256, NW = 4;      // threads, waves per workgroup: ONE wave per SIMD, up to 512 registers per lane
constexpr int NU = 8, P = 32;        // units (XCDs), workgroups per unit
constexpr int R = 4, S = 8;          // utterances per unit, frame slices per utterance
constexpr int RING = 4;
constexpr size_t TABLE_BYTES = 4096;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  int L, U, E, Te, FS;               // FS = frames per slice
  const int32_t *dec_len, *enc_len, *ids;
  const float *kperm, *bias, *emb, *wq, *v, *keys, *values;
  float *H, *Cs, *acts, *q, *ctx, *align;
  unsigned *table;
  char *xbuf;
  int *status;
  unsigned long long timeout_ticks;
  int dbg;
};

__device__ __forceinline__ bool has_sentinel(const u32x4 v) {
  return v.x == SENT || v.y == SENT || v.z == SENT || v.w == SENT;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float quad_bcast(float v, int i) { return __shfl(v, (threadIdx.x & ~3) | i); }

struct Spin {
  unsigned long long t0;
  unsigned n;
  __device__ __forceinline__ void start() { t0 = wall_clock64(); n = 0; }
  __device__ __forceinline__ bool expired(const Args &p) {
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

// NABU_PERSIST_DEBUG bit 2: wall-clock stamps of the phases of step L/2 in block 0 (status[16 + i], 10 ns ticks)
#define SP_STAMP(i)                                                                \
  do {                                                                             \
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && t == L / 2)                  \
      p.status[16 + (i)] = (int)wall_clock64();                                    \
  } while (0)

// KR = weight registers per lane >= (E+U)/4
template <int KR>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void speller_persist_fwd_kernel(Args p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int flag[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int unit = blockIdx.x % NU, slot = blockIdx.x / NU;
  const int U = p.U, E = p.E, Te = p.Te, FS = p.FS, L = p.L;
  const int K = E + U, KW = K / NW;          // k range of a wave
  const int CW = 4 * U / P, UW = U / P;      // my gate columns / my units (= my q columns)
  const int CB = E / S;                      // my context columns in duty D
  const int B = NU * R;

  // ---- start-up: XCC ids of the unit (lstm_persist.hip: unit_handshake)
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
  if (tid == 0) {
    flag[0] = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = 0;
    __hip_atomic_store(p.table + unit + NU * slot, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (flag[0]) return;          // an earlier launch on this workspace timed out
  if (tid < 64) {
    Spin g;
    g.start();
    unsigned vx = xcc;
    bool failed = false;
    for (;;) {
      if (tid < P) vx = __hip_atomic_load(p.table + unit + NU * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(vx != SENT)) break;
      if (g.expired(p)) { failed = true; break; }
    }
    const bool same = __all(vx == xcc) && !(p.dbg & 8);
    if (tid == 0) {
      if (failed) {
        flag[0] = 1;
        __hip_atomic_store(p.status, 3 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      flag[1] = same ? 1 : 0;
    }
  }
  __syncthreads();
  if (flag[0]) return;
  const bool coloc = flag[1] != 0;
  if ((p.dbg & 4) && tid == 0 && blockIdx.x < 32) p.status[160 + blockIdx.x] = (int)xcc * 2 + (coloc ? 1 : 0);

  // ---- exchange rings of my unit: h [R][U], q [R][U], ctx [R][E], partials [R*S][E+4]
  const unsigned hb = (unsigned)(R * U * 4), cb = (unsigned)(R * E * 4), pb = (unsigned)(R * S * (E + 4) * 4);
  const size_t unit_bytes = (size_t)RING * (2 * hb + cb + pb);
  char *ub = p.xbuf + (size_t)unit * unit_bytes;
  __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(ub, 0, (int)(RING * hb), 0x00020000);
  __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * hb, 0, (int)(RING * hb), 0x00020000);
  __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * 2 * hb, 0, (int)(RING * cb), 0x00020000);
  __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(ub + (size_t)RING * (2 * hb + cb), 0, (int)(RING * pb), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // ---- LDS: keys / values slices of my (utterance, frame slice), my columns of Wq, scratch
  float *keys_s = smem;                               // [FS][U]
  float *vals_s = keys_s + (size_t)FS * U;            // [FS][E]
  float *wq_s = vals_s + (size_t)FS * E;              // [U][UW]
  float *v_s = wq_s + (size_t)U * UW;                 // [U] attention vector
  float *es = v_s + U;                                // [FS] exp(score - local max) of my frames
  float *scr = es + NW * 64;                          // (one copy of es per wave)  scratch, sized by the host
  // scratch, re-used by the duties of a step (barriers in between):
  //   A: Xs = my wave's X[4 rows][KW]; its partial tile (256 floats) goes where its own X was (or behind all X)
  //   B: hs = h_t [R][U];  C: qs = q_t of my utterance [U], sc_s [FS] behind hs;  D: parts [S][CB], mz [S][2] over hs
  float *Xs = scr + (size_t)w * R * KR;              // rows of KR floats: columns >= KW stay zero (so do their weights)
  constexpr bool red_in_x = R * KR >= 256;
  constexpr int red_stride = red_in_x ? R * KR : 256;
  float *red0 = red_in_x ? scr : scr + NW * R * KR;   // tile of wave ww at red0 + ww * red_stride: [64 columns][4 rows]
  // (KW < KR: the staged X has zero padding that must stay zero -> the other duties' scratch lies behind it)
  float *aux = KW == KR ? scr : scr + NW * R * KR + (red_in_x ? 0 : NW * 256);
  float *hs = aux;
  float *qs = aux + R * U;
  float *sc_s = qs + U;
  float *parts = aux;
  float *mz = parts + S * CB;

  const int ci = slot / S, cs = slot % S;             // duties C, D: utterance of the unit, frame slice
  const int cbg = unit * R + ci;                      // its batch row
  const int f0 = cs * FS;
  for (int i = tid; i < FS * U; i += NT) {
    const int f = f0 + i / U;
    keys_s[i] = f < Te ? p.keys[((size_t)cbg * Te + f) * U + i % U] : 0.f;
  }
  for (int i = tid; i < FS * E; i += NT) {
    const int f = f0 + i / E;
    vals_s[i] = f < Te ? p.values[((size_t)cbg * Te + f) * E + i % E] : 0.f;
  }
  for (int i = tid; i < U * UW; i += NT) wq_s[(i % UW) * U + i / UW] = p.wq[(size_t)(i / UW) * U + UW * slot + i % UW];   // [column][k]
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
  const int gb = unit * R + grow;                       // batch row
  const int gunit = UW * slot + gu;                     // hidden unit
  const float gbias = gate_thr ? p.bias[gg * U + gunit] : 0.f;
  const int glen = gate_thr ? p.dec_len[gb] : 0;
  float c_state = 0.f, h_state = 0.f;
  // duty D state: my context column of the previous step, my frames' alignments of the previous step
  float ctx_prev = 0.f, al_prev = 0.f;
  // Saved tensors (what the backward pass reads): written one step late, right before the matrix product of the
  // next step — a poll loop waits for every older memory operation of its wave (one in-order counter), and an HBM
  // store in front of a poll put its acknowledgement (~4 us for the scattered 4-byte stores) on the critical path
  float a_last = 0.f, q_last = 0.f;
  auto save_step = [&](int ts) {     // step ts is complete in my registers
    if (gate_thr) {
      p.acts[((size_t)ts * B + gb) * 4 * U + gg * U + gunit] = a_last;
      if (gg == 0) {
        p.Cs[((size_t)(ts + 1) * B + gb) * U + gunit] = c_state;
        p.H[((size_t)(ts + 1) * B + gb) * U + gunit] = h_state;
      }
    }
    {
      const int NO = R * UW, KG = min(64, NT / NO);
      const int o = tid / KG, kg = tid % KG;
      if (o < NO && kg == 0) p.q[((size_t)ts * B + unit * R + o / UW) * U + UW * slot + o % UW] = q_last;
    }
    if (tid < CB) p.ctx[((size_t)(ts + 1) * B + cbg) * E + cs * CB + tid] = ctx_prev;
    if (tid < FS && f0 + tid < Te) p.align[((size_t)(ts + 1) * B + cbg) * Te + f0 + tid] = al_prev;
  };
  const int clen = p.dec_len[cbg], cn = min(max(p.enc_len[cbg], 0), Te);
  const float vreg_dummy = 0.f;
  (void)vreg_dummy;
  __syncthreads();

  for (int t = 0; t < L; ++t) {
    const unsigned so = (unsigned)(t % RING), sp = (unsigned)((t + RING - 1) % RING), sr = (unsigned)((t + RING - 2) % RING);
    if ((p.dbg & 4) && unit == 0 && tid == 0 && t == L / 2) p.status[128 + slot] = (int)wall_clock64();   // step start
    SP_STAMP(0);
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
      if (!(p.dbg & 16)) save_step(t - 1);
      // product (in-order LDS: my wave reads what it wrote)
      const float *xr = Xs + (lane & 3) * KR;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      f32x4 xq[2];
      xq[0] = *reinterpret_cast<const f32x4 *>(xr);
#pragma unroll
      for (int k4 = 0; k4 < KR / 4; ++k4) {      // straight line: the operands of the next four k travel while these multiply
        if (k4 + 1 < KR / 4) xq[(k4 + 1) & 1] = *reinterpret_cast<const f32x4 *>(xr + 4 * (k4 + 1));
        const f32x4 x = xq[k4 & 1];
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, Wr[4 * k4 + 0], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.y, Wr[4 * k4 + 1], a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.z, Wr[4 * k4 + 2], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x.w, Wr[4 * k4 + 3], a1, 0, 0, 0);
      }
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
      if (gate_thr) {
        z += gbias + p.emb[(size_t)p.ids[(size_t)t * B + gb] * 4 * U + gg * U + gunit];
        a = gg == 1 ? tanhf_(z) : sigmoidf_(gg == 2 ? z + 1.0f : z);
      }
      const float gi = quad_bcast(a, 0), gj = quad_bcast(a, 1), gf = quad_bcast(a, 2), go = quad_bcast(a, 3);
      const bool act = t < glen;
      if (gate_thr && act) {
        c_state = c_state * gf + gi * gj;
        h_state = tanhf_(c_state) * go;
      }
      a_last = act ? a : 0.f;        // saved tensors of this step go to HBM under the NEXT step's product
      if ((p.dbg & 4) && unit == 0 && tid == 0 && t == L / 2) p.status[64 + slot] = (int)wall_clock64();   // publish time of h_t
      const bool pub = gate_thr && gg == 0;
      xst1(fbits(h_state), rh, pub ? so * hb + (unsigned)((grow * U + gunit) * 4) : OOB, coloc);
      xst1(SENT, rh, (pub && t >= 2) ? sr * hb + (unsigned)((grow * U + gunit) * 4) : OOB, coloc);
    }
    SP_STAMP(3);
    __syncthreads();     // the partial tiles have been read: the scratch is free for h_t
    SP_STAMP(11);
    // =========================== B: query ==========================
    {
      const int NPC = R * U / 4;                       // <= 2 * NT (host check)
      const int q0 = min(tid, NPC - 1), q1 = min(NT + tid, NPC - 1);
      u32x4 v0, v1;
      Spin g;
      g.start();
      int iters = 0;
      for (;;) {
        v0 = xld4(rh, so * hb + (unsigned)(q0 * 16));
        v1 = xld4(rh, so * hb + (unsigned)(q1 * 16));
        ++iters;
        if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && t == L / 2 && iters < 30) p.status[192 + iters] = (int)wall_clock64();
        if (__all(!has_sentinel(v0) && !has_sentinel(v1))) break;
        if (g.expired(p)) { SP_TIMEOUT(1); break; }
      }
      if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && t == L / 2) { p.status[224] = iters; p.status[225] = (int)wall_clock64(); }
      if (tid < NPC) *reinterpret_cast<f32x4 *>(hs + 4 * q0) = __builtin_bit_cast(f32x4, v0);
      if (NT + tid < NPC) *reinterpret_cast<f32x4 *>(hs + 4 * q1) = __builtin_bit_cast(f32x4, v1);
    }
    SP_STAMP(4);
    __syncthreads();
    if (flag[0]) return;
    {
      // outputs (row, c): R*UW; KG lanes split k
      const int NO = R * UW, KG = min(64, NT / NO), kper = U / KG;
      const int o = tid / KG, kg = tid % KG;
      float s = 0.f;
      if (o < NO) {
        // lanes kg of an output take interleaved 16-byte pieces of k: conflict-free LDS reads, four independent chains
        const f32x4 *h4 = reinterpret_cast<const f32x4 *>(hs + (o / UW) * U), *w4 = reinterpret_cast<const f32x4 *>(wq_s + (o % UW) * U);
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int j = kg; j < U / 4; j += KG) a4 += h4[j] * w4[j];
        s = (a4.x + a4.y) + (a4.z + a4.w);
      }
      (void)kper;
      for (int m = 1; m < KG; m <<= 1) s += __shfl_xor(s, m);
      if ((p.dbg & 4) && unit == 0 && tid == 0 && t == L / 2) p.status[96 + slot] = (int)wall_clock64();   // publish time of q_t
      const bool pub = o < NO && kg == 0;
      const int row = pub ? o / UW : 0, c = pub ? o % UW : 0;
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
    __syncthreads();
    if (flag[0]) return;
    const bool frozen = t >= clen;
    for (int fg = 0; fg < FS; fg += 4 * NW) {      // four frames of a wave at a time: independent chains
      float sacc[4] = {0.f, 0.f, 0.f, 0.f};
      const f32x4 *q4 = reinterpret_cast<const f32x4 *>(qs), *v4 = reinterpret_cast<const f32x4 *>(v_s);
      for (int u4 = lane; u4 < U / 4; u4 += 64) {
        const f32x4 qq = q4[u4], vv = v4[u4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = min(fg + w + NW * i, FS - 1);
          const f32x4 kk = reinterpret_cast<const f32x4 *>(keys_s + (size_t)f * U)[u4];
          sacc[i] = fmaf(vv.x, tanhf_(kk.x + qq.x), sacc[i]);
          sacc[i] = fmaf(vv.y, tanhf_(kk.y + qq.y), sacc[i]);
          sacc[i] = fmaf(vv.z, tanhf_(kk.z + qq.z), sacc[i]);
          sacc[i] = fmaf(vv.w, tanhf_(kk.w + qq.w), sacc[i]);
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
      float m = fmaxf(sc, -3.0e38f);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
      const float e = lane < FS ? expf(sc - m) : 0.f;
      es_w[lane] = e;
      m_loc = m;
      z_loc = wsum(e);
    }
    {
      // partial context: thread = 4 columns
      for (int c4 = tid; c4 < E / 4; c4 += NT) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (!frozen) {
#pragma unroll 4
          for (int f = 0; f < FS; ++f) a += es_w[f] * *reinterpret_cast<const f32x4 *>(vals_s + (size_t)f * E + 4 * c4);
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
        fac[i] = expf(mz[2 * i] - M);
        Z += fac[i] * mz[2 * i + 1];
      }
      const float inv = 1.0f / Z;
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
        float a = frozen ? al_prev : es[tid] * expf(m_loc - M) * inv;     // (tid < FS <= 64: wave 0's copy)
        if (!frozen && f0 + tid >= cn) a = 0.f;
        al_prev = a;
      }
    }
    SP_STAMP(10);
    __syncthreads();     // the scratch is re-used by the next step's gather
  }
  save_step(L - 1);
}

int kr_for(int KW) { return KW <= 64 ? 64 : KW <= 192 ? 192 : 384; }
size_t lds_floats(const SpPersistDesc &d, int FS) {
  const size_t K = d.E + d.U, UW = d.U / P;
  const size_t KR = kr_for((int)(K / NW));
  size_t scr = NW * R * KR;
  if (R * KR < 256) scr += NW * 256;
  const size_t need_c = (size_t)R * d.U + d.U + FS + 64, need_d = (size_t)S * (d.E / S) + 2 * S + 64;
  const size_t aux = need_c > need_d ? need_c : need_d;
  if (K / NW == KR) scr = scr > aux ? scr : aux;   // aliased
  else scr += aux;
  return (size_t)FS * d.U + (size_t)FS * d.E + (size_t)d.U * UW + d.U + NW * 64 + scr + 64;
}
int frames_per_slice(const SpPersistDesc &d) { return (d.Te + S - 1) / S; }
size_t ring_bytes(const SpPersistDesc &d) {
  const size_t hb = (size_t)R * d.U * 4, cb = (size_t)R * d.E * 4, pb = (size_t)R * S * (d.E + 4) * 4;
  return (size_t)NU * RING * (2 * hb + cb + pb);
}

}  // namespace

bool speller_persist_ok(const SpPersistDesc &d) {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_SPELLER_PERSIST"); env = e ? atoi(e) : 1; }
  if (!env) return false;
  if (d.B != NU * R || d.U % 32 || d.E % 32 || d.U < 32 || d.E < 32) return false;
  const int K = d.E + d.U;
  if (K % (NW * 4) || K / NW > 384) return false;      // k range of a wave: whole 16-byte pieces, <= 384 weight registers
  if (R * d.U / 4 > 2 * NT || d.U / 4 > NT || S * (d.E / S / 4) + S > 2 * NT) return false;   // gathers: <= 2 pieces per thread
  if (4 * d.U / P > 64 || (4 * d.U / P) % 4) return false;
  if ((d.E / S) % 4 || d.E / S > NT) return false;
  const int FS = frames_per_slice(d);
  if (FS > 64 || FS < 1) return false;
  if (NT / (R * (d.U / P)) < 1) return false;
  if (d.U % (NT / (R * (d.U / P)) > 64 ? 64 : NT / (R * (d.U / P)))) return false;
  return lds_floats(d, FS) * 4 <= 160 * 1024 - 512;
}

size_t speller_persist_ws_bytes(const SpPersistDesc &d) {
  if (!speller_persist_ok(d)) return 0;
  return TABLE_BYTES + ring_bytes(d);
}

int speller_persist_fwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const int32_t *ids,
                        const float *kperm, const float *bias, const float *emb, const float *wq, const float *v,
                        const float *keys, const float *values, float *H, float *Cs, float *acts, float *q, float *ctx,
                        float *align, int *status, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (!speller_persist_ok(d)) return fail(NABU_EUNSUP, "persistent decoder: unsupported shape");
  if (ws_bytes < speller_persist_ws_bytes(d)) return fail(NABU_EWS, "persistent decoder: workspace too small");
  Args a;
  a.L = d.L; a.U = d.U; a.E = d.E; a.Te = d.Te; a.FS = frames_per_slice(d);
  a.dec_len = dec_len; a.enc_len = enc_len; a.ids = ids;
  a.kperm = kperm; a.bias = bias; a.emb = emb; a.wq = wq; a.v = v; a.keys = keys; a.values = values;
  a.H = H; a.Cs = Cs; a.acts = acts; a.q = q; a.ctx = ctx; a.align = align;
  a.table = static_cast<unsigned *>(ws);
  a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
  a.status = status;
  a.timeout_ticks = lstm_persist_timeout_ticks();
  const char *e = getenv("NABU_PERSIST_DEBUG");
  a.dbg = e ? atoi(e) : 0;
  NABU_HIP(hipMemsetAsync(ws, 0xFF, TABLE_BYTES + ring_bytes(d), stream));
  const size_t lds = lds_floats(d, a.FS) * 4;
  const int KW = (d.E + d.U) / NW;
  auto kern = KW <= 64 ? speller_persist_fwd_kernel<64> : KW <= 192 ? speller_persist_fwd_kernel<192> : speller_persist_fwd_kernel<384>;
  static thread_local const void *configured[3] = {nullptr, nullptr, nullptr};
  const void *fn = reinterpret_cast<const void *>(kern);
  bool done = false;
  for (auto c : configured) done = done || c == fn;
  if (!done) {
    NABU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    for (auto &c : configured)
      if (!c) { c = fn; break; }
  }
  hipLaunchKernelGGL(kern, dim3(NU * P), dim3(NT), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

}  // namespace nabu
