// lstm_persist.hip — whole-sequence persistent recurrent kernels for one BLSTM
// layer (both directions in one launch), MI355X / gfx950.
//
// WHY: the recurrence is 2 x sum(T_l) strictly sequential steps per pass; one
// launch per timestep pays a kernel boundary (>=1.5 us) plus a cold re-read of
// W_h every step.  Here one launch covers the whole sequence and W_h never
// leaves the register file.
//
// DECOMPOSITION (8 "units" x 32 workgroups on the 8 XCDs x 32 CUs):
//   unit = (direction, shard of BS=8 batch rows); a unit's P = H/16 workgroups
//   each own 16 hidden units (= 64 gate columns) of that direction, i.e. a
//   [H x 64] slice of W_h = H*64*4 B (128 KiB at H=512) held in VGPRs
//   (64 registers per lane at 512 threads).  Block b belongs to unit b % NU, so
//   with the observed round-robin block->XCD placement a unit lives on one XCD.
//
// PER-STEP EXCHANGE (the only inter-workgroup communication):
//   forward : all-gather of h_t      — every workgroup publishes its [16 x 8]
//             slice (512 B, 16-byte stores) and reads the unit's whole [H x 8]
//             vector (16 KiB);
//   backward: reduce-scatter of dh   — every workgroup publishes its partial
//             product [8 x H] (16 KiB) cut into per-destination pieces and reads
//             the P pieces addressed to it, summing them in a fixed order.
//   THE DATA IS THE FLAG: exchange slots are pre-filled with the bit pattern
//   0xFFFFFFFF (a NaN no h or dh value can take); a consumer re-loads a slot
//   until no word holds the sentinel.  No flags, no fences, no atomics; every
//   32-bit word is individually valid or sentinel, so torn 16-byte stores are
//   harmless.  Slots form a ring of R=4; a slot is reset to the sentinel by its
//   owner after it was consumed, and every workgroup drains its stores
//   (s_waitcnt vmcnt(0)) before publishing, which orders the reset before any
//   later write to the same slot.
//   PLACEMENT-INDEPENDENT CORRECTNESS: consumers always use sc1 loads (bypass
//   the per-CU L1).  At kernel start the workgroups of a unit exchange their
//   XCC ids (through sc1 stores); only if ALL of them sit on one XCD do they
//   publish with plain stores (the data then stays in that XCD's L2 and never
//   touches the fabric); otherwise they publish with write-through sc1 stores.
//   A different placement therefore changes speed, never results.
//   Every spin is bounded by a wall-clock timeout; a timeout sets a status
//   word, makes every workgroup leave, and is reported to the host.
//
// MATH per workgroup and step: [8 x H] x [H x 64] on the fp32 VALU with packed
// FMAs (v_pk_fma_f32; 2*8*H*64 flop = 524 kflop at H=512 = 2048 cycles at the
// CU's fp32 peak): lanes are (hidden unit, k-slice) / (k-quad, gate) register
// tiles of 8x4 accumulators so that one 16-byte LDS broadcast read feeds 16 FMAs.
#include "lstm_persist.h"

#include <stdlib.h>

namespace nabu {

constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr int PT = 512;   // threads per workgroup (8 wave64, 2 per SIMD)
constexpr int UC = 16;    // hidden units per workgroup
constexpr int BS = 8;     // batch rows per unit
constexpr int RING = 4;   // exchange ring depth
constexpr int NCU = 256;  // MI355X
constexpr size_t TABLE_BYTES = 4096;   // XCC-id table in front of the ring

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PersistArgs {
  int B, T, D, H, max_len, nshard;
  const int32_t *len;
  const float *kernel[2];
  float *gates[2];
  float *cs[2];
  float *out;         // forward
  const float *dout;  // backward
  unsigned *table;    // [grid] XCC ids, pre-set to SENT
  char *xbuf;         // exchange ring
  int *status;
  unsigned long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
  int dbg;  // NABU_PERSIST_DEBUG: 1 no exchange wait, 2 no matrix product, 4 phase stamps,
            // 8 force write-through publishing (timing experiments only)
};

__device__ __forceinline__ float dpp_f(float v, const int ctrl_sel) {
  // quad permutes only (well defined on every wave64 target)
  int r;
  const int x = __builtin_bit_cast(int, v);
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;   // [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;   // [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, true); break;   // bcast lane 0
    case 3: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, true); break;   // bcast lane 1
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, true); break;   // bcast lane 2
    default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xF, 0xF, true); break;  // bcast lane 3
  }
  return __builtin_bit_cast(float, r);
}
#define QUAD_XOR1(v) dpp_f(v, 0)
#define QUAD_XOR2(v) dpp_f(v, 1)
#define QUAD_BCAST(v, i) dpp_f(v, 2 + (i))

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

__device__ __forceinline__ bool has_sentinel(const u32x4 v) {
  return v.x == SENT || v.y == SENT || v.z == SENT || v.w == SENT;
}

// publish 16 bytes: plain store when the whole unit shares one L2, else write-through
__device__ __forceinline__ void xstore(const u32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned off, bool coloc) {
  if (coloc) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
  else       __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}

// Bounded spin bookkeeping: returns true when the caller must give up.
struct SpinGuard {
  unsigned long long t0;
  unsigned spins;
  __device__ __forceinline__ void start() { t0 = wall_clock64(); spins = 0; }
  __device__ __forceinline__ bool expired(const PersistArgs &p) {
    if ((++spins & 31u) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    return wall_clock64() - t0 > p.timeout_ticks;
  }
};

// NABU_PERSIST_DEBUG & 4: block 0 / thread 0 records the wall clock (10 ns units) at phase
// boundaries of the middle timestep into status[320 + 32*pass + i] (pass 0 fwd, 1 bwd).
#define NABU_STAMP(pass, i)                                                       \
  do {                                                                            \
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)         \
      p.status[320 + 32 * (pass) + (i)] = (int)(wall_clock64());                  \
  } while (0)

// Kernel start: publish my XCC id, wait for the ids of my unit, decide whether the unit
// is co-located on one XCD.  Returns false on timeout.  flag[0] = failure, flag[1] = coloc.
__device__ __forceinline__ bool unit_handshake(const PersistArgs &p, int unit, int NU, int P, int *flag) {
  const int tid = threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID
  if (tid == 0) {
    flag[0] = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = 0;
    p.status[16 + blockIdx.x] = (int)xcc;   // diagnostic
    __hip_atomic_store(p.table + blockIdx.x, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (flag[0]) return false;   // an earlier kernel of this workspace timed out
  if (tid < 64) {
    SpinGuard guard;
    guard.start();
    unsigned v = xcc;
    bool failed = false;
    for (;;) {
      if (tid < P) v = __hip_atomic_load(p.table + unit + NU * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(v != SENT)) break;
      if (guard.expired(p)) { failed = true; break; }
    }
    const bool same = __all(v == xcc) && !(p.dbg & 8);
    if (tid == 0) {
      if (failed) {
        flag[0] = 1;
        __hip_atomic_store(p.status, 3 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      flag[1] = same ? 1 : 0;
    }
  }
  __syncthreads();
  return flag[0] == 0;
}

// LDS carve (floats)
template <int KPL>
struct FwdLds {
  static constexpr int H = 32 * KPL;
  static constexpr int SLICE = KPL * BS + 4;     // padded k-slice of the h vector
  static constexpr int HS = 0;                   // 32 slices
  static constexpr int PART = HS + 32 * SLICE;   // [8 waves][512]
  static constexpr int XS = PART + 8 * 512;      // [32][17] x-projection of the step
  static constexpr int SG = XS + 32 * 17;        // [32][17] activations to store
  static constexpr int SC = SG + 32 * 17;        // [8][17] cell state to store
  static constexpr int SO = SC + 8 * 17;         // [8][17] output to store
  static constexpr int FLAG = SO + 8 * 17;
  static constexpr int TOTAL = FLAG + 4;
};

// ===========================================================================
// forward
template <int KPL>
__global__ __launch_bounds__(PT) void lstm_persist_fwd_kernel(PersistArgs p) {
  using L = FwdLds<KPL>;
  constexpr int H = L::H;
  constexpr int P = H / UC;
  constexpr int NQ = (H * BS / 4 + PT - 1) / PT;   // 16-byte pieces of the h vector per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *hs = smem + L::HS, *part = smem + L::PART, *xs = smem + L::XS;
  float *sg = smem + L::SG, *sc = smem + L::SC, *so = smem + L::SO;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6;
  const int NU = 2 * p.nshard;
  const int unit = blockIdx.x % NU, slot = blockIdx.x / NU;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * BS;
  const int T = p.T;
  // matrix-phase identity: (hidden unit, k-slice); 4 adjacent lanes = 4 k-slices
  const int fu = (tid >> 2) & 15, fq = tid & 3, ks = 4 * w + fq;
  // gate-phase identity: (gate, batch row, hidden unit); 4 adjacent lanes = 4 gates
  const int gg = tid & 3, gb = (tid >> 2) & 7, gu = tid >> 5;
  const int gbg = b0 + gb;
  const int n_g = gbg < p.B ? p.len[gbg] : 0;
  // memory-phase identity: (hidden unit fastest -> 64-byte segments, gate, batch row)
  const int iu = tid & 15, ig = (tid >> 4) & 3, ib = tid >> 6;
  const int ibg = b0 + ib;
  const int n_i = ibg < p.B ? p.len[ibg] : 0;

  // this lane's slice of W_h stays in registers for the whole sequence (gate pairs packed)
  f32x2 Wr[KPL][2];
  {
    const float *Wh = p.kernel[dir] + (size_t)p.D * 4 * H + U0 + fu;
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
      const float *row = Wh + (size_t)(ks * KPL + j) * 4 * H;
      Wr[j][0] = (f32x2){row[0], row[H]};
      Wr[j][1] = (f32x2){row[2 * H], row[3 * H]};
    }
  }
  float c_state = 0.f, h_state = 0.f;
  if (!unit_handshake(p, unit, NU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  const size_t slot_bytes = (size_t)H * BS * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  const bool pub_lane = (gg == 0) && ((gb & 3) == 0);
  const unsigned pub_off = (unsigned)(((U0 + gu) * BS + gb) * 4);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  for (int s = 0; s < p.max_len; ++s) {
    NABU_STAMP(0, 0);
    // (a) x-projection of this step (GEMM output, bias included), coalesced
    const bool act_i = s < n_i;
    const int t_i = dir ? n_i - 1 - s : s;
    float xg = 0.f;
    if (act_i) xg = p.gates[dir][((size_t)ibg * T + t_i) * 4 * H + ig * H + U0 + iu];

    // (b) wait for h_{s-1} of the whole unit
    if (s > 0 && !(p.dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes);
      u32x4 v[NQ] = {};
      SpinGuard guard;
      guard.start();
      for (;;) {
        // every lane always loads a valid piece (surplus lanes re-read the last one): values
        // that are only conditionally defined across this loop were miscompiled by hipcc 7.2
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int qr = tid + i * PT, q = min(qr, H * BS / 4 - 1);
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)q * 16u, 0, 16);
          ok = ok && (qr >= H * BS / 4 || !has_sentinel(v[i]));
        }
        if (__all(ok)) break;
        if (guard.expired(p)) {
          if ((tid & 63) == 0) {
            flag[0] = 1;
            __hip_atomic_store(p.status, 1 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * PT;
        if (q < H * BS / 4) {
          const int k = (q * 4) / BS;
          *reinterpret_cast<u32x4 *>(hs + q * 4 + (k / KPL) * 4) = v[i];
        }
      }
    }
    NABU_STAMP(0, 1);
    xs[(ib * 4 + ig) * 17 + iu] = xg;
    __syncthreads();                                            // B1
    if (flag[0]) return;
    NABU_STAMP(0, 2);

    // reset my piece of the slot everybody finished reading (h_{s-2})
    if (s >= 2 && pub_lane) xstore(sent4, rs, (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off, coloc);

    // (c) recurrent product on the VALU (packed FMAs): acc[b][g] += h[b][k] * W[k][g]
    f32x2 acc[BS][2];
#pragma unroll
    for (int b = 0; b < BS; ++b) acc[b][0] = acc[b][1] = (f32x2){0.f, 0.f};
    if (s > 0 && !(p.dbg & 2)) {
      const float *hrow = hs + ks * L::SLICE;
#pragma unroll
      for (int j = 0; j < KPL; ++j) {
        const float4 h0 = *reinterpret_cast<const float4 *>(hrow + j * BS);
        const float4 h1 = *reinterpret_cast<const float4 *>(hrow + j * BS + 4);
        const float hb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int b = 0; b < BS; ++b) {
          const f32x2 hh = {hb[b], hb[b]};
          acc[b][0] = __builtin_elementwise_fma(hh, Wr[j][0], acc[b][0]);
          acc[b][1] = __builtin_elementwise_fma(hh, Wr[j][1], acc[b][1]);
        }
      }
      // sum the 4 k-slices of the quad (all lanes get the total)
#pragma unroll
      for (int b = 0; b < BS; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = acc[b][g >> 1][g & 1];
          v += QUAD_XOR1(v);
          v += QUAD_XOR2(v);
          acc[b][g >> 1][g & 1] = v;
        }
    }
    NABU_STAMP(0, 3);
    {  // lane fq hands rows 2fq, 2fq+1 of its wave's partial to the gate phase
      float *dst = part + w * 512 + (fu * BS + 2 * fq) * 4;
      f32x2 a0 = fq == 0 ? acc[0][0] : fq == 1 ? acc[2][0] : fq == 2 ? acc[4][0] : acc[6][0];
      f32x2 a1 = fq == 0 ? acc[0][1] : fq == 1 ? acc[2][1] : fq == 2 ? acc[4][1] : acc[6][1];
      f32x2 b0_ = fq == 0 ? acc[1][0] : fq == 1 ? acc[3][0] : fq == 2 ? acc[5][0] : acc[7][0];
      f32x2 b1_ = fq == 0 ? acc[1][1] : fq == 1 ? acc[3][1] : fq == 2 ? acc[5][1] : acc[7][1];
      *reinterpret_cast<float4 *>(dst) = make_float4(a0.x, a0.y, a1.x, a1.y);
      *reinterpret_cast<float4 *>(dst + 4) = make_float4(b0_.x, b0_.y, b1_.x, b1_.y);
    }
    __syncthreads();                                            // B2
    NABU_STAMP(0, 4);

    // (d) gates: thread = (gate gg, row gb, unit gu); part[w][tid] is its partial
    float z = xs[(gb * 4 + gg) * 17 + gu];
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) z += part[ww * 512 + tid];
    const float a = (gg == 1) ? fast_tanh(z) : fast_sigmoid(gg == 2 ? z + 1.0f : z);
    const float gi = QUAD_BCAST(a, 0), gj = QUAD_BCAST(a, 1), gf = QUAD_BCAST(a, 2), go = QUAD_BCAST(a, 3);
    const bool act_g = s < n_g;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act_g) { c_state = c_new; h_state = h_new; }

    NABU_STAMP(0, 5);
    // (e) publish h_s (frozen rows republish their state): 4 rows -> one 16-byte store
    {
      const float h1 = __shfl_down(h_state, 4), h2 = __shfl_down(h_state, 8), h3 = __shfl_down(h_state, 12);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my reset (and older stores) are performed
      if (pub_lane && s + 1 < p.max_len) {
        u32x4 pv;
        pv.x = __builtin_bit_cast(unsigned, h_state);
        pv.y = __builtin_bit_cast(unsigned, h1);
        pv.z = __builtin_bit_cast(unsigned, h2);
        pv.w = __builtin_bit_cast(unsigned, h3);
        xstore(pv, rs, (unsigned)((s % RING) * slot_bytes) + pub_off, coloc);
      }
    }

    NABU_STAMP(0, 6);
    // (f) off the critical path: results to HBM in 64-byte segments
    sg[(gb * 4 + gg) * 17 + gu] = a;
    if (gg == 0) {
      sc[gb * 17 + gu] = c_new;
      so[gb * 17 + gu] = act_g ? h_new : 0.f;
    }
    __syncthreads();                                            // B3
    NABU_STAMP(0, 7);
    if (ibg < p.B) {
      if (act_i) p.gates[dir][((size_t)ibg * T + t_i) * 4 * H + ig * H + U0 + iu] = sg[(ib * 4 + ig) * 17 + iu];
      if (ig == 0) {
        if (act_i) p.cs[dir][((size_t)ibg * T + t_i) * H + U0 + iu] = sc[ib * 17 + iu];
        p.out[((size_t)ibg * T + (act_i ? t_i : s)) * 2 * H + (size_t)dir * H + U0 + iu] = so[ib * 17 + iu];
      }
    }
    NABU_STAMP(0, 8);
  }
}

// ===========================================================================
// backward
template <int KPL>
struct BwdLds {
  static constexpr int QS = 16 * BS + 4;         // padded gate quarter of dz [16 cols][8 rows]
  static constexpr int DZ = 0;                   // 4 quarters
  static constexpr int RED = DZ + 4 * QS;        // [16 groups][128] partial dh sums
  static constexpr int XS = RED + 16 * 128;      // [32][17] saved activations
  static constexpr int XC = XS + 32 * 17;        // [3][8][17] c, c_prev, dout
  static constexpr int SG = XC + 3 * 8 * 17;     // [32][17] dz to store
  static constexpr int FLAG = SG + 32 * 17;
  static constexpr int TOTAL = FLAG + 4;
};

template <int KPL>
__global__ __launch_bounds__(PT) void lstm_persist_bwd_kernel(PersistArgs p) {
  using L = BwdLds<KPL>;
  constexpr int H = 32 * KPL;
  constexpr int P = H / UC;
  constexpr int KQ = H / 4;                       // k-quads of the product's output
  static_assert(KQ * 4 <= PT, "backward kernel supports H <= 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *dzs = smem + L::DZ, *red = smem + L::RED, *xs = smem + L::XS, *xc = smem + L::XC;
  float *sg = smem + L::SG;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x;
  const int NU = 2 * p.nshard;
  const int unit = blockIdx.x % NU, slot = blockIdx.x / NU;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * BS;
  const int T = p.T;
  // matrix-phase identity: (gate quarter cq, k-quad kq)
  const int cq = tid & 3, kq = tid >> 2;
  const bool mat_lane = kq < KQ;
  // gate-phase / memory-phase identities as in the forward kernel
  const int gg = tid & 3, gb = (tid >> 2) & 7, gu = tid >> 5;
  const int gbg = b0 + gb;
  const int n_g = gbg < p.B ? p.len[gbg] : 0;
  const int iu = tid & 15, ig = (tid >> 4) & 3, ib = tid >> 6;
  const int ibg = b0 + ib;
  const int n_i = ibg < p.B ? p.len[ibg] : 0;

  // Wr[c][jp] = (W_h[4kq+2jp][cq*H + U0 + c], W_h[4kq+2jp+1][...]) — k pairs packed
  f32x2 Wr[16][2];
  if (mat_lane) {
    const float *Wh = p.kernel[dir] + (size_t)p.D * 4 * H + (size_t)cq * H + U0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      Wr[c][0] = (f32x2){Wh[(size_t)(4 * kq + 0) * 4 * H + c], Wh[(size_t)(4 * kq + 1) * 4 * H + c]};
      Wr[c][1] = (f32x2){Wh[(size_t)(4 * kq + 2) * 4 * H + c], Wh[(size_t)(4 * kq + 3) * 4 * H + c]};
    }
  } else {
#pragma unroll
    for (int c = 0; c < 16; ++c) Wr[c][0] = Wr[c][1] = (f32x2){0.f, 0.f};
  }
  float dc_state = 0.f;
  if (!unit_handshake(p, unit, NU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // ring slot = [dest P][src P][16 u][8 b] floats
  const size_t piece_bytes = (size_t)UC * BS * 4;          // 512
  const size_t block_bytes = (size_t)P * piece_bytes;      // what one destination reads
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // saved forward values of step s (coalesced), fetched one step ahead
  auto fetch = [&](int s, float &av, float &xv) {
    av = 0.f; xv = 0.f;
    if (s >= 0 && s < n_i) {
      const int t = dir ? n_i - 1 - s : s;
      av = p.gates[dir][((size_t)ibg * T + t) * 4 * H + ig * H + U0 + iu];
      if (ig == 0) xv = p.cs[dir][((size_t)ibg * T + t) * H + U0 + iu];
      else if (ig == 1) xv = s > 0 ? p.cs[dir][((size_t)ibg * T + (dir ? t + 1 : t - 1)) * H + U0 + iu] : 0.f;
      else if (ig == 2) xv = p.dout[((size_t)ibg * T + t) * 2 * H + (size_t)dir * H + U0 + iu];
    }
  };
  float av, xv;
  fetch(p.max_len - 1, av, xv);

  for (int s = p.max_len - 1; s >= 0; --s) {
    NABU_STAMP(1, 0);
    const bool act_i = s < n_i;
    const int t_i = dir ? n_i - 1 - s : s;

    // (b) reduce-scatter input: the P partial products addressed to me (step s+1)
    float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int NQ = (P * 32 + PT - 1) / PT;
    u32x4 v[NQ] = {};
    const unsigned base = (unsigned)(((s + 1) % RING) * slot_bytes + (size_t)slot * block_bytes);
    const bool have_in = s + 1 < p.max_len && !(p.dbg & 1);
    if (have_in) {
      SpinGuard guard;
      guard.start();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int qr = tid + i * PT, q = min(qr, P * 32 - 1);   // see the forward kernel
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)q * 16u, 0, 16);
          // a surplus lane's piece may already have been handed back by its owner: ignore it
          ok = ok && (qr >= P * 32 || !has_sentinel(v[i]));
        }
        if (__all(ok)) break;
        if (guard.expired(p)) {
          if ((tid & 63) == 0) {
            flag[0] = 1;
            __hip_atomic_store(p.status, 2 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
    }
    NABU_STAMP(1, 1);
    // stage this step's saved values (fetched during the previous iteration) ...
    xs[(ib * 4 + ig) * 17 + iu] = av;
    if (ig < 3) xc[(ig * 8 + ib) * 17 + iu] = xv;
    // ... and start fetching the next step's while this one computes
    fetch(s - 1, av, xv);
    if (have_in) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int q = tid + i * PT;
        if (q < P * 32) {
          const f32x4 fv = __builtin_bit_cast(f32x4, v[i]);
          psum.x += fv.x;
          psum.y += fv.y;
          psum.z += fv.z;
          psum.w += fv.w;
          // I am the only reader of this block: hand the slot back
          xstore(sent4, rs, base + (unsigned)q * 16u, coloc);
        }
      }
    }
    // group tid/32 holds the sum over sources {tid/32 + 16m}; element (tid%32) = (u = ./2, 4 rows)
    *reinterpret_cast<float4 *>(red + (tid >> 5) * 128 + (tid & 31) * 4) = psum;
    __syncthreads();                                            // B1
    if (flag[0]) return;
    NABU_STAMP(1, 2);

    // (c) gate gradients: thread = (gate gg, row gb, unit gu)
    float dh = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) dh += red[(gg + 4 * m) * 128 + gu * 8 + gb];
    dh += QUAD_XOR1(dh);
    dh += QUAD_XOR2(dh);
    const float a = xs[(gb * 4 + gg) * 17 + gu];
    const float gi = QUAD_BCAST(a, 0), gj = QUAD_BCAST(a, 1), gf = QUAD_BCAST(a, 2), go = QUAD_BCAST(a, 3);
    const float c = xc[(0 * 8 + gb) * 17 + gu], cprev = xc[(1 * 8 + gb) * 17 + gu];
    const float dout = xc[(2 * 8 + gb) * 17 + gu];
    const bool act_g = s < n_g;
    const float tc = fast_tanh(c);
    const float dht = dout + dh;
    const float dct = dc_state + dht * go * (1.f - tc * tc);
    float dz = 0.f;
    if (act_g) {
      dz = gg == 0 ? dct * gj * gi * (1.f - gi)
         : gg == 1 ? dct * gi * (1.f - gj * gj)
         : gg == 2 ? dct * cprev * gf * (1.f - gf)
                   : dht * tc * go * (1.f - go);
      dc_state = dct * gf;
    }
    dzs[gg * L::QS + gu * BS + gb] = dz;
    sg[(gb * 4 + gg) * 17 + gu] = dz;
    NABU_STAMP(1, 3);
    __syncthreads();                                            // B2
    NABU_STAMP(1, 4);

    // (d) partial product for step s-1: acc[b][j] = sum_c dz[b][cq,c] * W[4kq+j][cq,c]
    if (s > 0) {
      f32x2 acc[BS][2];
#pragma unroll
      for (int b = 0; b < BS; ++b) acc[b][0] = acc[b][1] = (f32x2){0.f, 0.f};
      const float *dq = dzs + cq * L::QS;
      if (!(p.dbg & 2)) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float4 d0 = *reinterpret_cast<const float4 *>(dq + c * BS);
          const float4 d1 = *reinterpret_cast<const float4 *>(dq + c * BS + 4);
          const float db[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
          for (int b = 0; b < BS; ++b) {
            const f32x2 dd = {db[b], db[b]};
            acc[b][0] = __builtin_elementwise_fma(dd, Wr[c][0], acc[b][0]);
            acc[b][1] = __builtin_elementwise_fma(dd, Wr[c][1], acc[b][1]);
          }
        }
      }
      // quad all-reduce over the 4 gate quarters; lane cq then keeps k = 4kq + cq, all 8 rows
      float r[BS];
#pragma unroll
      for (int b = 0; b < BS; ++b) {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = acc[b][j >> 1][j & 1];
          x += QUAD_XOR1(x);
          x += QUAD_XOR2(x);
          t[j] = x;
        }
        r[b] = cq == 0 ? t[0] : cq == 1 ? t[1] : cq == 2 ? t[2] : t[3];
      }
      NABU_STAMP(1, 5);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my slot resets are performed before I publish
      NABU_STAMP(1, 6);
      if (mat_lane) {
        const int k = 4 * kq + cq;
        const int dest = k / UC, ul = k % UC;
        const unsigned off = (unsigned)((s % RING) * slot_bytes + (size_t)dest * block_bytes +
                                        (size_t)slot * piece_bytes + (size_t)ul * BS * 4);
        u32x4 p0, p1;
        p0.x = __builtin_bit_cast(unsigned, r[0]); p0.y = __builtin_bit_cast(unsigned, r[1]);
        p0.z = __builtin_bit_cast(unsigned, r[2]); p0.w = __builtin_bit_cast(unsigned, r[3]);
        p1.x = __builtin_bit_cast(unsigned, r[4]); p1.y = __builtin_bit_cast(unsigned, r[5]);
        p1.z = __builtin_bit_cast(unsigned, r[6]); p1.w = __builtin_bit_cast(unsigned, r[7]);
        xstore(p0, rs, off, coloc);
        xstore(p1, rs, off + 16, coloc);
      }
    }

    NABU_STAMP(1, 7);
    // (e) dz to HBM (in place over the activations) in 64-byte segments; padded frames get 0
    if (ibg < p.B)
      p.gates[dir][((size_t)ibg * T + (act_i ? t_i : s)) * 4 * H + ig * H + U0 + iu] = sg[(ib * 4 + ig) * 17 + iu];
    NABU_STAMP(1, 8);
  }
}

// ===========================================================================
static int cu_count() {
  static int n = -1;
  if (n < 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, 0) != hipSuccess || v <= 0) {
      (void)hipGetLastError();
      v = NCU;
    }
    n = v;
  }
  return n;
}

static int nshard_of(int B) { return (B + BS - 1) / BS; }

bool lstm_persist_supported(int B, int T, int H) {
  if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
  if (B <= 0 || T <= 0) return false;
  const int grid = 2 * nshard_of(B) * (H / UC);
  return grid <= NCU;
}

size_t lstm_persist_ws_bytes(int B, int T, int H) {
  if (!lstm_persist_supported(B, T, H)) return 0;
  const size_t NU = 2 * (size_t)nshard_of(B), P = H / UC;
  const size_t fwd = NU * RING * (size_t)H * BS * 4;
  const size_t bwd = NU * RING * P * P * UC * BS * 4;
  return TABLE_BYTES + (fwd > bwd ? fwd : bwd);
}

static constexpr size_t PERSIST_LDS = 96 * 1024;   // > half of 160 KiB: exactly one workgroup per CU

template <typename K>
static int launch(K kernel, const PersistArgs &a, int grid, hipStream_t stream) {
  static thread_local const void *configured[8] = {nullptr};
  const void *fn = reinterpret_cast<const void *>(kernel);
  bool done = false;
  for (auto c : configured) done = done || c == fn;
  if (!done) {
    NABU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PERSIST_LDS));
    for (auto &c : configured)
      if (!c) { c = fn; break; }
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(PT), PERSIST_LDS, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

static int run(bool fwd, int B, int T, int D, int H, int max_len, const int32_t *len,
               const float *const kernel[2], float *const gates[2], float *const cs[2], float *out,
               const float *dout, int *status, void *ws, size_t ws_bytes, hipStream_t stream) {
  if (!lstm_persist_supported(B, T, H)) return fail(NABU_EUNSUP, "persistent LSTM: unsupported B=%d H=%d", B, H);
  const size_t need = lstm_persist_ws_bytes(B, T, H);
  if (ws_bytes < need) return fail(NABU_EWS, "persistent LSTM: workspace %zu < %zu", ws_bytes, need);
  PersistArgs a;
  a.B = B; a.T = T; a.D = D; a.H = H; a.max_len = max_len; a.nshard = nshard_of(B);
  a.len = len;
  for (int i = 0; i < 2; ++i) { a.kernel[i] = kernel[i]; a.gates[i] = gates[i]; a.cs[i] = cs[i]; }
  a.out = out; a.dout = dout;
  a.status = status;
  a.table = static_cast<unsigned *>(ws);
  a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
  a.timeout_ticks = 20000000ull;   // 0.2 s at 100 MHz: a step takes microseconds
  { const char *e = getenv("NABU_PERSIST_DEBUG"); a.dbg = e ? atoi(e) : 0; }
  const int NU = 2 * a.nshard, P = H / UC;
  const int grid = NU * P;
  if (grid > cu_count()) return fail(NABU_EUNSUP, "persistent LSTM: %d workgroups > %d CUs", grid, cu_count());
  const size_t ring = fwd ? (size_t)NU * RING * H * BS * 4 : (size_t)NU * RING * P * P * UC * BS * 4;
  NABU_HIP(hipMemsetAsync(ws, 0xFF, TABLE_BYTES + ring, stream));
#define NABU_PERSIST_CASE(kpl)                                                              \
  case 32 * kpl:                                                                            \
    return fwd ? launch(lstm_persist_fwd_kernel<kpl>, a, grid, stream)                       \
               : launch(lstm_persist_bwd_kernel<kpl>, a, grid, stream);
  switch (H) {
    NABU_PERSIST_CASE(2)
    NABU_PERSIST_CASE(4)
    NABU_PERSIST_CASE(8)
    NABU_PERSIST_CASE(16)
  }
  return fail(NABU_EUNSUP, "persistent LSTM: unsupported H=%d", H);
}

int lstm_persist_fwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     float *out, int *status, void *ws, size_t ws_bytes, hipStream_t stream) {
  return run(true, B, T, D, H, max_len, len, kernel, gates, cs, out, nullptr, status, ws, ws_bytes, stream);
}

int lstm_persist_bwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     const float *dout, int *status, void *ws, size_t ws_bytes, hipStream_t stream) {
  return run(false, B, T, D, H, max_len, len, kernel, gates, cs, nullptr, dout, status, ws, ws_bytes, stream);
}

}  // namespace nabu
