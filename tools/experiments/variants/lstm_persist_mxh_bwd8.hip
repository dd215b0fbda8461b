// lstm_persist_mxh_bwd8.hip — the backward kernel of the fp16-plane persistent recurrence with EIGHT waves per workgroup
// (two per SIMD), round 6.  Same arithmetic, same exchange ring and same protocol as lstm_mxh_bwd_kernel
// (lstm_persist_mxh_bwd.hip: three fp16 plane products of row-scaled operands, the reduce-scatter of partial dh tagged by the
// last bit of every word, gate factors ahead of the exchange, results stored a step late) — what changes is who does what:
//
//   * a wave is ONE batch row of the unit in the exchange / gate phase (lane = (source group s16, k quad kq) while the
//     partial sums are fetched and added; then lane = (unit 4 kq + s16 / 4, gate s16 % 4): one gate gradient per lane,
//     the row's largest |dz| is a maximum over the wave), and FOUR output tiles in the product phase (the 4-wave kernel:
//     two rows and eight tiles per wave);
//   * the chain a wave walks per step is therefore about half as long, and the two waves of a SIMD fill each other's
//     stalls (matrix results, LDS round trips, the barrier): the counters of the 4-wave kernel say 41 % of its wave cycles
//     issue instructions, 17 % are issue stalls, 42 % wait (profiles/r06_cfg2_pmc_sq.json), and with every poll
//     succeeding at once its step is still 1.57 us — a dependent chain of one wave per SIMD.
//
// Compiled, like the 4-wave kernel, with -mllvm -amdgpu-mfma-vgpr-form=1 (nabu_amd/build.py, EXTRA_FLAGS).
// H = 256 and 512 (H = 128 keeps the 4-wave kernel).
#include "lstm_persist_mxh.h"

#include <type_traits>

namespace nabu {

#define MXH8_STAMP(i)                                                              \
  do {                                                                             \
    if constexpr (DBG) {                                                           \
      if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)          \
        p.status[320 + 32 + (i)] = (int)(wall_clock64());                          \
    }                                                                              \
  } while (0)

template <int H>
struct Mxh8BwdLds {
  static constexpr int DROWB = 64 * 2 + 16;                  // bytes per slot row of dz planes: 64 columns fp16 + pad
  static constexpr int DZ = 0;                               // [2][16][DROWB] bytes
  static constexpr int INVD = (2 * 16 * DROWB + 15) / 16 * 4;   // floats: [2][8] inverse row scales of dz
  static constexpr int XST = INVD + 16;                      // [2][3 parts][512] prefetched saved values
  static constexpr int RED = XST + 2 * 3 * 512;              // [8 rows][64] floats, final reductions
  static constexpr int FLAG = RED + 8 * 64;
  static constexpr int TOTAL = FLAG + 4;
};

// (one fp32 -> its two fp16 planes, in the low halves of two words)
__device__ __forceinline__ void mxh_split1(float a, unsigned &h, unsigned &l) {
  h = mxh_cvt2(a, 0.f) & 0xFFFFu;
  l = mxh_cvt2(a - (float)__builtin_bit_cast(mxh16x2, h).x, 0.f) & 0xFFFFu;
}

template <int H, bool DBG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_mxh_bwd8_kernel(PersistArgs p) {
  const int dbg = DBG ? p.dbg : 0;
  using L = Mxh8BwdLds<H>;
  constexpr int P = H / UC;
  constexpr int NT = P / 8;          // 16-k output tiles (= destination workgroups) per wave
  constexpr int NQ = P / 16;         // source pieces per lane
  constexpr int QT = NT / 2;         // tiles a lane publishes per step
  constexpr int HT = NT / 2;         // tiles per product half
  constexpr int HQ = HT / 2 > 0 ? HT / 2 : 1;    // tiles a lane publishes per half (HT = 1, H = 256: the lanes n < 8 only)
  static_assert(NT >= 2 && NQ >= 1, "mxh backward, 8 waves: H = 256 or 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char *dzs = reinterpret_cast<char *>(smem) + L::DZ;
  float *invd = smem + L::INVD, *xst = smem + L::XST, *red = smem + L::RED;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * MXR;
  const int T = p.T;
  const int n = lane & 15, q = lane >> 4;                 // matrix-phase identity
  // exchange identity: source group s16, k quad kq; gate identity: unit 4 kq + (s16 >> 2), gate s16 & 3.  The wave is row w.
  const int s16 = lane & 15, kq = lane >> 4;
  const int grow = w, gb = b0 + grow;
  const int gu = 4 * kq + (s16 >> 2), gg = s16 & 3;
  const int n_g = gb < p.B ? p.len[gb] : 0;

  // A operands: W^T as two scaled fp16 planes.  Row m = output k = 16 (NT w + t) + n; reduction index c' = 32 j + 8 q + e
  // = 4 unit + gate.  Row scale: the largest magnitude over this workgroup's 64 gate columns (lanes q: shuffles).
  // inv_sel[t][i]: the inverse scale of the output k this lane PUBLISHES in register i of its piece t
  // (D layout: k = 16 tile + 4 q + i).
  u32x4 Wp[2][NT][2];
  float inv_sel[2 * HQ][4];
  {
    float inv_lane[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float *Wh = p.kernel[dir] + ((size_t)p.D + 16 * (NT * w + t) + n) * 4 * H + U0;
      float x[2][8], m = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x[j][e] = Wh[(size_t)(e & 3) * H + 8 * j + 2 * q + (e >> 2)];
          m = fmaxf(m, fabsf(x[j][e]));
        }
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      const float sc = mxh_scale_of(m);
      inv_lane[t] = mxh_inv_scale_of(m);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e) x[j][e] *= sc;
        mxh_split8(x[j], Wp[0][t][j], Wp[1][t][j]);
      }
    }
    // (two product halves of NT / 2 tiles: in half hf the lanes n < 8 publish its first HQ tiles, the others the rest)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < HQ; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __shfl(inv_lane[hf * HT + t], 4 * q + i);
          const float hi = __shfl(inv_lane[hf * HT + (HT >= 2 ? HQ + t : t)], 4 * q + i);
          inv_sel[hf * HQ + t][i] = n < 8 ? lo : hi;
        }
  }
  float dc_state = 0.f;
  double db = 0.0;          // bias gradient of my gate column, my row (float64: lstm_persist_mxh_bwd.hip)
  float am = 0.f;           // largest |dz| of my gate column, my row
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  clock_stamp(p, 1, 0);

  // ring slot = [dest P][src P][8 rows][4 k quads] x 16 bytes (the 4-wave kernel's)
  const size_t piece_bytes = (size_t)MXR * UC * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * MXHRINGB * slot_bytes, 0, (int)(MXHRINGB * slot_bytes), 0x00020000);
  const unsigned in_off = (unsigned)((size_t)slot * block_bytes + ((size_t)s16 * MXR + grow) * 64 + kq * 16);

  // saved forward values of step s, one step ahead, three LDS-DMA instructions per wave: A = the activation of my gate; B = c
  // (fetched by the gate-0 lane of a quad) / c_prev (gate-1 lane); D = dout (gate-2 lane); the quad shares them by DPP
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gb * T * 4 * H + (size_t)gg * H + U0 + gu) * 4);
  const unsigned coff = (unsigned)(((size_t)gb * T * H + U0 + gu) * 4);
  const unsigned doff = (unsigned)(((size_t)gb * T * 2 * H + (size_t)dir * H + U0 + gu) * 4);
  // (three staging words per lane and buffer: c / c_prev and dout come through different resources)
  auto fetch3 = [&](int s, int part_i) {
    const bool act = s >= 0 && s < n_g && !(dbg & 64);
    const int t = dir ? n_g - 1 - s : s;
    float *st = xst + (s & 1) * 1536 + 64 * w;
    if (part_i == 0) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part_i == 1) {
      const int tp = dir ? t + 1 : t - 1;
      const unsigned o = !act ? OOB : gg == 0 ? coff + (unsigned)t * (unsigned)(4 * H)
                                : (gg == 1 && s > 0) ? coff + (unsigned)tp * (unsigned)(4 * H) : OOB;
      prefetch_lds_b32(rc, o, smem, st + 512);
    }
    if (part_i == 2) prefetch_lds_b32(rd, (act && gg == 2) ? doff + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 1024);
  };
  for (int i = 0; i < 3; ++i) fetch3(p.max_len - 1, i);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // dz of step s goes to HBM at the top of step s - 1, behind that step's exchange loads; always issued
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  const bool st_ok = gb < p.B && !(dbg & 128);
  float d_v = 0.f;
  unsigned d_mb = 0u;
  int d_t = 0;
  bool d_any = false;
  __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc(
      p.rowmax_part, 0, p.rowmax_part ? (int)((size_t)2 * P * p.rowmax_stride * 4) : 0, 0x00020000);
  const unsigned moff = (unsigned)((((size_t)dir * P + slot) * p.rowmax_stride + (size_t)gb * T) * 4);
  const bool m_ok = st_ok && lane == 0;
  auto dz_stores = [&]() {
    const unsigned o = (d_any && st_ok) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_v), rsg, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(d_mb, rsm, (d_any && m_ok) ? moff + (unsigned)d_t * 4u : OOB, 0, 0);
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // GATE FACTORS AHEAD OF THE EXCHANGE (lstm_persist_mxh_bwd.hip): claimed by a counted wait behind the first poll round's
  // loads — everything but the operations issued since the last prefetch instruction: the publishes of the previous step
  // (QT), this round's loads (NQ), the two result stores
  constexpr int VM_AFTER = 2 * HQ + NQ + 2;
  float fA = 0.f, fF = 0.f, fG = 0.f, f_dout = 0.f;
  bool act_g = false;
  auto gate_factors = [&](int s) {
    asm volatile("" ::: "memory");
    const float *st = xst + (s & 1) * 1536 + tid;
    const float sA = st[0], sB = st[512], sD = st[1024];
    // the quad (gates i, j, f, o of one unit): every lane needs all four activations, c, c_prev and dout
    const float gi = QUAD_BCAST(sA, 0), gj = QUAD_BCAST(sA, 1), gf = QUAD_BCAST(sA, 2), go = QUAD_BCAST(sA, 3);
    const float c = QUAD_BCAST(sB, 0), cprev = QUAD_BCAST(sB, 1);
    f_dout = QUAD_BCAST(sD, 2);
    act_g = s < n_g;
    const float tc = fast_tanh(c);
    fA = go * (1.f - tc * tc);
    const float a = gg == 0 ? gj * gi * (1.f - gi) : gg == 1 ? gi * (1.f - gj * gj) : gg == 2 ? cprev * gf * (1.f - gf)
                                                                                            : tc * go * (1.f - go);
    fF = act_g ? a : 0.f;
    fG = gf;
  };

  auto steps = [&](auto CO) __attribute__((always_inline)) -> bool {
  constexpr bool coloc = decltype(CO)::value;
  for (int s = p.max_len - 1; s >= 0; --s) {
    MXH8_STAMP(0);
    // (a) reduce-scatter input: the partial products of step s + 1 addressed to my units
    u32x4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = zero4;
    const int it = p.max_len - 1 - s;                       // iteration count: slot it & 1, generation it >> 1
    const unsigned base = (unsigned)(((it - 1) & 1) * slot_bytes) + in_off;
    const bool have_in = it > 0 && !(dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      const bool want1 = (((it - 1) >> 1) & 1) != 0;        // the tag of the pieces published in iteration it - 1
      __builtin_amdgcn_s_sleep(4);                          // (a first round issued at once fails and costs a round trip)
      bool first = true;
      for (;;) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(16 * i) * (unsigned)(MXR * 64), 0, 16);
        if (first) {
          dz_stores();
          wait_vm<VM_AFTER>();
          gate_factors(s);
          first = false;
        }
        // every word must carry the tag: AND of the last bits (tag 1) / OR of the last bits (tag 0)
        unsigned a = v[0].x & v[0].y & v[0].z & v[0].w, o = v[0].x | v[0].y | v[0].z | v[0].w;
#pragma unroll
        for (int i = 1; i < NQ; ++i) {
          a &= v[i].x & v[i].y & v[i].z & v[i].w;
          o |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        if (__all(want1 ? (a & 1u) != 0 : (o & 1u) == 0)) break;
        if (poll_round_failed(p, flag, lane, fails, t_fail, 2)) break;
      }
    } else {
      dz_stores();
      wait_vm<0>();     // (no exchange loads to order the prefetched values: first step, or the no-waiting experiment)
      gate_factors(s);
    }
    MXH8_STAMP(1);
    mxf32x4 ps = __builtin_bit_cast(mxf32x4, v[0]);
#pragma unroll
    for (int i = 1; i < NQ; ++i) ps += __builtin_bit_cast(mxf32x4, v[i]);
    // sum over the 16 source groups (one DPP row), every lane of the row ends with the total (fixed order, bitwise equal)
    ps.x += mx_dpp<DPP_ROW_MIRROR>(ps.x); ps.y += mx_dpp<DPP_ROW_MIRROR>(ps.y);
    ps.z += mx_dpp<DPP_ROW_MIRROR>(ps.z); ps.w += mx_dpp<DPP_ROW_MIRROR>(ps.w);
    ps.x += mx_dpp<DPP_HALF_MIRROR>(ps.x); ps.y += mx_dpp<DPP_HALF_MIRROR>(ps.y);
    ps.z += mx_dpp<DPP_HALF_MIRROR>(ps.z); ps.w += mx_dpp<DPP_HALF_MIRROR>(ps.w);
    ps.x += mx_dpp<DPP_XOR1>(ps.x); ps.y += mx_dpp<DPP_XOR1>(ps.y);
    ps.z += mx_dpp<DPP_XOR1>(ps.z); ps.w += mx_dpp<DPP_XOR1>(ps.w);
    ps.x += mx_dpp<DPP_XOR2>(ps.x); ps.y += mx_dpp<DPP_XOR2>(ps.y);
    ps.z += mx_dpp<DPP_XOR2>(ps.z); ps.w += mx_dpp<DPP_XOR2>(ps.w);
    const float dh = sel4(s16 >> 2, ps.x, ps.y, ps.z, ps.w);

    // (b) the gate gradient of (row, unit, gate) from the factors computed above
    const float dht = f_dout + dh;
    const float dct = dc_state + dht * fA;
    const float dv = (gg == 3 ? dht : dct) * fF;
    if (act_g) dc_state = dct * fG;
    char *const dzb = dzs + (s & 1) * (16 * L::DROWB);
    {
      // this row's largest |dz| over the workgroup's 64 columns = over the wave: 16 lanes by DPP, the four rows of 16 by
      // readlane (bit patterns of |dz| compare like the magnitudes)
      unsigned mb = __builtin_bit_cast(unsigned, dv) & 0x7FFFFFFFu;
      mb = max(mb, mx_dppu<DPP_XOR1>(mb));
      mb = max(mb, mx_dppu<DPP_XOR2>(mb));
      mb = max(mb, mx_dppu<DPP_HALF_MIRROR>(mb));
      mb = max(mb, mx_dppu<DPP_ROW_MIRROR>(mb));
      const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)mb, 0), m1 = (unsigned)__builtin_amdgcn_readlane((int)mb, 16),
                     m2 = (unsigned)__builtin_amdgcn_readlane((int)mb, 32), m3 = (unsigned)__builtin_amdgcn_readlane((int)mb, 48);
      mb = max(max(m0, m1), max(m2, m3));
      // scale / inverse straight from the exponent field (clamped to [15, 253]: both normal; an all-zero row takes the
      // smallest exponent, 0 * scale = 0)
      const unsigned ex = min(max(mb >> 23, 15u), 253u);
      const float sc = __builtin_bit_cast(float, (268u - ex) << 23);
      unsigned ph, pl;
      mxh_split1(dv * sc, ph, pl);
      const unsigned o = (unsigned)grow * L::DROWB + (unsigned)(4 * gu + gg) * 2;
      *reinterpret_cast<unsigned short *>(dzb + o) = (unsigned short)ph;
      *reinterpret_cast<unsigned short *>(dzb + o + 8 * L::DROWB) = (unsigned short)pl;
      if (lane == 0) invd[(s & 1) * 8 + grow] = __builtin_bit_cast(float, (ex - 14u) << 23);
      d_mb = mb;
    }
    {   // dz of this step: stored at the top of the next one; padded frames get 0
      const int t_g = dir ? n_g - 1 - s : s;
      d_any = true; d_v = dv; d_t = act_g ? t_g : s;
    }
    MXH8_STAMP(2);
    __syncthreads();                                            // the step's only barrier
    const int abort_now = *reinterpret_cast<volatile int *>(flag);
    MXH8_STAMP(3);
    if (s > 0) {
      // (c) partial dh of step s - 1: dz planes [16 slots x 64 columns] against W^T, tile t -> destination NT w + t; lanes
      // n < 8 publish the first QT tiles, the others (same sums) the rest: piece (dest, me)[row n & 7][quad q], the last
      // bit of every word = the slot's generation tag.  Next step's saved values are requested from inside the matrix stream.
      u32x4 b1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)n * L::DROWB + 64 * j + 16 * q);
      const float idz = invd[(s & 1) * 8 + (n & 7)];
      asm volatile("" :: "v"(b1[0]), "v"(b1[1]), "v"(idz));    // (the operands are loaded before the test below)
      if (abort_now) return false;
      const unsigned tag = (unsigned)(it >> 1) & 1u;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        // (two halves: the first half's pieces are on their way while the second half multiplies — a single burst of all
        // pieces at the end of the step made the hand-off longer than the shorter chain made the step: 1.85 against 1.80 us)
        mxf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int t = 0; t < HT; ++t) acc[t] = MXH_MFMA(Wp[1 - g][hf * HT + t][j], b1[j], acc[t]);
            if (hf == 0 && 2 * j + g < 3) fetch3(s - 1, 2 * j + g);   // one memory instruction behind a group of matrix instructions
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (hf == 0) MXH8_STAMP(11);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          acc[t].x += mx_dpp<DPP_ROR8>(acc[t].x);
          acc[t].y += mx_dpp<DPP_ROR8>(acc[t].y);
          acc[t].z += mx_dpp<DPP_ROR8>(acc[t].z);
          acc[t].w += mx_dpp<DPP_ROR8>(acc[t].w);
        }
        if (hf == 0) MXH8_STAMP(4);
        const int t0 = NT * w + hf * HT + (n < 8 ? 0 : HT / 2);
        const unsigned pbase = (unsigned)((it & 1) * slot_bytes + (size_t)t0 * block_bytes + (size_t)slot * piece_bytes +
                                          (size_t)(n & 7) * 64 + q * 16);
#pragma unroll
        for (int t = 0; t < HQ; ++t) {
          const mxf32x4 lo = acc[t], hi = acc[HT >= 2 ? HQ + t : t];
          // descaled: 1 / (scale of output k) x 1 / (scale of the dz row), both powers of two
          const float *isel = inv_sel[hf * HQ + t];
          const mxf32x4 o = {(n < 8 ? lo.x : hi.x) * isel[0] * idz, (n < 8 ? lo.y : hi.y) * isel[1] * idz,
                             (n < 8 ? lo.z : hi.z) * isel[2] * idz, (n < 8 ? lo.w : hi.w) * isel[3] * idz};
          const u32x4 ob = __builtin_bit_cast(u32x4, o);
          const u32x4 ot = {(ob.x & ~1u) | tag, (ob.y & ~1u) | tag, (ob.z & ~1u) | tag, (ob.w & ~1u) | tag};
          xstore(ot, rs, (HT >= 2 || n < 8) ? pbase + (unsigned)t * (unsigned)block_bytes : OOB, coloc);
        }
      }
      MXH8_STAMP(9);
    }
    else if (abort_now) return false;
    // (behind the publish: nothing waits for these)
    db += (double)dv;
    am = fmaxf(am, fabsf(dv));
    MXH8_STAMP(5);
  }
  return true;
  };
  if (!(flag[1] != 0 ? steps(std::true_type{}) : steps(std::false_type{}))) return;
  dz_stores();
  clock_stamp(p, 1, 1);
  // bias gradient / column maxima of my 64 gate columns over the unit's 8 rows
  __syncthreads();
  double *redd = reinterpret_cast<double *>(smem);      // [8 rows][64] doubles over the dz plane / staging area
  redd[grow * 64 + gg * 16 + gu] = db;
  __syncthreads();
  if (tid < 64) {
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < MXR; ++r) sum += redd[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = (float)sum;
  }
  __syncthreads();
  red[grow * 64 + gg * 16 + gu] = am;
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) m = fmaxf(m, red[r * 64 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = m;
  }
}

template <typename K>
static int mxh8_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[8] = {};
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  int blocks = -1;
  for (const Seen &c : seen)
    if (c.fn == fn && c.dev == dev) blocks = c.blocks;
  if (blocks < 0) {
    NABU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 512, lds));
    for (Seen &c : seen)
      if (!c.fn) { c = Seen{fn, dev, blocks}; break; }
  }
  if (blocks < 1 || grid > NCU)
    return fail(NABU_EUNSUP, "persistent LSTM (mxh, 8 waves): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// NABU_PERSIST_BWD8=0: the 4-wave kernel everywhere
bool lstm_mxh_bwd8_takes(int H) {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_BWD8"); env = e ? atoi(e) : 1; }
  return env && (H == 256 || H == 512);
}
int lstm_mxh_bwd8_launch(int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
  // (the final float64 reduction overlays [8][64] doubles = 4 KiB on the front of the LDS request: it fits)
  switch (H) {
    case 256:
      return a.dbg ? mxh8_launch(lstm_mxh_bwd8_kernel<256, true>, a, grid, Mxh8BwdLds<256>::TOTAL * sizeof(float), stream, dry)
                   : mxh8_launch(lstm_mxh_bwd8_kernel<256, false>, a, grid, Mxh8BwdLds<256>::TOTAL * sizeof(float), stream, dry);
    case 512:
      return a.dbg ? mxh8_launch(lstm_mxh_bwd8_kernel<512, true>, a, grid, Mxh8BwdLds<512>::TOTAL * sizeof(float), stream, dry)
                   : mxh8_launch(lstm_mxh_bwd8_kernel<512, false>, a, grid, Mxh8BwdLds<512>::TOTAL * sizeof(float), stream, dry);
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mxh, 8 waves): unsupported H=%d", H);
}

}  // namespace nabu
