// lstm_persist_mxh_bwd.hip — the BACKWARD kernel of the fp16-plane persistent recurrence (description: lstm_persist_mxh.hip).
// A translation unit of its own because it is compiled with -mllvm -amdgpu-mfma-vgpr-form=1 (nabu_amd/build.py, EXTRA_FLAGS):
// hipcc otherwise lets the matrix instructions of this kernel accumulate in the ACCUMULATION half of the register file and
// copies every result out before the vector ALU may touch it (56 v_accvgpr_read per step); with the flag they accumulate in
// ordinary registers (232 of them, no spills): 1.87 -> 1.81 us per sequential step.  The same flag makes the FORWARD kernel
// slower (it already needs all 256 ordinary registers: 1.31 -> 1.40 us), hence two files.
#include "lstm_persist_mxh.h"

#include <type_traits>

namespace nabu {

#ifndef MXH_TAG_MODE
#define MXH_TAG_MODE 0
#endif
#ifndef MXH_BWD_ORDER
#define MXH_BWD_ORDER 0
#endif


#define MXH_STAMP(pass, i)                                                         \
  do {                                                                             \
    if constexpr (DBG) {                                                           \
      if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)          \
        p.status[320 + 32 * (pass) + (i)] = (int)(wall_clock64());                 \
    }                                                                              \
  } while (0)

// ===========================================================================
// backward
template <int H>
struct MxhBwdLds {
  static constexpr int DROWB = 64 * 2 + 16;                  // bytes per slot row of dz planes: 64 columns fp16 + pad
  static constexpr int DZ = 0;                               // [2][16][DROWB] bytes
  static constexpr int INVD = (2 * 16 * DROWB + 15) / 16 * 4;   // floats: [2][8] inverse row scales of dz
  static constexpr int XST = INVD + 16;                      // [2][4][256] prefetched saved values
  static constexpr int RED = XST + 2 * 4 * 256;              // [8 rows][64] floats, final reductions
  static constexpr int FLAG = RED + 8 * 64;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H, bool DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mxh_bwd_kernel(PersistArgs p) {
  const int dbg = DBG ? p.dbg : 0;
  using L = MxhBwdLds<H>;
  constexpr int P = H / UC;
  constexpr int NT = P / 4;          // 16-k output tiles (= destination workgroups) per wave
  constexpr int NQ = P / 8;          // source pieces per lane
  static_assert(NT >= 2 && NQ >= 1 && NQ <= 4, "mxh backward: 128 <= H <= 512");
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
  // exchange / gate identity: source group s8, k quad kq, row 2 w + r2; after the butterfly: unit 4 kq + (s8 >> 1),
  // gate pair dup (0: i, j; 1: f, o).  The 32 lanes of a row are one half of the wave.
  const int s8 = lane & 7, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int grow = 2 * w + r2, gb = b0 + grow;
  const int gu = 4 * kq + (s8 >> 1), dup = s8 & 1;
  const int n_g = gb < p.B ? p.len[gb] : 0;
  constexpr int HT = NT / 2;                              // tiles per product half
  constexpr int QT = HT / 2 > 0 ? HT / 2 : 1;             // tiles per lane half and product half

  // A operands: W^T as two scaled fp16 planes.  Row m = output k = 16 (NT w + t) + n; reduction index c' = 32 j + 8 q + e
  // = 4 unit + gate.  Row scale: the largest magnitude over this workgroup's 64 gate columns (lanes q: shuffles).
  // inv_sel[hf][t][i]: the inverse scale of the output k this lane PUBLISHES in register i of piece t of half hf
  // (D layout: k = 16 tile + 4 q + i; lanes n < 8 publish the first QT tiles of a half, the others the rest).
  u32x4 Wp[2][NT][2];
  float inv_sel[2][QT][4];
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
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __shfl(inv_lane[hf * HT + t], 4 * q + i);
          const float hi = __shfl(inv_lane[hf * HT + (HT / 2 + t < HT ? HT / 2 + t : t)], 4 * q + i);
          inv_sel[hf][t][i] = n < 8 ? lo : hi;
        }
  }
  float dc_state = 0.f;
  // bias gradient / largest |dz| of my two gate columns, my row.  The sums run over up to T steps: in float64 (a T-term
  // fp32 chain was measurably worse than the step-wise path's tree over the stored dz: 1.1-1.3 x its error against a
  // float64 layer), added at the END of a step, behind the publish, where the wave only waits for the next exchange
  double db0 = 0.0, db1 = 0.0;
  float am0 = 0.f, am1 = 0.f;
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  clock_stamp(p, 1, 0);

  // ring slot = [dest P][src P][8 rows][4 k quads] x 16 bytes
  const size_t piece_bytes = (size_t)MXR * UC * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * MXHRINGB * slot_bytes, 0, (int)(MXHRINGB * slot_bytes), 0x00020000);
  const unsigned in_off = (unsigned)((size_t)slot * block_bytes + ((size_t)s8 * MXR + grow) * 64 + kq * 16);

  // saved forward values of step s, one step ahead: A, B = the activations of my two gates, C = c (dup 0) / c_prev
  // (dup 1), D = dout (dup 0); the pair exchanges what the other needs
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gb * T * 4 * H + (size_t)(2 * dup) * H + U0 + gu) * 4);
  const unsigned coff = (unsigned)(((size_t)gb * T * H + U0 + gu) * 4);
  const unsigned doff = (unsigned)(((size_t)gb * T * 2 * H + (size_t)dir * H + U0 + gu) * 4);
  auto fetch_part = [&](int s, int part_i) {
    const bool act = s >= 0 && s < n_g && !(dbg & 64);
    const int t = dir ? n_g - 1 - s : s;
    const int tc = dup == 0 ? t : (dir ? t + 1 : t - 1);
    const bool want_c = act && (dup == 0 || s > 0);
    float *st = xst + (s & 1) * 1024 + 64 * w;
    if (part_i == 0) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part_i == 1) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
    if (part_i == 2) prefetch_lds_b32(rc, want_c ? coff + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + 512);
    if (part_i == 3) prefetch_lds_b32(rd, (act && dup == 0) ? doff + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 768);
  };
  auto fetch = [&](int s) {
    for (int i = 0; i < 4; ++i) fetch_part(s, i);
  };
  fetch(p.max_len - 1);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // dz of step s goes to HBM at the top of step s - 1, behind that step's exchange loads; always issued
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  const bool st_ok = gb < p.B && !(dbg & 128);
  float d_0 = 0.f, d_1 = 0.f;
  unsigned d_mb = 0u;
  int d_t = 0;
  bool d_any = false;
  // ... and with it this workgroup's largest |dz| of the frame row (the bit pattern the plane scale is derived from
  // anyway), one word per row: rowmax_part[(direction, workgroup)][b T + t] — what the f16x3 pack of dZ as [BT, 8H]
  // needs as its row scale, reduced over the 2 P workgroups by pk_amax_persist_kernel (no pass over dz).  Always
  // issued like the other result stores; no buffer (null base, 0 records): dropped.
  __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc(
      p.rowmax_part, 0, p.rowmax_part ? (int)((size_t)2 * P * p.rowmax_stride * 4) : 0, 0x00020000);
  const unsigned moff = (unsigned)((((size_t)dir * P + slot) * p.rowmax_stride + (size_t)gb * T) * 4);
  const bool m_ok = st_ok && (lane & 31) == 0;
  auto dz_stores = [&]() {
    const unsigned o = (d_any && st_ok) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0), rsg, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(d_mb, rsm, (d_any && m_ok) ? moff + (unsigned)d_t * 4u : OOB, 0, 0);
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // GATE FACTORS AHEAD OF THE EXCHANGE.  dz is linear in what the exchange delivers (dh): with dht = dout + dh and
  // dct = dc + dht A, the gate gradients are dct F0 and (dct | dht) F1, the carry dct G, where A, F0, F1, G depend on the
  // saved forward values only.  Those were prefetched a step ahead, so they are claimed with a COUNTED wait right behind
  // the first poll round's loads (everything but the VM_AFTER operations issued since the prefetch: the publishes of the
  // previous step, this round's loads, the result stores) and turned into the factors while the exchange is in flight:
  // the LDS round trip, the pair exchange, tanh c and a dozen multiplies leave the chain between "dh arrived" and
  // "dz planes written".
  constexpr int VM_AFTER = 2 * QT + NQ + 3;
  float fA = 0.f, f0 = 0.f, f1 = 0.f, fG = 0.f, f_dout = 0.f;
  bool act_g = false;
  auto gate_factors = [&](int s) {
    asm volatile("" ::: "memory");
    const float *st = xst + (s & 1) * 1024 + tid;
    const float sA = st[0], sB = st[256], sC = st[512], sD = st[768];
    const float pA = mx_dpp<DPP_XOR1>(sA), pB = mx_dpp<DPP_XOR1>(sB), pC = mx_dpp<DPP_XOR1>(sC), pD = mx_dpp<DPP_XOR1>(sD);
    const float gi = dup ? pA : sA, gj = dup ? pB : sB, gf = dup ? sA : pA, go = dup ? sB : pB;
    const float c = dup ? pC : sC, cprev = dup ? sC : pC;
    f_dout = dup ? pD : sD;
    act_g = s < n_g;
    const float tc = fast_tanh(c);
    fA = go * (1.f - tc * tc);
    const float a0 = dup ? cprev * gf * (1.f - gf) : gj * gi * (1.f - gi);
    const float a1 = dup ? tc * go * (1.f - go) : gi * (1.f - gj * gj);
    f0 = act_g ? a0 : 0.f;
    f1 = act_g ? a1 : 0.f;
    fG = gf;
  };

  // The step loop, compiled twice: co-located units (the normal case: plain stores) and units spread over several XCDs
  // (write-through stores).  As a run-time flag the choice cost a taken branch around every publishing store (four per
  // step: the wave sat ~50 ns of a step in branch bubbles).  Returns false when the launch is abandoned (a poll timed out).
  auto steps = [&](auto CO) __attribute__((always_inline)) -> bool {
  constexpr bool coloc = decltype(CO)::value;
  for (int s = p.max_len - 1; s >= 0; --s) {
    MXH_STAMP(1, 0);
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
      // (a first round issued at once fails and costs the memory queue a round trip)
#if MXH_BWD_ORDER == 1
      // experiment: the idle time in front of the first round spent on the gate factors (claimed behind the previous
      // step's publishes only), the result stores still behind the round's loads
      wait_vm<2 * QT>();
      gate_factors(s);
#else
      __builtin_amdgcn_s_sleep(4);
#endif
      // (re-loading only the cells that failed, the others out of range, was measured: 2.11 against 2.02 us per step; two
      // poll rounds in flight — round k + 1 issued before round k is tested — 1.96 against 1.80: poll traffic is exchange traffic)
      bool first = true;
      for (;;) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, MXH_QVOL(kq) ? OOB : base + (unsigned)(8 * i) * (unsigned)(MXR * 64), 0, 16);
        if (first) {
          dz_stores();
#if MXH_BWD_ORDER != 1
          wait_vm<VM_AFTER>();
          gate_factors(s);
#endif
          first = false;
        }
        // every word must carry the tag: AND of the last bits (tag 1) / OR of the last bits (tag 0)
        unsigned a = v[0].x & v[0].y & v[0].z & v[0].w, o = v[0].x | v[0].y | v[0].z | v[0].w;
#pragma unroll
        for (int i = 1; i < NQ; ++i) {
          a &= v[i].x & v[i].y & v[i].z & v[i].w;
          o |= v[i].x | v[i].y | v[i].z | v[i].w;
        }
        if (__all(MXH_QVOL(kq) || (want1 ? (a & 1u) != 0 : (o & 1u) == 0))) break;
        if (poll_round_failed(p, flag, lane, fails, t_fail, 2)) break;
      }
    } else {
      dz_stores();
      wait_vm<0>();     // (no exchange loads to order the prefetched values: first step, or the no-waiting experiment)
      gate_factors(s);
    }
    MXH_STAMP(1, 1);
    mxf32x4 ps = __builtin_bit_cast(mxf32x4, v[0]);
#pragma unroll
    for (int i = 1; i < NQ; ++i) ps += __builtin_bit_cast(mxf32x4, v[i]);
    // sum over the 8 source groups, every lane of the group ends with the total (fixed order, bitwise equal)
    ps.x += mx_dpp<DPP_HALF_MIRROR>(ps.x); ps.y += mx_dpp<DPP_HALF_MIRROR>(ps.y);
    ps.z += mx_dpp<DPP_HALF_MIRROR>(ps.z); ps.w += mx_dpp<DPP_HALF_MIRROR>(ps.w);
    ps.x += mx_dpp<DPP_XOR1>(ps.x); ps.y += mx_dpp<DPP_XOR1>(ps.y);
    ps.z += mx_dpp<DPP_XOR1>(ps.z); ps.w += mx_dpp<DPP_XOR1>(ps.w);
    ps.x += mx_dpp<DPP_XOR2>(ps.x); ps.y += mx_dpp<DPP_XOR2>(ps.y);
    ps.z += mx_dpp<DPP_XOR2>(ps.z); ps.w += mx_dpp<DPP_XOR2>(ps.w);
    const float dh = sel4(s8 >> 1, ps.x, ps.y, ps.z, ps.w);

    // (b) gate gradients of (row, unit) from the factors computed above
    const float dht = f_dout + dh;
    const float dct = dc_state + dht * fA;
    const float d0 = dct * f0, d1 = (dup ? dht : dct) * f1;
    if (act_g) dc_state = dct * fG;
    char *const dzb = dzs + (s & 1) * (16 * L::DROWB);
    {
      // this row's largest |dz| over the workgroup's 64 columns = over the 32 lanes of the row: 16 by DPP, the two
      // halves by one swizzle; every lane of the row ends with the same value
      // (on the BIT PATTERNS of |dz|, which compare like the magnitudes: one integer maximum per stage where fmaxf costs
      // a canonicalising move besides; NaN patterns compare largest and keep a finite scale)
      unsigned mb = max(__builtin_bit_cast(unsigned, d0) & 0x7FFFFFFFu, __builtin_bit_cast(unsigned, d1) & 0x7FFFFFFFu);
      mb = max(mb, mx_dppu<DPP_XOR1>(mb));
      mb = max(mb, mx_dppu<DPP_XOR2>(mb));
      mb = max(mb, mx_dppu<DPP_HALF_MIRROR>(mb));
      mb = max(mb, mx_dppu<DPP_ROW_MIRROR>(mb));
      {   // the two 16-lane rows of my half-wave: v_permlane16_swap (gfx950) exchanges the odd rows of one copy with the even
          // rows of the other — a vector-ALU instruction where the ds_swizzle before it was an LDS round trip on the chain
        typedef unsigned mxu2 __attribute__((ext_vector_type(2)));
        const mxu2 sw = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
        mb = max(sw.x, sw.y);
      }
      // scale / inverse straight from the exponent field (clamped to [15, 253]: both normal; an all-zero row takes the
      // smallest exponent, 0 * scale = 0)
      const unsigned ex = min(max(mb >> 23, 15u), 253u);
      const float sc = __builtin_bit_cast(float, (268u - ex) << 23);
      unsigned ph, pl;
      mxh_split2x2(d0 * sc, d1 * sc, ph, pl);
      const unsigned o = (unsigned)grow * L::DROWB + (unsigned)(4 * gu + 2 * dup) * 2;
      *reinterpret_cast<unsigned *>(dzb + o) = ph;
      *reinterpret_cast<unsigned *>(dzb + o + 8 * L::DROWB) = pl;
      if ((lane & 31) == 0) invd[(s & 1) * 8 + grow] = __builtin_bit_cast(float, (ex - 14u) << 23);
      d_mb = mb;
    }
    {   // dz of this step: stored at the top of the next one; padded frames get 0
      const int t_g = dir ? n_g - 1 - s : s;
      d_any = true; d_0 = d0; d_1 = d1; d_t = act_g ? t_g : s;
    }
    MXH_STAMP(1, 2);
    __syncthreads();                                            // the step's only barrier
    // (the abort flag is read WITH the product's operands and tested behind them: one LDS round trip, not two)
    const int abort_now = *reinterpret_cast<volatile int *>(flag);
    MXH_STAMP(1, 3);
    if (s > 0) {
      // (c) partial dh of step s - 1: dz planes [16 slots x 64 columns] against W^T, tile t -> destination NT w + t,
      // in two halves of NT / 2 tiles; lanes n < 8 publish the first tiles of a half, the others (same sums) the
      // rest: piece (dest, me)[row n & 7][quad q], the last bit of every word = the slot's generation tag.  Next
      // step's saved values (HBM latency: as early as possible) are requested from inside the first half's matrix stream.
      u32x4 b1[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) b1[j] = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)n * L::DROWB + 64 * j + 16 * q);
      const float idz = invd[(s & 1) * 8 + (n & 7)];
      asm volatile("" :: "v"(b1[0]), "v"(b1[1]), "v"(idz));    // (the operands are loaded before the test below)
      if (abort_now) return false;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        mxf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
#ifndef MXH_EXP_NOPROD
#pragma unroll
            for (int t = 0; t < HT; ++t) acc[t] = MXH_MFMA(Wp[1 - g][hf * HT + t][j], b1[j], acc[t]);
#endif
            if (hf == 0) fetch_part(s - 1, 2 * j + g);     // one memory instruction behind every group of matrix instructions
            __builtin_amdgcn_sched_barrier(0);
            if (hf == 0 && j == 0 && g == 0) MXH_STAMP(1, 10);
          }
        }
        if (hf == 0) MXH_STAMP(1, 11); else MXH_STAMP(1, 12);
#ifndef MXH_EXP_NOFOLD      // (timing experiment: the fold of the two plane halves left out — wrong sums)
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          acc[t].x += mx_dpp<DPP_ROR8>(acc[t].x);
          acc[t].y += mx_dpp<DPP_ROR8>(acc[t].y);
          acc[t].z += mx_dpp<DPP_ROR8>(acc[t].z);
          acc[t].w += mx_dpp<DPP_ROR8>(acc[t].w);
        }
#endif
        if (hf == 0) MXH_STAMP(1, 4); else MXH_STAMP(1, 13);
        const int t0 = NT * w + hf * HT + (n < 8 ? 0 : HT / 2);
        const unsigned pbase = (unsigned)((it & 1) * slot_bytes + (size_t)t0 * block_bytes + (size_t)slot * piece_bytes +
                                          (size_t)(n & 7) * 64 + q * 16);
        const unsigned tag = (unsigned)(it >> 1) & 1u;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const mxf32x4 lo = acc[t], hi = acc[HT / 2 + t < HT ? HT / 2 + t : t];
          // descaled: 1 / (scale of output k) x 1 / (scale of the dz row), both powers of two
#ifdef MXH_EXP_NODESCALE    // (timing experiment: no descale — wrong sums)
          const mxf32x4 o = {(n < 8 ? lo.x : hi.x), (n < 8 ? lo.y : hi.y), (n < 8 ? lo.z : hi.z), (n < 8 ? lo.w : hi.w)};
#else
          const mxf32x4 o = {(n < 8 ? lo.x : hi.x) * inv_sel[hf][t][0] * idz, (n < 8 ? lo.y : hi.y) * inv_sel[hf][t][1] * idz,
                             (n < 8 ? lo.z : hi.z) * inv_sel[hf][t][2] * idz, (n < 8 ? lo.w : hi.w) * inv_sel[hf][t][3] * idz};
#endif
          const u32x4 ob = __builtin_bit_cast(u32x4, o);
#if MXH_TAG_MODE == 1      // diagnostic: unbiased tagging (a wrong last bit moves the word up or down by its own bit 1)
          auto tg = [&](unsigned b) {
            const unsigned wrong = (b ^ tag) & 1u;
            const bool up = (b & 2u) != 0 || (b & 0x7FFFFFFEu) == 0;
            return wrong ? (up ? b + 1u : b - 1u) : b;
          };
          const u32x4 ot = {tg(ob.x), tg(ob.y), tg(ob.z), tg(ob.w)};
#else
          const u32x4 ot = {(ob.x & ~1u) | tag, (ob.y & ~1u) | tag, (ob.z & ~1u) | tag, (ob.w & ~1u) | tag};
#endif
          // (HT = 1, H = 128: one tile per half, published by the lanes n < 8 only)
          xstore(ot, rs, ((HT >= 2 || n < 8) && !MXH_QVOL(q)) ? pbase + (unsigned)t * (unsigned)block_bytes : OOB, coloc);
        }
        if (hf == 0) MXH_STAMP(1, 14);
      }
      MXH_STAMP(1, 9);
    }
    else if (abort_now) return false;
    // (behind the publish: nothing waits for these)
    db0 += (double)d0; db1 += (double)d1;
    am0 = fmaxf(am0, fabsf(d0)); am1 = fmaxf(am1, fabsf(d1));
    MXH_STAMP(1, 5);
  }
  return true;
  };
  if (!(flag[1] != 0 ? steps(std::true_type{}) : steps(std::false_type{}))) return;
  dz_stores();
  clock_stamp(p, 1, 1);
  // bias gradient / column maxima of my 64 gate columns over the unit's 8 rows
  __syncthreads();
  // (the rows' sums meet in float64 as well: red as [8 rows][64] doubles fits the dz plane + staging area in front of it)
  double *redd = reinterpret_cast<double *>(smem);
  redd[grow * 64 + (2 * dup) * 16 + gu] = db0;
  redd[grow * 64 + (2 * dup + 1) * 16 + gu] = db1;
  __syncthreads();
  if (tid < 64) {
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < MXR; ++r) sum += redd[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = (float)sum;
  }
  __syncthreads();
  red[grow * 64 + (2 * dup) * 16 + gu] = am0;
  red[grow * 64 + (2 * dup + 1) * 16 + gu] = am1;
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) m = fmaxf(m, red[r * 64 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = m;
  }
}


int lstm_mxh_bwd_launch(int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
#define NABU_MXH_BWD_CASE(h)                                                                                         \
  case h:                                                                                                            \
    return a.dbg ? mxh_launch(lstm_mxh_bwd_kernel<h, true>, a, grid, MxhBwdLds<h>::TOTAL * sizeof(float), stream, dry)  \
                 : mxh_launch(lstm_mxh_bwd_kernel<h, false>, a, grid, MxhBwdLds<h>::TOTAL * sizeof(float), stream, dry);
  switch (H) {
    NABU_MXH_BWD_CASE(128)
    NABU_MXH_BWD_CASE(256)
    NABU_MXH_BWD_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mxh): unsupported H=%d", H);
}

}  // namespace nabu
