// lstm_persist_mx16.hip — the bf16-plane persistent recurrence (lstm_persist_mx.hip) with SIXTEEN batch rows per unit:
// batches of 33 .. 64 rows (cfg5: 64) in ONE launch per layer and pass instead of two launches of 32.
//
// With 16 rows the N side of v_mfma_f32_16x16x32_bf16 is filled by ONE plane of h (resp. dz): no pairing of planes,
// no fold of the two N halves, and the six plane products hh, hm, mh, mm, hl, lh cost six instructions per
// (16 columns x 32 k) for 16 rows where the 8-row kernel spends four for 8: 96 matrix instructions per wave and
// step.  Everything else — unit = one XCD's workgroups, k split over the four waves, the sentinel rings, results
// stored behind the next step's exchange loads, memory instructions riding in the matrix stream — as in
// lstm_persist_mx.hip; what grows with the rows is the exchange: a wave fetches 12 KiB of h planes per forward step
// (cells [k / 8][48 = plane * 16 + row]), a workgroup publishes and sums 32 KiB of partial dh per backward step.
#include "lstm_persist_mx.h"

namespace nabu {

constexpr int MXR16 = 16;
// backward exchange ring of the 16-row kernel: a slot is P x P KiB (1 MiB per unit at H = 512).  Three slots do not
// stay in the 4 MiB L2 next to the step's streamed tensors (WRITE_SIZE of a cfg5-shaped launch, 64 x 400: 4.2 GB against
// 0.4 GB of dz); two (-DNABU_RING_BWD16=2: lstm_persist.hip's drain + execution barrier in front of the first publish)
// spill less (2.8 GB) and are SLOWER (4.76 against 4.40 us per step), so three it is: the write-backs are asynchronous,
// and the total stays near the algorithmic bytes of SURVEY.md 8(d) (9.8 against 8.65 GB per cfg5 launch)
#ifndef NABU_RING_BWD16
#define NABU_RING_BWD16 3
#endif
constexpr int MXRINGB16 = NABU_RING_BWD16;

// ===========================================================================
// forward
template <int H>
struct Mx16FwdLds {
  static constexpr int ROWF = 17 * 4;                          // floats per (wave, row): 16 units x 4 gates + pad
  static constexpr int PART = 0;                               // [2][4 waves][16 rows][ROWF]
  static constexpr int XST = PART + 2 * 4 * MXR16 * ROWF;      // [2][4 gates][256] prefetched x-projection
  static constexpr int FLAG = XST + 2 * 4 * 256;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx16_fwd_kernel(PersistArgs p) {
  using L = Mx16FwdLds<H>;
  constexpr int P = H / UC;
  constexpr int KW = H / 4;          // k values multiplied by one wave
  constexpr int NKS = KW / 32;       // k-steps of 32 per wave
  static_assert(NKS >= 1, "mx16 forward: H >= 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *part = smem + L::PART, *xst = smem + L::XST;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * MXR16;
  const int T = p.T;
  const int n = lane & 15, q = lane >> 4;          // matrix phase: n = batch row (N index), q = k group / column group
  const int u16 = lane & 15, r4 = lane >> 4;       // finishing phase: (row 4 w + r4, unit u16), one per lane
  const int frow = 4 * w + r4, fb = b0 + frow;
  const int n_f = fb < p.B ? p.len[fb] : 0;

  // W_h slice as three bf16 planes (A operands): column (gate c, unit U0 + n), k = w KW + 32 j + 8 q + e
  u32x4 Wp[3][4][NKS];
  {
    const float *Wh = p.kernel[dir] + ((size_t)p.D + (size_t)w * KW + 8 * q) * 4 * H + U0 + n;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[((size_t)32 * j + e) * 4 * H + (size_t)c * H];
        mx_split8(x, Wp[0][c][j], Wp[1][c][j], Wp[2][c][j]);
      }
  }
  float c_state = 0.f, h_state = 0.f;
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // exchange slot of a unit: cells of 16 bytes = 8 consecutive k of one (plane, row): [k / 8][48 = plane * 16 + row]
  constexpr int NSL = 3 * MXR16;
  const size_t slot_bytes = (size_t)NSL * H * 2;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  constexpr int KGW = KW / 8;
  constexpr unsigned KSTEP_BYTES = 4 * NSL * 16;
  const unsigned off0 = (unsigned)((((size_t)w * KGW + q) * NSL + n) * 16);      // plane p: + p * 256
  const int ppl = u16 & 7;
  const bool pub_lane = ppl < 3;
  const unsigned pub_off = (unsigned)((((size_t)(U0 >> 3) + (u16 >> 3)) * NSL + ppl * MXR16 + frow) * 16);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // x-projection of step s (bias included): the four gates of (row frow, unit u16), one step ahead by LDS-DMA
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const unsigned goff = (unsigned)(((size_t)fb * T * 4 * H + U0 + u16) * 4);
  auto fetch_x_part = [&](int s, int g) {
    const int t = dir ? n_f - 1 - s : s;
    const bool act = s < n_f && !(p.dbg & 64);
    prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) + (unsigned)g * (unsigned)(4 * H) : OOB, smem,
                     xst + (s & 1) * 1024 + g * 256 + 64 * w);
  };
  auto fetch_x = [&](int s) {
    for (int g = 0; g < 4; ++g) fetch_x_part(s, g);
  };
  fetch_x(0);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // results of step s go to HBM at the top of step s + 1, behind that step's exchange loads (lstm_persist_mx.hip)
  float d_g[4] = {0.f, 0.f, 0.f, 0.f}, d_c = 0.f, d_h = 0.f;
  int d_t = 0, d_to = 0;
  bool d_act = false, d_any = false;
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.cs[dir], 0, (int)((size_t)p.B * T * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.B * T * 2 * H * 4), 0x00020000);
  const unsigned coff = (unsigned)(((size_t)fb * T * H + U0 + u16) * 4);
  const unsigned ooff = (unsigned)(((size_t)fb * T * 2 * H + (size_t)dir * H + U0 + u16) * 4);
  const bool st_ok = fb < p.B && !(p.dbg & 128);
  auto result_stores = [&]() {
    const bool on = d_any && st_ok;
    const unsigned go_ = (on && d_act) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_g[g]), rsg,
                                            go_ == OOB ? OOB : go_ + (unsigned)g * (unsigned)(4 * H), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_c), rsc,
                                          (on && d_act) ? coff + (unsigned)d_t * (unsigned)(4 * H) : OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_h), rso,
                                          on ? ooff + (unsigned)d_to * (unsigned)(8 * H) : OOB, 0, 0);
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int s = 0; s < p.max_len; ++s) {
    NABU_STAMP(0, 0);
    mxf32x4 acc[4];
    unsigned long long t_fail = 0;
    int fails = 0;
    // (a) h_{s-1} as planes: the poll loop IS the operand fetch (12 KiB of full lines per wave)
    u32x4 bp[3][NKS];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int j = 0; j < NKS; ++j) bp[pl][j] = zero4;
    if (s > 0 && !(p.dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes);
      bool first = true;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            bp[pl][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off0 + pl * 256u + j * KSTEP_BYTES, 0, 16);
        if (first) { result_stores(); first = false; }
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) mx = mx_max4(mx, bp[pl][j]);
        if (__all(mx != SENT)) break;
        if (fails == 0) t_fail = wall_clock64();
        if ((++fails & 7) == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              wall_clock64() - t_fail > p.timeout_ticks) {
            if (lane == 0) {
              flag[0] = 1;
              __hip_atomic_store(p.status, 1 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
          }
        }
      }
    } else {
      result_stores();
      wait_vm<0>();
    }
    NABU_STAMP(0, 1);
    // (b) product: 4 column tiles (gate c) x NKS k-steps x {Wl.h, Wm.m, Wh.l, Wm.h, Wh.m, Wh.h}; next step's
    // x-projection is requested from inside the matrix stream
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !(p.dbg & 2)) {
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[2][c][j], bp[0][j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[1][c][j], bp[1][j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[0][c][j], bp[2][j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[1][c][j], bp[0][j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 3); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[0][c][j], bp[1][j], acc[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[0][c][j], bp[0][j], acc[c]);
      }
    } else {
      fetch_x(s + 1);
    }
    NABU_STAMP(0, 2);
    // partial sums -> LDS [wave][row n][unit 4 q + i][4 gates]
    float *const pbuf = part + (s & 1) * (4 * MXR16 * L::ROWF);
    {
      float *d = pbuf + ((size_t)(w * MXR16 + n)) * L::ROWF + (4 * q) * 4;
      *reinterpret_cast<mxf32x4 *>(d) = (mxf32x4){acc[0].x, acc[1].x, acc[2].x, acc[3].x};
      *reinterpret_cast<mxf32x4 *>(d + 4) = (mxf32x4){acc[0].y, acc[1].y, acc[2].y, acc[3].y};
      *reinterpret_cast<mxf32x4 *>(d + 8) = (mxf32x4){acc[0].z, acc[1].z, acc[2].z, acc[3].z};
      *reinterpret_cast<mxf32x4 *>(d + 12) = (mxf32x4){acc[0].w, acc[1].w, acc[2].w, acc[3].w};
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    NABU_STAMP(0, 3);

    // (c) gates of (row frow, unit u16)
    mxf32x4 z;
    {
      const float *xs = xst + (s & 1) * 1024 + tid;
      z = (mxf32x4){xs[0], xs[256], xs[512], xs[768]};
      const float *pr = pbuf + (size_t)frow * L::ROWF + u16 * 4;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) z += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR16 * L::ROWF);
    }
    const float gi = fast_sigmoid(z.x), gj = fast_tanh(z.y), gf = fast_sigmoid(z.z + 1.0f), go = fast_sigmoid(z.w);
    const bool act = s < n_f;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act) { c_state = c_new; h_state = h_new; }

    // (d) publish h_s as planes: lane 8 g + pl collects the four pair words of plane pl -> one 16-byte store
    {
      unsigned pl[3], pr[3];
      mx_split3(h_state, pl[0], pl[1], pl[2]);
#pragma unroll
      for (int i = 0; i < 3; ++i) pr[i] = pl[i] | (mx_dppu<DPP_XOR1>(pl[i]) << 16);
      const u32x4 v0 = {pr[0], mx_dppu<0x102>(pr[0]), mx_dppu<0x104>(pr[0]), mx_dppu<0x106>(pr[0])};
      const u32x4 v1 = {mx_dppu<0x111>(pr[1]), mx_dppu<0x101>(pr[1]), mx_dppu<0x103>(pr[1]), mx_dppu<0x105>(pr[1])};
      const u32x4 v2 = {mx_dppu<0x112>(pr[2]), pr[2], mx_dppu<0x102>(pr[2]), mx_dppu<0x104>(pr[2])};
      const u32x4 pv = ppl == 0 ? v0 : ppl == 1 ? v1 : v2;
      xstore(pv, rs, (pub_lane && s + 1 < p.max_len) ? (unsigned)((s % RING) * slot_bytes) + pub_off : OOB, coloc);
      xstore(sent4, rs, (pub_lane && s >= 2) ? (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off : OOB, coloc);
    }
    NABU_STAMP(0, 4);
    {
      const int t_g = dir ? n_f - 1 - s : s;
      d_any = true; d_act = act; d_t = t_g; d_to = act ? t_g : s;
      d_g[0] = gi; d_g[1] = gj; d_g[2] = gf; d_g[3] = go;
      d_c = c_new;
      d_h = act ? h_new : 0.f;
    }
    NABU_STAMP(0, 5);
  }
  result_stores();
}

// ===========================================================================
// backward
template <int H>
struct Mx16BwdLds {
  static constexpr int DROWB = 64 * 2 + 16;                    // bytes per slot row of dz planes: 64 columns bf16 + pad
  static constexpr int DZ = 0;                                 // [2][48][DROWB] bytes
  static constexpr int XST = (2 * 48 * DROWB + 15) / 16 * 4;   // floats: [2 parities][2 passes][4][256] saved values
  static constexpr int RED = XST + 2 * 2 * 4 * 256;            // [16 rows][64] floats, final reductions
  static constexpr int FLAG = RED + 16 * 64;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx16_bwd_kernel(PersistArgs p) {
  using L = Mx16BwdLds<H>;
  constexpr int P = H / UC;
  constexpr int NT = P / 4;          // 16-k output tiles (= destination workgroups) per wave
  constexpr int NQ = P / 8;          // source pieces per lane and pass
  static_assert(NT >= 2 && NQ >= 1 && NQ <= 4, "mx16 backward: 128 <= H <= 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char *dzs = reinterpret_cast<char *>(smem) + L::DZ;
  float *xst = smem + L::XST, *red = smem + L::RED;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * MXR16;
  const int T = p.T;
  const int n = lane & 15, q = lane >> 4;                 // matrix-phase identity: n = batch row
  // exchange / gate identity, two passes (rows 2 w + r2 and 8 + 2 w + r2): source group s8, k quad kq; after the
  // butterfly: unit 4 kq + (s8 >> 1), gate pair dup
  const int s8 = lane & 7, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int gu = 4 * kq + (s8 >> 1), dup = s8 & 1;
  int grow[2], gb[2], n_g[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    grow[ps] = 8 * ps + 2 * w + r2;
    gb[ps] = b0 + grow[ps];
    n_g[ps] = gb[ps] < p.B ? p.len[gb[ps]] : 0;
  }

  // A operands: W^T planes.  Row m = output k = 16 (NT w + t) + n; reduction index c' = 32 j + 8 q + e = 4 unit + gate
  u32x4 Wp[3][NT][2];
  {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float *Wh = p.kernel[dir] + ((size_t)p.D + 16 * (NT * w + t) + n) * 4 * H + U0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[(size_t)(e & 3) * H + 8 * j + 2 * q + (e >> 2)];
        mx_split8(x, Wp[0][t][j], Wp[1][t][j], Wp[2][t][j]);
      }
    }
  }
  float dc_state[2] = {0.f, 0.f};
  float db0[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, am0[2] = {0.f, 0.f}, am1[2] = {0.f, 0.f};
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // ring slot = [dest P][src P][16 rows][4 k quads] x 16 bytes
  const size_t piece_bytes = (size_t)MXR16 * UC * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * MXRINGB16 * slot_bytes, 0, (int)(MXRINGB16 * slot_bytes), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};
  unsigned in_off[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
    in_off[ps] = (unsigned)((size_t)slot * block_bytes + ((size_t)s8 * MXR16 + grow[ps]) * 64 + kq * 16);
  constexpr unsigned SRC8 = 8 * MXR16 * 64;     // 8 sources further

  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  unsigned goff[2], coff[2], doff[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    goff[ps] = (unsigned)(((size_t)gb[ps] * T * 4 * H + (size_t)(2 * dup) * H + U0 + gu) * 4);
    coff[ps] = (unsigned)(((size_t)gb[ps] * T * H + U0 + gu) * 4);
    doff[ps] = (unsigned)(((size_t)gb[ps] * T * 2 * H + (size_t)dir * H + U0 + gu) * 4);
  }
  // saved forward values of step s (pass ps), one step ahead: A, B = activations of my two gates, C = c / c_prev,
  // D = dout (dup 0)
  auto fetch_part = [&](int s, int idx) {
    const int ps = idx >> 2, part = idx & 3;
    const bool act = s >= 0 && s < n_g[ps];
    const int t = dir ? n_g[ps] - 1 - s : s;
    const int tc = dup == 0 ? t : (dir ? t + 1 : t - 1);
    const bool want_c = act && (dup == 0 || s > 0);
    float *st = xst + ((s & 1) * 2 + ps) * 1024 + 64 * w;
    if (part == 0) prefetch_lds_b32(rg, act ? goff[ps] + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part == 1) prefetch_lds_b32(rg, act ? goff[ps] + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
    if (part == 2) prefetch_lds_b32(rc, want_c ? coff[ps] + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + 512);
    if (part == 3) prefetch_lds_b32(rd, (act && dup == 0) ? doff[ps] + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 768);
  };
  for (int i = 0; i < 8; ++i) fetch_part(p.max_len - 1, i);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  float d_0[2] = {0.f, 0.f}, d_1[2] = {0.f, 0.f};
  int d_t[2] = {0, 0};
  bool d_any = false;
  auto dz_stores = [&]() {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const bool ok = d_any && gb[ps] < p.B && !(p.dbg & 128);
      const unsigned o = ok ? goff[ps] + (unsigned)d_t[ps] * (unsigned)(16 * H) : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0[ps]), rsg, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1[ps]), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
    }
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int s = p.max_len - 1; s >= 0; --s) {
    NABU_STAMP(1, 0);
    // (a) reduce-scatter input: the partial products of step s + 1 addressed to my units, both passes
    u32x4 v[2][NQ];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int i = 0; i < NQ; ++i) v[ps][i] = zero4;
    const unsigned sbase = (unsigned)(((s + 1) % MXRINGB16) * slot_bytes);
    const bool have_in = s + 1 < p.max_len && !(p.dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      bool first = true;
      __builtin_amdgcn_s_sleep(4);      // (lstm_persist_mx.hip: a first round issued at once fails)
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int i = 0; i < NQ; ++i)
            v[ps][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, sbase + in_off[ps] + (unsigned)i * SRC8, 0, 16);
        if (first) { dz_stores(); first = false; }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int i = 0; i < NQ; ++i) mx = mx_max4(mx, v[ps][i]);
        if (__all(mx != SENT)) break;
        if (fails == 0) t_fail = wall_clock64();
        if ((++fails & 7) == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              wall_clock64() - t_fail > p.timeout_ticks) {
            if (lane == 0) {
              flag[0] = 1;
              __hip_atomic_store(p.status, 2 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
          }
        }
      }
    } else {
      dz_stores();
      wait_vm<0>();
    }
    NABU_STAMP(1, 1);
    char *const dzb = dzs + (s & 1) * (48 * L::DROWB);
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      mxf32x4 sum = __builtin_bit_cast(mxf32x4, v[ps][0]);
#pragma unroll
      for (int i = 1; i < NQ; ++i) sum += __builtin_bit_cast(mxf32x4, v[ps][i]);
      sum.x += mx_dpp<DPP_HALF_MIRROR>(sum.x); sum.y += mx_dpp<DPP_HALF_MIRROR>(sum.y);
      sum.z += mx_dpp<DPP_HALF_MIRROR>(sum.z); sum.w += mx_dpp<DPP_HALF_MIRROR>(sum.w);
      sum.x += mx_dpp<DPP_XOR1>(sum.x); sum.y += mx_dpp<DPP_XOR1>(sum.y);
      sum.z += mx_dpp<DPP_XOR1>(sum.z); sum.w += mx_dpp<DPP_XOR1>(sum.w);
      sum.x += mx_dpp<DPP_XOR2>(sum.x); sum.y += mx_dpp<DPP_XOR2>(sum.y);
      sum.z += mx_dpp<DPP_XOR2>(sum.z); sum.w += mx_dpp<DPP_XOR2>(sum.w);
      const float dh = sel4(s8 >> 1, sum.x, sum.y, sum.z, sum.w);

      // (b) gate gradients of (row, unit): the pair shares its saved values
      const float *st = xst + ((s & 1) * 2 + ps) * 1024 + tid;
      const float sA = st[0], sB = st[256], sC = st[512], sD = st[768];
      const float pA = mx_dpp<DPP_XOR1>(sA), pB = mx_dpp<DPP_XOR1>(sB), pC = mx_dpp<DPP_XOR1>(sC), pD = mx_dpp<DPP_XOR1>(sD);
      const float gi = dup ? pA : sA, gj = dup ? pB : sB, gf = dup ? sA : pA, go = dup ? sB : pB;
      const float c = dup ? pC : sC, cprev = dup ? sC : pC, dout = dup ? pD : sD;
      const bool act_g = s < n_g[ps];
      const float tc = fast_tanh(c);
      const float dht = dout + dh;
      const float dct = dc_state[ps] + dht * go * (1.f - tc * tc);
      float d0 = 0.f, d1 = 0.f;
      if (act_g) {
        d0 = dup ? dct * cprev * gf * (1.f - gf) : dct * gj * gi * (1.f - gi);
        d1 = dup ? dht * tc * go * (1.f - go) : dct * gi * (1.f - gj * gj);
        dc_state[ps] = dct * gf;
      }
      db0[ps] += d0; db1[ps] += d1;
      am0[ps] = fmaxf(am0[ps], fabsf(d0)); am1[ps] = fmaxf(am1[ps], fabsf(d1));
      unsigned ph, pm, pl;
      mx_split3x2(d0, d1, ph, pm, pl);
      const unsigned o = (unsigned)grow[ps] * L::DROWB + (unsigned)(4 * gu + 2 * dup) * 2;
      *reinterpret_cast<unsigned *>(dzb + o) = ph;
      *reinterpret_cast<unsigned *>(dzb + o + 16 * L::DROWB) = pm;
      *reinterpret_cast<unsigned *>(dzb + o + 32 * L::DROWB) = pl;
      const int t_g = dir ? n_g[ps] - 1 - s : s;
      d_0[ps] = d0; d_1[ps] = d1; d_t[ps] = act_g ? t_g : s;
    }
    d_any = true;
    NABU_STAMP(1, 2);
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    NABU_STAMP(1, 3);

    if (s > 0) {
      // (c) partial dh of step s - 1: dz planes [16 rows x 64 columns] against W^T, tile t -> destination NT w + t, in
      // two halves of NT / 2 tiles; the other memory instructions of the step ride in the matrix stream: next step's
      // saved values first, then the slot hand-back, as far in front of the last publish as possible
      u32x4 bp[3][2];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bp[pl][j] = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)(16 * pl + n) * L::DROWB + 64 * j + 16 * q);
      constexpr int HT = NT / 2;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        mxf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int g = 0; g < 6; ++g) {
            // {Wl.h, Wm.m, Wh.l, Wm.h, Wh.m, Wh.h}
            constexpr int WPL[6] = {2, 1, 0, 1, 0, 0}, BPL[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < HT; ++t) acc[t] = MX_MFMA(Wp[WPL[g]][hf * HT + t][j], bp[BPL[g]][j], acc[t]);
            const int slot_i = hf * 12 + 6 * j + g;     // one memory instruction behind every group
            if (MXRINGB16 >= 3) {
              if (slot_i < 8) fetch_part(s - 1, slot_i);
              if (slot_i >= 8 && slot_i - 8 < 2 * NQ) {
                const int ri = slot_i - 8;
                xstore(sent4, rs, have_in ? sbase + in_off[ri / NQ] + (unsigned)(ri % NQ) * SRC8 : OOB, coloc);
              }
            } else {
              // ring of 2: every hand-back in front of the first publish (drained below), then the prefetches
              if (slot_i < 2 * NQ)
                xstore(sent4, rs, have_in ? sbase + in_off[slot_i / NQ] + (unsigned)(slot_i % NQ) * SRC8 : OOB, coloc);
              if (slot_i >= 2 * NQ && slot_i - 2 * NQ < 8) fetch_part(s - 1, slot_i - 2 * NQ);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (hf == 0) {
          NABU_STAMP(1, 4);
          if (MXRINGB16 == 2) {
            // the pieces I publish into were handed back by their readers in the step I have just polled: every wave
            // drains its hand-back stores (all but the prefetches issued behind them in this half) and the workgroup
            // meets at an execution barrier before anybody publishes (lstm_persist.hip, backward (c))
            constexpr int behind = 12 - 2 * NQ < 8 ? 12 - 2 * NQ : 8;
            wait_vm<behind>();
            __builtin_amdgcn_s_barrier();
          }
        }
        // piece (dest, me)[row n][quad q]: every lane stores its tile rows
        const unsigned pbase = (unsigned)((s % MXRINGB16) * slot_bytes + (size_t)(NT * w + hf * HT) * block_bytes +
                                          (size_t)slot * piece_bytes + (size_t)n * 64 + q * 16);
#pragma unroll
        for (int t = 0; t < HT; ++t)
          xstore(__builtin_bit_cast(u32x4, acc[t]), rs, pbase + (unsigned)t * (unsigned)block_bytes, coloc);
      }
      NABU_STAMP(1, 9);
    } else {
#pragma unroll
      for (int ri = 0; ri < 2 * NQ; ++ri)
        xstore(sent4, rs, have_in ? sbase + in_off[ri / NQ] + (unsigned)(ri % NQ) * SRC8 : OOB, coloc);
    }
    NABU_STAMP(1, 5);
    NABU_STAMP(1, 6);
  }
  dz_stores();
  // bias gradient / column maxima of my 64 gate columns over the unit's 16 rows
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    red[grow[ps] * 64 + (2 * dup) * 16 + gu] = db0[ps];
    red[grow[ps] * 64 + (2 * dup + 1) * 16 + gu] = db1[ps];
  }
  __syncthreads();
  if (tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < MXR16; ++r) sum += red[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    red[grow[ps] * 64 + (2 * dup) * 16 + gu] = am0[ps];
    red[grow[ps] * 64 + (2 * dup + 1) * 16 + gu] = am1[ps];
  }
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR16; ++r) m = fmaxf(m, red[r * 64 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = m;
  }
}

// ===========================================================================
// backward, the product split in TWO dimensions (the default at H = 512; NABU_PERSIST_MX16_BWD2=0 and smaller H: the
// kernel above).  With 16 rows the reduce-scatter above moves 96 KiB per workgroup and step (32 published, 32 summed, 32
// handed back).  Here workgroup (cg, kg) of a unit owns W_h[64 k of group kg] x [H columns of group cg] (4 column
// groups x P/4 k groups: the same registers); per step
//   A  the dh of the 16 units it does the gate math for = FOUR pieces (one per column group) of 1 KiB;
//   its dz as bf16-plane cells [c'/8][48 = plane*16 + row] (c' = 4 unit + gate), 6 KiB;
//   B  the dz cells of ITS column group, 48 KiB, fetched straight into the operand layout (the forward pattern), the four
//      waves split the columns, partial sums meet in LDS behind the step's only barrier;
//   four 1-KiB pieces out.  63 KiB per workgroup and step, almost all reads of full lines.
// Rings: B 4 slots handed back by the producer two steps later, A 3 slots handed back by the reader — safe by causality:
// nobody can publish dz(s) before EVERYBODY has published dz(s+1) (its four A sources multiplied step s+1 against the
// cells of all four column groups), which in turn means everybody's sources finished step s+2, i.e. read slot s+2; a
// reader's hand-back of step s is performed before any of its waves publishes dz(s-1) (each wave publishes its own
// rows' cells behind its own poll of step s-1, issued behind its reset stores), and an A piece is written again only by
// a workgroup multiplying dz(s-2).  (The 8-row version of this split was measured and NOT adopted:
// tools/experiments/variants/lstm_persist_mx2.hip)
template <int H>
struct Mx16Bwd2Lds {
  static constexpr int KROW = 64 + 4;                            // floats per (wave, row): 64 k + pad
  static constexpr int PART = 0;                                 // [2][4 waves][16 rows][KROW]
  static constexpr int XST = PART + 2 * 4 * MXR16 * KROW;        // [2 parities][2 passes][4][256] saved values
  static constexpr int RED = XST + 2 * 2 * 4 * 256;              // [16 rows][64]
  static constexpr int FLAG = RED + 16 * 64;
  static constexpr int TOTAL = FLAG + 4;
};
constexpr int MX16RA = 3, MX16RB = 4;

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx16_bwd2_kernel(PersistArgs p) {
  using L = Mx16Bwd2Lds<H>;
  constexpr int P = H / UC;
  constexpr int KW = H / 4;          // columns c' of my column group multiplied by one wave
  constexpr int NKS = KW / 32;
  static_assert(NKS >= 1 && P % 4 == 0, "mx16 backward (2-D): H >= 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *part = smem + L::PART, *xst = smem + L::XST, *red = smem + L::RED;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int b0 = shard * MXR16;
  const int T = p.T;
  const int cg = slot & 3, kg = slot >> 2;
  const int Gb = 64 * kg + 16 * cg;                         // first of the 16 units whose gate math is mine
  const int n = lane & 15, q = lane >> 4;                   // matrix phase: n = batch row
  // gate identity: lane & 7 = 2 s4 + dup; kq = k quad of the A piece; rows 2 w + r2 (pass 0) and 8 + 2 w + r2 (pass 1)
  const int dup = lane & 1, s4 = (lane >> 1) & 3, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int gu = 4 * kq + s4;
  int grow[2], gb[2], n_g[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    grow[ps] = 8 * ps + 2 * w + r2;
    gb[ps] = b0 + grow[ps];
    n_g[ps] = gb[ps] < p.B ? p.len[gb[ps]] : 0;
  }

  // A operands: W^T planes of my block.  Row m = output k = 64 kg + 16 t + n; c' = cg H + w KW + 32 j + 8 q + e
  u32x4 Wp[3][4][NKS];
  {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float *Wh = p.kernel[dir] + ((size_t)p.D + 64 * kg + 16 * t + n) * 4 * H;
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
        float x[8];
        const int u0 = (cg * H + w * KW + 32 * j + 8 * q) >> 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[(size_t)(e & 3) * H + u0 + (e >> 2)];
        mx_split8(x, Wp[0][t][j], Wp[1][t][j], Wp[2][t][j]);
      }
    }
  }
  float dc_state[2] = {0.f, 0.f};
  float db0[2] = {0.f, 0.f}, db1[2] = {0.f, 0.f}, am0[2] = {0.f, 0.f}, am1[2] = {0.f, 0.f};
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // ring A: slot = [dest P][src 4][16 rows][4 k quads] x 16 bytes; ring B: slot = [c'/8][48 cells] x 16 bytes
  constexpr int NSL = 3 * MXR16;
  constexpr size_t A_SLOT = (size_t)P * 4 * MXR16 * 64, B_SLOT = (size_t)(4 * H / 8) * NSL * 16;
  char *const ubase = p.xbuf + (size_t)unit * (MX16RA * A_SLOT + MX16RB * B_SLOT);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, (int)(MX16RA * A_SLOT), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(ubase + MX16RA * A_SLOT, 0, (int)(MX16RB * B_SLOT), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  unsigned a_in[2], b_out[2];
  const int bpl = lane & 3;
  const bool b_pub = bpl < 3;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    a_in[ps] = (unsigned)((((size_t)slot * 4 + s4) * MXR16 + grow[ps]) * 64 + kq * 16);
    b_out[ps] = (unsigned)((((size_t)(Gb >> 1) + 2 * kq + (s4 >> 1)) * NSL + bpl * MXR16 + grow[ps]) * 16);
  }
  // A output piece (after the cross-wave sum: wave w = tile w, lane = (row, quad)): dest (cg = w, my kg)
  const int orow = lane >> 2, okq = lane & 3;
  const unsigned a_out = (unsigned)((((size_t)(4 * kg + w) * 4 + cg) * MXR16 + orow) * 64 + okq * 16);
  constexpr int KGW = KW / 8;
  constexpr unsigned KSTEP_BYTES = 4 * NSL * 16;
  const unsigned off0 = (unsigned)((((size_t)cg * (H / 8) + (size_t)w * KGW + q) * NSL + n) * 16);      // plane: + 256

  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  unsigned goff[2], coff[2], doff[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    goff[ps] = (unsigned)(((size_t)gb[ps] * T * 4 * H + (size_t)(2 * dup) * H + Gb + gu) * 4);
    coff[ps] = (unsigned)(((size_t)gb[ps] * T * H + Gb + gu) * 4);
    doff[ps] = (unsigned)(((size_t)gb[ps] * T * 2 * H + (size_t)dir * H + Gb + gu) * 4);
  }
  auto fetch_part = [&](int s, int idx) {
    const int ps = idx >> 2, part_i = idx & 3;
    const bool act = s >= 0 && s < n_g[ps];
    const int t = dir ? n_g[ps] - 1 - s : s;
    const int tc = dup == 0 ? t : (dir ? t + 1 : t - 1);
    const bool want_c = act && (dup == 0 || s > 0);
    float *st = xst + ((s & 1) * 2 + ps) * 1024 + 64 * w;
    if (part_i == 0) prefetch_lds_b32(rg, act ? goff[ps] + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part_i == 1) prefetch_lds_b32(rg, act ? goff[ps] + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
    if (part_i == 2) prefetch_lds_b32(rc, want_c ? coff[ps] + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + 512);
    if (part_i == 3) prefetch_lds_b32(rd, (act && dup == 0) ? doff[ps] + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 768);
  };
  for (int i = 0; i < 8; ++i) fetch_part(p.max_len - 1, i);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  float d_0[2] = {0.f, 0.f}, d_1[2] = {0.f, 0.f};
  int d_t[2] = {0, 0};
  bool d_any = false;
  auto dz_stores = [&]() {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const bool ok = d_any && gb[ps] < p.B && !(p.dbg & 128);
      const unsigned o = ok ? goff[ps] + (unsigned)d_t[ps] * (unsigned)(16 * H) : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0[ps]), rsg, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1[ps]), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
    }
  };
  auto timed_out = [&](unsigned long long &t_fail, int &fails, int code) -> bool {
    if (fails == 0) t_fail = wall_clock64();
    if ((++fails & 7) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t_fail > p.timeout_ticks) {
      if (lane == 0) {
        flag[0] = 1;
        __hip_atomic_store(p.status, code + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return true;
    }
    return false;
  };

  for (int s = p.max_len - 1; s >= 0; --s) {
    NABU_STAMP(1, 0);
    // (a) the four pieces of dh for my units, both passes
    u32x4 v[2] = {zero4, zero4};
    const unsigned abase = (unsigned)(((s + 1) % MX16RA) * A_SLOT);
    const bool have_in = s + 1 < p.max_len && !(p.dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      bool first = true;
      for (;;) {
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(ra, abase + a_in[0], 0, 16);
        v[1] = __builtin_amdgcn_raw_buffer_load_b128(ra, abase + a_in[1], 0, 16);
        if (first) { dz_stores(); first = false; }
        const bool ok = __all(mx_max4(mx_max4(0u, v[0]), v[1]) != SENT);
        if (ok) break;
        if (timed_out(t_fail, fails, 2)) break;
      }
    } else {
      dz_stores();
      wait_vm<0>();
    }
    NABU_STAMP(1, 1);
    unsigned cell[2][3];      // my packed pair words (gates 2 dup, 2 dup + 1) of both passes, per plane
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      mxf32x4 sum = __builtin_bit_cast(mxf32x4, v[ps]);
      // sum over the four sources (lanes 2 s4 + dup; both lanes of a pair hold the same piece): fixed order
      sum.x += mx_dpp<DPP_XOR2>(sum.x); sum.y += mx_dpp<DPP_XOR2>(sum.y);
      sum.z += mx_dpp<DPP_XOR2>(sum.z); sum.w += mx_dpp<DPP_XOR2>(sum.w);
      sum.x += mx_dpp<DPP_HALF_MIRROR>(sum.x); sum.y += mx_dpp<DPP_HALF_MIRROR>(sum.y);
      sum.z += mx_dpp<DPP_HALF_MIRROR>(sum.z); sum.w += mx_dpp<DPP_HALF_MIRROR>(sum.w);
      const float dh = sel4(s4, sum.x, sum.y, sum.z, sum.w);
      const float *st = xst + ((s & 1) * 2 + ps) * 1024 + tid;
      const float sA = st[0], sB = st[256], sC = st[512], sD = st[768];
      const float pA = mx_dpp<DPP_XOR1>(sA), pB = mx_dpp<DPP_XOR1>(sB), pC = mx_dpp<DPP_XOR1>(sC), pD = mx_dpp<DPP_XOR1>(sD);
      const float gi = dup ? pA : sA, gj = dup ? pB : sB, gf = dup ? sA : pA, go = dup ? sB : pB;
      const float c = dup ? pC : sC, cprev = dup ? sC : pC, dout = dup ? pD : sD;
      const bool act_g = s < n_g[ps];
      const float tc = fast_tanh(c);
      const float dht = dout + dh;
      const float dct = dc_state[ps] + dht * go * (1.f - tc * tc);
      float d0 = 0.f, d1 = 0.f;
      if (act_g) {
        d0 = dup ? dct * cprev * gf * (1.f - gf) : dct * gj * gi * (1.f - gi);
        d1 = dup ? dht * tc * go * (1.f - go) : dct * gi * (1.f - gj * gj);
        dc_state[ps] = dct * gf;
      }
      db0[ps] += d0; db1[ps] += d1;
      am0[ps] = fmaxf(am0[ps], fabsf(d0)); am1[ps] = fmaxf(am1[ps], fabsf(d1));
      mx_split3x2(d0, d1, cell[ps][0], cell[ps][1], cell[ps][2]);
      const int t_g = dir ? n_g[ps] - 1 - s : s;
      d_0[ps] = d0; d_1[ps] = d1; d_t[ps] = act_g ? t_g : s;
    }
    d_any = true;
    // I am the only reader of my A pieces: hand them back
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) xstore(sent4, ra, have_in ? abase + a_in[ps] : OOB, coloc);
    if (s == 0) break;      // no step in front of the first: nothing to multiply

    // (c) publish dz(s) as plane cells: a quad holds the 8 values c' = 4 u .. 4 u + 7 of two units in lane order; quad lane
    // = plane collects the plane's four pair words; hand back my cells of two steps ago (header)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const unsigned ph = cell[ps][0], pm = cell[ps][1], pl = cell[ps][2];
      const u32x4 v0 = {mx_dppu<0x00>(ph), mx_dppu<0x55>(ph), mx_dppu<0xAA>(ph), mx_dppu<0xFF>(ph)};
      const u32x4 v1 = {mx_dppu<0x00>(pm), mx_dppu<0x55>(pm), mx_dppu<0xAA>(pm), mx_dppu<0xFF>(pm)};
      const u32x4 v2 = {mx_dppu<0x00>(pl), mx_dppu<0x55>(pl), mx_dppu<0xAA>(pl), mx_dppu<0xFF>(pl)};
      const u32x4 pv = bpl == 0 ? v0 : bpl == 1 ? v1 : v2;
      xstore(pv, rb, b_pub ? (unsigned)((s % MX16RB) * B_SLOT) + b_out[ps] : OOB, coloc);
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
      xstore(sent4, rb, (b_pub && s + 2 < p.max_len) ? (unsigned)(((s + 2) % MX16RB) * B_SLOT) + b_out[ps] : OOB, coloc);
    NABU_STAMP(1, 2);

    // (d) the dz planes of my column group: the poll loop is the operand fetch
    u32x4 bp[3][NKS];
    {
      const unsigned bbase = (unsigned)((s % MX16RB) * B_SLOT);
      unsigned long long t_fail = 0;
      int fails = 0;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            bp[pl][j] = __builtin_amdgcn_raw_buffer_load_b128(rb, bbase + off0 + pl * 256u + j * KSTEP_BYTES, 0, 16);
#pragma unroll
        for (int j = 0; j < NKS; ++j)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) mx = mx_max4(mx, bp[pl][j]);
        const bool ok = __all(mx != SENT);       // (examined on every path out of the loop: lstm_persist_mx2.hip)
        if (ok || (p.dbg & 1)) break;
        if (timed_out(t_fail, fails, 1)) break;
      }
    }
    NABU_STAMP(1, 3);
    // (e) product: 4 k tiles x NKS k-steps x {Wl.h, Wm.m, Wh.l, Wm.h, Wh.m, Wh.h}; next step's saved values are
    // requested from inside the matrix stream
    mxf32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
#pragma unroll
      for (int g = 0; g < 6; ++g) {
        constexpr int WPL[6] = {2, 1, 0, 1, 0, 0}, BPL[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MX_MFMA(Wp[WPL[g]][t][j], bp[BPL[g]][j], acc[t]);
        const int slot_i = 6 * j + g;
        if (slot_i < 8) { fetch_part(s - 1, slot_i); __builtin_amdgcn_sched_barrier(0); }
      }
    }
    if (6 * NKS < 8)
      for (int i = 6 * NKS; i < 8; ++i) fetch_part(s - 1, i);
    NABU_STAMP(1, 4);
    // partial sums of my column quarter -> LDS [wave][row n][k = 16 t + 4 q + i]
    float *const pbuf = part + (s & 1) * (4 * MXR16 * L::KROW);
    {
      float *d = pbuf + ((size_t)(w * MXR16 + n)) * L::KROW + 4 * q;
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<mxf32x4 *>(d + 16 * t) = acc[t];
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    // (f) the four 16-k pieces: wave w sums tile w over the waves and sends it to workgroup (cg = w, my kg)
    {
      const float *pr = pbuf + (size_t)orow * L::KROW + 16 * w + 4 * okq;
      mxf32x4 o = *reinterpret_cast<const mxf32x4 *>(pr);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) o += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR16 * L::KROW);
      xstore(__builtin_bit_cast(u32x4, o), ra, (unsigned)((s % MX16RA) * A_SLOT) + a_out, coloc);
    }
    NABU_STAMP(1, 9);
    NABU_STAMP(1, 5);
    NABU_STAMP(1, 6);
  }
  dz_stores();
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    red[grow[ps] * 64 + (2 * dup) * 16 + gu] = db0[ps];
    red[grow[ps] * 64 + (2 * dup + 1) * 16 + gu] = db1[ps];
  }
  __syncthreads();
  if (tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < MXR16; ++r) sum += red[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + Gb + (tid & 15)] = sum;
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    red[grow[ps] * 64 + (2 * dup) * 16 + gu] = am0[ps];
    red[grow[ps] * 64 + (2 * dup + 1) * 16 + gu] = am1[ps];
  }
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR16; ++r) m = fmaxf(m, red[r * 64 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + Gb + (tid & 15)] = m;
  }
}

// measured (64 rows, us per sequential step, 2-D against 1-D): H = 512 3.29 against 4.22; H = 256 2.71 against 2.47;
// H = 128 2.53 against 2.17 — the two hand-offs of the 2-D split only pay where the 1-D reduce-scatter is large
static bool mx16_bwd2_on(int H) {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_MX16_BWD2"); env = e ? atoi(e) : 1; }
  return env != 0 && H >= 512;
}
static size_t mx16_bwd2_ring_bytes(int H) {
  const size_t P = H / UC;
  return (size_t)MXNU * ((size_t)MX16RA * P * 4 * MXR16 * 64 + (size_t)MX16RB * (4 * H / 8) * 3 * MXR16 * 16);
}

// ===========================================================================
// host side (called from lstm_persist.hip's run_chunk through lstm_mx_launch)
size_t lstm_mx16_ring_bytes(bool fwd, int H) {
  const size_t P = H / UC;
  if (!fwd && mx16_bwd2_on(H)) return mx16_bwd2_ring_bytes(H);
  return fwd ? (size_t)MXNU * RING * 3 * MXR16 * H * 2 : (size_t)MXNU * MXRINGB16 * P * P * MXR16 * UC * 4;
}

template <typename K>
static int mx16_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[8] = {};
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  int blocks = -1;
  for (const Seen &c : seen)
    if (c.fn == fn && c.dev == dev) blocks = c.blocks;
  if (blocks < 0) {
    if (lds > 48 * 1024) NABU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NABU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 256, lds));
    for (Seen &c : seen)
      if (!c.fn) { c = Seen{fn, dev, blocks}; break; }
  }
  if (blocks < 1 || grid > NCU)
    return fail(NABU_EUNSUP, "persistent LSTM (mx16): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// one launch over 33 .. 64 rows; `a` comes filled from run_chunk (nshard = ceil(B / 16))
int lstm_mx16_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
#define NABU_MX16_CASE(h)                                                                                          \
  case h:                                                                                                          \
    if (fwd) return mx16_launch(lstm_mx16_fwd_kernel<h>, a, grid, Mx16FwdLds<h>::TOTAL * sizeof(float), stream, dry);          \
    return mx16_bwd2_on(h) ? mx16_launch(lstm_mx16_bwd2_kernel<h>, a, grid, Mx16Bwd2Lds<h>::TOTAL * sizeof(float), stream, dry) \
                          : mx16_launch(lstm_mx16_bwd_kernel<h>, a, grid, Mx16BwdLds<h>::TOTAL * sizeof(float), stream, dry);
  switch (H) {
    NABU_MX16_CASE(128)
    NABU_MX16_CASE(256)
    NABU_MX16_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mx16): unsupported H=%d", H);
}

}  // namespace nabu
