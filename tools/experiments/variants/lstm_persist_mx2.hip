// lstm_persist_mx2.hip — MEASURED AND NOT ADOPTED (round 4; kept here for the record, not part of the library: to try it
// again, copy it into nabu_amd/csrc/ and dispatch lstm_mx2_bwd_launch from run_chunk of lstm_persist.hip for the
// backward pass of <= 32 rows, with lstm_mx2_ring_bytes in the workspace size).  Parity-green on the cfg2 / cfg1 layer
// shapes at its first run; 2.32 us per sequential step at the cfg2 layer shape against 2.25 for lstm_mx_bwd_kernel
// (H = 256: 2.0 against 1.76, H = 128: 1.9 against 1.51): two small hand-offs (0.20 + 0.44 us) cost what the one large
// reduce-scatter costs.
//
// The backward recurrence of lstm_persist_mx.hip with the product split in TWO dimensions
// (8 batch rows per unit, one workgroup per CU, a unit per XCD; bf16-plane arithmetic unchanged).
//
// WHY.  lstm_mx_bwd_kernel gives a workgroup 64 gate columns and ALL H output k: its partial dh is H x 8 fp32 = 16 KiB,
// published in P pieces, summed over P sources and handed back — 48 KiB of L2 traffic per workgroup and step, 1.5 MB
// per XCD and step, and that hand-off (~0.8 us) is what its 2.25 us step waits for.  Here workgroup (cg, kg) of a unit
// owns the block W_h[64 k of group kg] x [H columns of group cg] (4 column groups x P/4 k groups; the same 128 KiB of
// weights as planes in registers).  Per step:
//   A  (reduce, small): dh of the 16 units this workgroup does the gate math for = the sum of FOUR pieces (one per
//      column group) of 512 B;
//   gate math: dz of those 16 units, as bf16 planes, published as 16-byte cells [c'/8][plane*8 + row] (c' = 4 unit +
//      gate) — 3 KiB per workgroup;
//   B  (all-gather, the forward kernel's pattern): the dz planes of MY column group, 24 KiB, fetched straight into the
//      matrix instruction's operand layout; the four waves split the columns, their partial sums meet in LDS behind
//      the step's only barrier;
//   the four 16-k pieces of the result go to the four workgroups (column group = piece index, same k group) that do
//   the gate math for those units.
// 31 KiB per workgroup and step, almost all of it reads of full lines; two small hand-offs instead of one large.
//
// RINGS (the data is the flag, sentinel words, as everywhere):
//   B: 4 slots, cells handed back by their PRODUCER two steps after publishing (like the forward ring).  Safe because
//      a workgroup can publish dz(s) only after ALL workgroups have published dz(s+1): its four A sources multiplied
//      step s+1 against the dz(s+1) cells of all four column groups = of everybody.  Everybody having published dz(s+1)
//      means everybody's A sources finished step s+2, i.e. every workgroup has read the cells of slot s+2.  And a reset
//      is performed before the same lane's next publish is visible: the B loads of the step in between were issued
//      behind it (in-order completion), the forward kernel's argument.
//   A: 3 slots, pieces handed back by their single reader.  A piece reset in step s is written again in step s-2 by a
//      workgroup that multiplies dz(s-2), which nobody can publish before EVERY workgroup — the reader included, all
//      four of its waves, each publishing its own rows' cells behind its own poll of step s-1, whose loads were issued
//      behind its reset stores — has published dz(s-1).
#include "lstm_persist_mx.h"

namespace nabu {

template <int H>
struct Mx2Lds {
  static constexpr int KROW = 64 + 4;                           // floats per (wave, row): 64 k + pad
  static constexpr int PART = 0;                                // [2][4 waves][8 rows][KROW]
  static constexpr int XST = PART + 2 * 4 * MXR * KROW;         // [2][4][256] prefetched saved values
  static constexpr int RED = XST + 2 * 4 * 256;                 // [8 rows][64] final reductions
  static constexpr int FLAG = RED + 8 * 64;
  static constexpr int TOTAL = FLAG + 4;
};

constexpr int MX2RINGA = 3;     // reduce ring (pieces of partial dh)
constexpr int MX2RINGB = 4;     // all-gather ring (dz plane cells)

size_t lstm_mx2_ring_a_bytes(int H) { return (size_t)MX2RINGA * (H / UC) * 4 * MXR * UC * 4; }     // per unit
size_t lstm_mx2_ring_b_bytes(int H) { return (size_t)MX2RINGB * (4 * H / 8) * 24 * 16; }            // per unit
size_t lstm_mx2_ring_bytes(int H) { return (size_t)MXNU * (lstm_mx2_ring_a_bytes(H) + lstm_mx2_ring_b_bytes(H)); }

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx2_bwd_kernel(PersistArgs p) {
  using L = Mx2Lds<H>;
  constexpr int P = H / UC;
  constexpr int KW = H / 4;          // columns c' of my column group multiplied by one wave
  constexpr int NKS = KW / 32;       // k-steps of 32 per wave
  constexpr int NPR = (NKS + 1) / 2;
  static_assert(NKS >= 1 && P % 4 == 0, "mx2 backward: H >= 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *part = smem + L::PART, *xst = smem + L::XST, *red = smem + L::RED;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int b0 = shard * MXR;
  const int T = p.T;
  const int cg = slot & 3, kg = slot >> 2;                  // column group, k group of my block of W_h
  const int Gb = 64 * kg + 16 * cg;                         // first of the 16 units whose gate math is mine
  const int n = lane & 15, q = lane >> 4;                   // matrix-phase identity
  // gate identity: lane & 7 = 2 s4 + dup (s4: A source = column group of the piece, then unit 4 kq + s4; dup: gate pair),
  // kq = k quad of the A piece, row 2 w + r2
  const int dup = lane & 1, s4 = (lane >> 1) & 3, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int grow = 2 * w + r2, gb = b0 + grow;
  const int gu = 4 * kq + s4;
  const int n_g = gb < p.B ? p.len[gb] : 0;

  // A operands: W^T planes of my block.  Row m = output k = 64 kg + 16 t + n; reduction index c' = cg H + w KW + 32 j +
  // 8 q + e = 4 unit + gate
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
  float dc_state = 0.f;
  float db0 = 0.f, db1 = 0.f, am0 = 0.f, am1 = 0.f;
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // ring A: slot = [dest P][src 4][8 rows][4 k quads] x 16 bytes; ring B: slot = [c'/8][24 cells] x 16 bytes
  constexpr size_t A_SLOT = (size_t)P * 4 * MXR * 64, B_SLOT = (size_t)(4 * H / 8) * 24 * 16;
  char *const ubase = p.xbuf + (size_t)unit * (MX2RINGA * A_SLOT + MX2RINGB * B_SLOT);
  __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(ubase, 0, (int)(MX2RINGA * A_SLOT), 0x00020000);
  __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(ubase + MX2RINGA * A_SLOT, 0, (int)(MX2RINGB * B_SLOT), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // my A input piece: (dest me, src s4)[row grow][quad kq] (both lanes of a gate pair load the same piece)
  const unsigned a_in = (unsigned)((((size_t)slot * 4 + s4) * MXR + grow) * 64 + kq * 16);
  // my A output piece (after the cross-wave sum: wave w = tile w, lanes 0..31 = (row, quad)): dest (cg = w, my kg)
  const int orow = (lane >> 2) & 7, okq = lane & 3;
  const unsigned a_out = (unsigned)((((size_t)(4 * kg + w) * 4 + cg) * MXR + orow) * 64 + okq * 16);
  // my B output cell: units Gb + 4 kq + 2 (s4 >> 1) + {0, 1}, plane = lane & 3 (< 3), row grow
  const int bpl = lane & 3;
  const bool b_pub = bpl < 3;
  const unsigned b_out = (unsigned)((((size_t)(Gb >> 1) + 2 * kq + (s4 >> 1)) * 24 + bpl * 8 + grow) * 16);
  // my B operands (the forward kernel's): cells of column group cg, k-steps of wave w
  constexpr int KGW = KW / 8;
  constexpr unsigned KSTEP_BYTES = 4 * 24 * 16;
  const unsigned off1 = (unsigned)((((size_t)cg * (H / 8) + (size_t)w * KGW + q) * 24 + n) * 16);
  const unsigned off2 = (unsigned)((((size_t)cg * (H / 8) + (size_t)w * KGW + q) * 24 + 16 + (n & 7)) * 16) + (unsigned)(n >> 3) * KSTEP_BYTES;

  // saved forward values of step s, one step ahead (lstm_persist_mx.hip): A, B = activations of my two gates, C = c /
  // c_prev, D = dout (dup 0)
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gb * T * 4 * H + (size_t)(2 * dup) * H + Gb + gu) * 4);
  const unsigned coff = (unsigned)(((size_t)gb * T * H + Gb + gu) * 4);
  const unsigned doff = (unsigned)(((size_t)gb * T * 2 * H + (size_t)dir * H + Gb + gu) * 4);
  auto fetch_part = [&](int s, int part_i) {
    const bool act = s >= 0 && s < n_g;
    const int t = dir ? n_g - 1 - s : s;
    const int tc = dup == 0 ? t : (dir ? t + 1 : t - 1);
    const bool want_c = act && (dup == 0 || s > 0);
    float *st = xst + (s & 1) * 1024 + 64 * w;
    if (part_i == 0) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part_i == 1) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
    if (part_i == 2) prefetch_lds_b32(rc, want_c ? coff + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + 512);
    if (part_i == 3) prefetch_lds_b32(rd, (act && dup == 0) ? doff + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 768);
  };
  for (int i = 0; i < 4; ++i) fetch_part(p.max_len - 1, i);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // dz of step s goes to HBM at the top of step s - 1, behind that step's first exchange load; always issued
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  const bool st_ok = gb < p.B && !(p.dbg & 128);
  float d_0 = 0.f, d_1 = 0.f;
  int d_t = 0;
  bool d_any = false;
  auto dz_stores = [&]() {
    const unsigned o = (d_any && st_ok) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0), rsg, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
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
    // (a) the four pieces of dh for my units (partial products of step s + 1)
    u32x4 v = zero4;
    const unsigned abase = (unsigned)(((s + 1) % MX2RINGA) * A_SLOT) + a_in;
    const bool have_in = s + 1 < p.max_len && !(p.dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      bool first = true;
      for (;;) {
        v = __builtin_amdgcn_raw_buffer_load_b128(ra, abase, 0, 16);
        if (first) { dz_stores(); first = false; }
        if (__all(mx_max4(0u, v) != SENT)) break;
        if (timed_out(t_fail, fails, 2)) break;
      }
    } else {
      dz_stores();
      wait_vm<0>();
    }
    NABU_STAMP(1, 1);
    mxf32x4 ps = __builtin_bit_cast(mxf32x4, v);
    // sum over the four sources: lanes 2 s4 + dup; both lanes of a pair hold the same piece, so i <-> i ^ 2 (s4 0-1,
    // 2-3) and i <-> 7 - i (s4 <-> 3 - s4) complete the sum in every lane, in a fixed order
    ps.x += mx_dpp<DPP_XOR2>(ps.x); ps.y += mx_dpp<DPP_XOR2>(ps.y);
    ps.z += mx_dpp<DPP_XOR2>(ps.z); ps.w += mx_dpp<DPP_XOR2>(ps.w);
    ps.x += mx_dpp<DPP_HALF_MIRROR>(ps.x); ps.y += mx_dpp<DPP_HALF_MIRROR>(ps.y);
    ps.z += mx_dpp<DPP_HALF_MIRROR>(ps.z); ps.w += mx_dpp<DPP_HALF_MIRROR>(ps.w);
    const float dh = sel4(s4, ps.x, ps.y, ps.z, ps.w);

    // (b) gate gradients of (row grow, unit Gb + gu): the pair shares its saved values
    const float *st = xst + (s & 1) * 1024 + tid;
    const float sA = st[0], sB = st[256], sC = st[512], sD = st[768];
    const float pA = mx_dpp<DPP_XOR1>(sA), pB = mx_dpp<DPP_XOR1>(sB), pC = mx_dpp<DPP_XOR1>(sC), pD = mx_dpp<DPP_XOR1>(sD);
    const float gi = dup ? pA : sA, gj = dup ? pB : sB, gf = dup ? sA : pA, go = dup ? sB : pB;
    const float c = dup ? pC : sC, cprev = dup ? sC : pC, dout = dup ? pD : sD;
    const bool act_g = s < n_g;
    const float tc = fast_tanh(c);
    const float dht = dout + dh;
    const float dct = dc_state + dht * go * (1.f - tc * tc);
    float d0 = 0.f, d1 = 0.f;
    if (act_g) {
      d0 = dup ? dct * cprev * gf * (1.f - gf) : dct * gj * gi * (1.f - gi);
      d1 = dup ? dht * tc * go * (1.f - go) : dct * gi * (1.f - gj * gj);
      dc_state = dct * gf;
    }
    db0 += d0; db1 += d1;
    am0 = fmaxf(am0, fabsf(d0)); am1 = fmaxf(am1, fabsf(d1));
    {
      const int t_g = dir ? n_g - 1 - s : s;
      d_any = true; d_0 = d0; d_1 = d1; d_t = act_g ? t_g : s;
    }
    // I am the only reader of my A piece: hand it back (both lanes of the pair write the same sentinel)
    xstore(sent4, ra, have_in ? abase : OOB, coloc);
    if (s == 0) break;      // no step in front of the first: nothing to multiply

    // (c) publish dz(s) as plane cells: a quad (lanes 2 s4 + dup, s4 in {0,1} or {2,3}) holds the 8 values c' = 4 u ..
    // 4 u + 7 of two units in lane order; lane (quad lane = plane) collects the plane's four pair words
    {
      unsigned ph, pm, pl;
      mx_split3x2(d0, d1, ph, pm, pl);
      const u32x4 v0 = {mx_dppu<0x00>(ph), mx_dppu<0x55>(ph), mx_dppu<0xAA>(ph), mx_dppu<0xFF>(ph)};
      const u32x4 v1 = {mx_dppu<0x00>(pm), mx_dppu<0x55>(pm), mx_dppu<0xAA>(pm), mx_dppu<0xFF>(pm)};
      const u32x4 v2 = {mx_dppu<0x00>(pl), mx_dppu<0x55>(pl), mx_dppu<0xAA>(pl), mx_dppu<0xFF>(pl)};
      const u32x4 pv = bpl == 0 ? v0 : bpl == 1 ? v1 : v2;
      xstore(pv, rb, b_pub ? (unsigned)((s % MX2RINGB) * B_SLOT) + b_out : OOB, coloc);
      // my cells of two steps ago are read by now (header)
      xstore(sent4, rb, (b_pub && s + 2 < p.max_len) ? (unsigned)(((s + 2) % MX2RINGB) * B_SLOT) + b_out : OOB, coloc);
    }
    NABU_STAMP(1, 2);

    // (d) the dz planes of my column group: the poll loop is the operand fetch (forward kernel)
    u32x4 b1[NKS], bl[NPR];
    {
      const unsigned bbase = (unsigned)((s % MX2RINGB) * B_SLOT);
      unsigned long long t_fail = 0;
      int fails = 0;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
          b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rb, bbase + off1 + j * KSTEP_BYTES, 0, 16);
          if ((j & 1) == 0)
            bl[j / 2] = __builtin_amdgcn_raw_buffer_load_b128(
                rb, (j + 1 < NKS || n < 8) ? bbase + off2 + j * KSTEP_BYTES : OOB, 0, 16);
        }
#pragma unroll
        for (int j = 0; j < NKS; ++j) mx = mx_max4(mx, b1[j]);
#pragma unroll
        for (int j = 0; j < NPR; ++j) mx = mx_max4(mx, bl[j]);
        // (the words are examined on EVERY path out of the loop: a path that leaves with loads formally pending makes
        // hipcc wait for them again in front of the matrix instructions — with counts that do not know about the
        // LDS-DMA prefetches issued in between, i.e. for those HBM loads too)
        const bool ok = __all(mx != SENT);
        if (ok || (p.dbg & 1)) break;
        if (timed_out(t_fail, fails, 1)) break;
      }
    }
    NABU_STAMP(1, 3);
    // (e) product: 4 k tiles x NKS k-steps x {Wl.B1, Wm.B1, Wh.B2, Wh.B1}; next step's saved values are requested from
    // inside the matrix stream
    mxf32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
      u32x4 b2;
      if ((j & 1) == 0) {
        b2 = n < 8 ? bl[j / 2] : zero4;
      } else {
        const u32x4 r = {mx_dppu<DPP_ROR8>(bl[j / 2].x), mx_dppu<DPP_ROR8>(bl[j / 2].y), mx_dppu<DPP_ROR8>(bl[j / 2].z),
                         mx_dppu<DPP_ROR8>(bl[j / 2].w)};
        b2 = n < 8 ? r : zero4;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MX_MFMA(Wp[2][t][j], b1[j], acc[t]);
      if (j == 0) { fetch_part(s - 1, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MX_MFMA(Wp[1][t][j], b1[j], acc[t]);
      if (j == 0) { fetch_part(s - 1, 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MX_MFMA(Wp[0][t][j], b2, acc[t]);
      if (j == 0) { fetch_part(s - 1, 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = MX_MFMA(Wp[0][t][j], b1[j], acc[t]);
      if (j == 0) { fetch_part(s - 1, 3); __builtin_amdgcn_sched_barrier(0); }
    }
    NABU_STAMP(1, 4);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t].x += mx_dpp<DPP_ROR8>(acc[t].x);
      acc[t].y += mx_dpp<DPP_ROR8>(acc[t].y);
      acc[t].z += mx_dpp<DPP_ROR8>(acc[t].z);
      acc[t].w += mx_dpp<DPP_ROR8>(acc[t].w);
    }
    // partial sums of my column quarter -> LDS [wave][row n & 7][k = 16 t + 4 q + i]: lanes n < 8 write tiles 0, 1,
    // the others (same sums) tiles 2, 3
    float *const pbuf = part + (s & 1) * (4 * MXR * L::KROW);
    {
      const bool lo = n < 8;
      float *d = pbuf + ((size_t)(w * MXR + (n & 7))) * L::KROW + (lo ? 0 : 32) + 4 * q;
      *reinterpret_cast<mxf32x4 *>(d) = lo ? acc[0] : acc[2];
      *reinterpret_cast<mxf32x4 *>(d + 16) = lo ? acc[1] : acc[3];
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    // (f) the four 16-k pieces: wave w sums tile w over the waves and sends it to workgroup (cg = w, my kg)
    {
      const float *pr = pbuf + (size_t)orow * L::KROW + 16 * w + 4 * okq;
      mxf32x4 o = *reinterpret_cast<const mxf32x4 *>(pr);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) o += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR * L::KROW);
      xstore(__builtin_bit_cast(u32x4, o), ra, lane < 32 ? (unsigned)((s % MX2RINGA) * A_SLOT) + a_out : OOB, coloc);
    }
    NABU_STAMP(1, 9);
    NABU_STAMP(1, 5);
    NABU_STAMP(1, 6);
  }
  dz_stores();
  // bias gradient / column maxima of my 16 units x 4 gates over the unit's 8 rows
  __syncthreads();
  red[grow * 64 + (2 * dup) * 16 + gu] = db0;
  red[grow * 64 + (2 * dup + 1) * 16 + gu] = db1;
  __syncthreads();
  if (tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) sum += red[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + Gb + (tid & 15)] = sum;
  }
  __syncthreads();
  red[grow * 64 + (2 * dup) * 16 + gu] = am0;
  red[grow * 64 + (2 * dup + 1) * 16 + gu] = am1;
  __syncthreads();
  if (tid < 64) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) m = fmaxf(m, red[r * 64 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + Gb + (tid & 15)] = m;
  }
}

// ===========================================================================
template <typename K>
static int mx2_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[4] = {};
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  int blocks = -1;
  for (const Seen &c : seen)
    if (c.fn == fn && c.dev == dev) blocks = c.blocks;
  if (blocks < 0) {
    NABU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, 256, lds));
    for (Seen &c : seen)
      if (!c.fn) { c = Seen{fn, dev, blocks}; break; }
  }
  if (blocks < 1 || grid > NCU)
    return fail(NABU_EUNSUP, "persistent LSTM (mx2): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// the backward pass over B <= 32 rows; `a` comes filled from run_chunk (nshard = ceil(B / 8))
int lstm_mx2_bwd_launch(int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
#define NABU_MX2_CASE(h) \
  case h: return mx2_launch(lstm_mx2_bwd_kernel<h>, a, grid, Mx2Lds<h>::TOTAL * sizeof(float), stream, dry);
  switch (H) {
    NABU_MX2_CASE(128)
    NABU_MX2_CASE(256)
    NABU_MX2_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mx2): unsupported H=%d", H);
}

}  // namespace nabu
