// lstm_persist_mx.hip — the persistent recurrence with its product on the 16-bit matrix pipe
// (v_mfma_f32_16x16x32_bf16) over EXACTLY split operands, 8 batch rows per unit, ONE 256-thread workgroup
// per CU.  Round 4; protocol (sentinel rings, XCC-id handshake, bounded spins) as in lstm_persist.hip.
//
// WHY.  The exact-fp32 4x4x1 form (lstm_persist.hip) issues 128 matrix instructions per wave and step at the
// fp32 rate (0.72 us of a 2.35 us step, two workgroups per CU in each other's way).  A fp32 number is the sum of
// three bf16 numbers (8 + 8 + 8 significand bits: h = rne(x), m = rne(x - h), l = x - h - m, nothing lost), so
// x . w = sum over plane pairs of exact 16-bit products accumulated in fp32 by the matrix pipe, which is 16 times
// faster per multiply-add.  Kept: the six pairs down to 2^-16 of the leading one (hh, hm, mh, mm, hl, lh) plus lm,
// which rides along for free; dropped: ml, ll (<= 2^-24 |x||w|, below one fp32 rounding of the term itself).
//
// GEOMETRY.  unit = (direction, 8 batch rows) = the P = H/16 workgroups of ONE XCD (block b -> unit b % 8 -> XCD
// b % 8, verified at run time like before); a workgroup owns 16 hidden units = 64 gate columns.  B <= 32 rows per
// launch (8 units); larger batches run as consecutive launches over 32-row chunks.
// The 16-wide N side of the instruction holds 8 rows x 2 PLANES: B1 = [h_h | h_m], B2 = [h_l | 0]; the M side 16
// weight columns of one plane.  Per (16 columns x 32 k): Wh.B1, Wm.B1, Wl.B1, Wh.B2 — four instructions give the
// seven products; the two N halves of the result are added with one DPP row rotation.  64 instructions per wave
// and step at ~17 cycles instead of 2 x 128 at 12.
//
// FORWARD step: the exchange slot of a unit is an array of 16-byte CELLS = 8 consecutive k of one (plane, row),
// [k / 8][24 = plane * 8 + row]: a lane's B operand is ONE cell — the poll loop IS the operand fetch, straight into the
// instruction's register layout, full 128-byte lines, no LDS staging.  Wave w multiplies its quarter of k; partial sums
// meet in LDS behind the step's only barrier; wave w then finishes rows 2w, 2w+1 (lane = (row, unit), all four gates),
// splits h into planes, assembles one cell per plane with DPP and publishes with ONE store instruction.
// BACKWARD step: reduce-scatter of the partial dh (fp32 pieces [dest][src][row][4 k], ring of 3: header of
// lstm_persist_mx.h); lane = (row, unit, gate pair) after an 8-lane DPP butterfly over the sources; dz planes go to LDS
// in the B-operand layout; product dz[8 rows x 64 columns] . W^T against this workgroup's [64 x H] slice in two halves,
// tile t = destination workgroup t; hand-back and next step's prefetch ride in the first half's matrix stream.
// BOTH: the results of a step (activations, c, h; dz) go to HBM at the top of the NEXT step, behind its exchange loads,
// with always-issued stores — between a publish and the next poll a wave's memory queue holds exchange traffic only
// (DESIGN.md section 5.1 for the measurements behind every one of these choices).
#include "lstm_persist_mx.h"

namespace nabu {

#define MX_STAMP(pass, i)                                                          \
  do {                                                                             \
    if constexpr (DBG) {                                                           \
      if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)          \
        p.status[320 + 32 * (pass) + (i)] = (int)(wall_clock64());                 \
    }                                                                              \
  } while (0)

// ===========================================================================
// forward
template <int H>
struct MxFwdLds {
  static constexpr int ROWF = 17 * 4;                        // floats per (wave, row): 16 units x 4 gates + pad
  static constexpr int PART = 0;                             // [2][4 waves][8 rows][ROWF]
  static constexpr int XST = PART + 2 * 4 * MXR * ROWF;      // [2][2][256] prefetched x-projection
  static constexpr int FLAG = XST + 2 * 2 * 256;
  static constexpr int TOTAL = FLAG + 4;
};

// DBG: the instantiation that honours NABU_PERSIST_DEBUG (phase stamps, ablations); the production one carries none of
// those tests — a dozen scalar branches per step of a wave that has nothing to hide them behind
template <int H, bool DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx_fwd_kernel(PersistArgs p) {
  const int dbg = DBG ? p.dbg : 0;
  using L = MxFwdLds<H>;
  constexpr int P = H / UC;
  constexpr int KW = H / 4;          // k values multiplied by one wave
  constexpr int NKS = KW / 32;       // k-steps of 32 per wave
  static_assert(NKS >= 1, "mx forward: H >= 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *part = smem + L::PART, *xst = smem + L::XST;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mx_identity(&unit, &slot);
  if (unit >= NU) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * MXR;
  const int T = p.T;
  // matrix-phase identity: n = N index (plane half, row), q = k group (B) / column group (D)
  const int n = lane & 15, q = lane >> 4;
  // finishing identity (lanes 0..31 of every wave): row 2w + r2, unit u16 — and, for the prefetch, gate pair gp
  const int u16 = lane & 15, r2 = (lane >> 4) & 1, gp = lane >> 5;
  const int frow = 2 * w + r2, fb = b0 + frow;
  const int n_f = fb < p.B ? p.len[fb] : 0;
  const bool fin = lane < 32;

  // this lane's slice of W_h as three bf16 planes, A operands: column (gate c, unit U0 + n), k = w KW + 32 j + 8 q + e
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

  // exchange slot of a unit: cells of 16 bytes = 8 consecutive k of one (plane, row): [k / 8][24 = plane * 8 + row]
  // — a k group's 24 cells are 384 contiguous bytes, a wave's k range 6 KiB of full 128-byte lines
  const size_t slot_bytes = (size_t)24 * H * 2;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  // B operands of this lane: B1 (planes h | m) of k-step j = cell (kg0 + 4 j + q, n); the l planes of TWO k-steps
  // travel in one load (lanes n < 8: k-step 2 jp, the others: 2 jp + 1) and are separated after the load
  constexpr int KGW = KW / 8;                      // k groups per wave
  constexpr int NPR = (NKS + 1) / 2;               // l-plane loads
  constexpr unsigned KSTEP_BYTES = 4 * 24 * 16;    // 4 k groups
  const unsigned off1 = (unsigned)((((size_t)w * KGW + q) * 24 + n) * 16);
  const unsigned off2 = (unsigned)((((size_t)w * KGW + q) * 24 + 16 + (n & 7)) * 16) + (unsigned)(n >> 3) * KSTEP_BYTES;
  // my published piece (finishing half, lanes u16 & 7 = plane 0..2): units U0 + (u16 & 8) .. + 7 of row frow
  const int ppl = u16 & 7;
  const bool pub_lane = fin && ppl < 3;
  const unsigned pub_off = (unsigned)((((size_t)(U0 >> 3) + (u16 >> 3)) * 24 + ppl * 8 + frow) * 16);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // x-projection of step s (bias included), one step ahead, by LDS-DMA (lstm_persist.hip: PER-STEP PREFETCH):
  // lane (u16, r2, gp) fetches gates 2 gp and 2 gp + 1 of (row frow, unit u16); the finishing lane reads all four
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const unsigned goff = (unsigned)(((size_t)fb * T * 4 * H + (size_t)(2 * gp) * H + U0 + u16) * 4);
  auto fetch_x_part = [&](int s, int part) {
    const int t = dir ? n_f - 1 - s : s;
    const bool act = s < n_f && !(dbg & 64);
    float *st = xst + (s & 1) * 512 + 64 * w;
    if (part == 0) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part == 1) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
  };
  auto fetch_x = [&](int s) {
    fetch_x_part(s, 0);
    fetch_x_part(s, 1);
  };
  fetch_x(0);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // result stores: both lane halves hold the same values — the lower stores gates i, j and c, the upper f, o and h
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64();
  // RESULT STORES ARE DEFERRED: the values of step s (activations, c, h) go to HBM at the top of step s + 1, BEHIND
  // that step's exchange loads in the wave's in-order memory queue — between a publish and the next poll the queue
  // holds nothing but exchange traffic.  (The prefetched x-projection needs no explicit claim either: it is older
  // than the exchange loads every step waits for, and vector-memory operations complete in issue order.)
  float d_g0 = 0.f, d_g1 = 0.f, d_v = 0.f;
  int d_t = 0, d_to = 0;
  bool d_act = false, d_any = false;
  // (always issued, inactive lanes out of range: the wait counts of the loads in front stay exact)
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.cs[dir], 0, (int)((size_t)p.B * T * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.B * T * 2 * H * 4), 0x00020000);
  const unsigned coff = (unsigned)(((size_t)fb * T * H + U0 + u16) * 4);
  const unsigned ooff = (unsigned)(((size_t)fb * T * 2 * H + (size_t)dir * H + U0 + u16) * 4);
  const bool st_ok = fb < p.B && !(dbg & 128);
  auto result_stores = [&]() {
    const bool on = d_any && st_ok;
    const unsigned go_ = (on && d_act) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_g0), rsg, go_, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_g1), rsg, go_ == OOB ? OOB : go_ + (unsigned)(4 * H), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_v), rsc, (on && d_act && !gp) ? coff + (unsigned)d_t * (unsigned)(4 * H) : OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_v), rso, (on && gp) ? ooff + (unsigned)d_to * (unsigned)(8 * H) : OOB, 0, 0);
  };

  for (int s = 0; s < p.max_len; ++s) {
    MX_STAMP(0, 0);
    mxf32x4 acc[4];
    unsigned long long t_fail = 0;
    int fails = 0;
    // (a) h_{s-1} as planes: the poll loop IS the operand fetch — the loads of the wave's k range (6 KiB of full lines)
    // are repeated until no word holds the sentinel
    u32x4 b1[NKS], bl[NPR];
#pragma unroll
    for (int j = 0; j < NKS; ++j) b1[j] = zero4;
#pragma unroll
    for (int j = 0; j < NPR; ++j) bl[j] = zero4;
    if (s > 0 && !(dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes);
      bool first = true;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
          b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off1 + j * KSTEP_BYTES, 0, 16);
          if ((j & 1) == 0)
            bl[j / 2] = __builtin_amdgcn_raw_buffer_load_b128(
                rs, (j + 1 < NKS || n < 8) ? base + off2 + j * KSTEP_BYTES : OOB, 0, 16);
        }
        if (first) { result_stores(); first = false; }     // step s - 1's results, behind the loads
#pragma unroll
        for (int j = 0; j < NKS; ++j) mx = mx_max4(mx, b1[j]);
#pragma unroll
        for (int j = 0; j < NPR; ++j) mx = mx_max4(mx, bl[j]);
        if (__all(mx != SENT)) break;
        if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2) p.status[320 + 20] += 1;
        // a failed round: bounded-spin bookkeeping (the clock is first read here)
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
      wait_vm<0>();        // (no exchange loads to order the prefetch: s = 0, or the no-waiting experiment)
    }
    MX_STAMP(0, 1);
    // (b) product: 4 column tiles (gate c) x NKS k-steps x {Wl.B1, Wm.B1, Wh.B2, Wh.B1}; the l planes of k-steps
    // 2 jp (lanes n < 8) and 2 jp + 1 (the others) arrived in one register set.  Next step's x-projection (HBM
    // latency: as early as possible) is requested from inside the matrix stream, one instruction behind each of the
    // first two groups of matrix instructions.
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !(dbg & 2)) {
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
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[2][c][j], b1[j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[1][c][j], b1[j], acc[c]);
        if (j == 0) { fetch_x_part(s + 1, 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[0][c][j], b2, acc[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MX_MFMA(Wp[0][c][j], b1[j], acc[c]);
      }
    } else {
      fetch_x(s + 1);
    }
    MX_STAMP(0, 2);
    if ((dbg & 4096) && (unit == 0 || unit == 4) && lane == 0 && s == p.max_len / 2 + 1)
      p.status[384 + 64 * (unit != 0) + 2 * slot + 1] = (int)wall_clock64() + (w << 28);   // product done (last wave wins)
    // the two plane halves of N: lanes n and n ^ 8 end with the same sums (row n & 7; units 4 q + i, gate c)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c].x += mx_dpp<DPP_ROR8>(acc[c].x);
      acc[c].y += mx_dpp<DPP_ROR8>(acc[c].y);
      acc[c].z += mx_dpp<DPP_ROR8>(acc[c].z);
      acc[c].w += mx_dpp<DPP_ROR8>(acc[c].w);
    }
    // partial sums -> LDS [wave][row][unit][4 gates]: lanes n < 8 write units 4 q + {0, 1}, the others 4 q + {2, 3}
    float *const pbuf = part + (s & 1) * (4 * MXR * L::ROWF);
    {
      const bool lo = n < 8;
      float *d = pbuf + ((size_t)(w * MXR + (n & 7))) * L::ROWF + (4 * q + (lo ? 0 : 2)) * 4;
      const mxf32x4 v0 = {lo ? acc[0].x : acc[0].z, lo ? acc[1].x : acc[1].z, lo ? acc[2].x : acc[2].z, lo ? acc[3].x : acc[3].z};
      const mxf32x4 v1 = {lo ? acc[0].y : acc[0].w, lo ? acc[1].y : acc[1].w, lo ? acc[2].y : acc[2].w, lo ? acc[3].y : acc[3].w};
      *reinterpret_cast<mxf32x4 *>(d) = v0;
      *reinterpret_cast<mxf32x4 *>(d + 4) = v1;
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    MX_STAMP(0, 3);

    // (c) gates of (row frow, unit u16): both lane halves compute the same
    mxf32x4 z;
    {
      const float *xs = xst + (s & 1) * 512 + 64 * w + (lane & 31);
      z = (mxf32x4){xs[0], xs[256], xs[32], xs[256 + 32]};
      const float *pr = pbuf + (size_t)frow * L::ROWF + u16 * 4;
#pragma unroll
      for (int ww = 0; ww < 4; ++ww) z += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR * L::ROWF);
    }
    const float gi = fast_sigmoid(z.x), gj = fast_tanh(z.y), gf = fast_sigmoid(z.z + 1.0f), go = fast_sigmoid(z.w);
    const bool act = s < n_f;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act) { c_state = c_new; h_state = h_new; }

    // (d) publish h_s as planes (frozen rows republish): the pair words of a plane sit in the even lanes of an
    // 8-lane group; lane 8 g + pl collects the four words of plane pl -> ONE 16-byte store instruction per wave
    {
      unsigned pl[3], pr[3];
      mx_split3(h_state, pl[0], pl[1], pl[2]);
#pragma unroll
      for (int i = 0; i < 3; ++i) pr[i] = pl[i] | (mx_dppu<DPP_XOR1>(pl[i]) << 16);     // even lanes: units u, u + 1
      const u32x4 v0 = {pr[0], mx_dppu<0x102>(pr[0]), mx_dppu<0x104>(pr[0]), mx_dppu<0x106>(pr[0])};   // lane 8 g
      const u32x4 v1 = {mx_dppu<0x111>(pr[1]), mx_dppu<0x101>(pr[1]), mx_dppu<0x103>(pr[1]), mx_dppu<0x105>(pr[1])};   // 8 g + 1
      const u32x4 v2 = {mx_dppu<0x112>(pr[2]), pr[2], mx_dppu<0x102>(pr[2]), mx_dppu<0x104>(pr[2])};   // 8 g + 2
      const u32x4 pv = ppl == 0 ? v0 : ppl == 1 ? v1 : v2;
      xstore(pv, rs, (pub_lane && s + 1 < p.max_len) ? (unsigned)((s % RING) * slot_bytes) + pub_off : OOB, coloc);
      // hand back my pieces of h_{s-2} (ordering: lstm_persist.hip, forward (d))
      xstore(sent4, rs, (pub_lane && s >= 2) ? (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off : OOB, coloc);
    }
    if ((dbg & 4096) && (unit == 0 || unit == 4) && tid == 0 && s == p.max_len / 2)
      p.status[384 + 64 * (unit != 0) + 2 * slot] = (int)wall_clock64();
    MX_STAMP(0, 4);
    // (e) results of this step: stored at the top of the next one (see result_stores)
    {
      const int t_g = dir ? n_f - 1 - s : s;
      d_any = true; d_act = act; d_t = t_g; d_to = act ? t_g : s;
      d_g0 = gp ? gf : gi;
      d_g1 = gp ? go : gj;
      d_v = gp ? (act ? h_new : 0.f) : c_new;
    }
    MX_STAMP(0, 5);
  }
  result_stores();
  if ((dbg & 4) && blockIdx.x == 0 && tid == 0) {   // effective shader clock over the sequence
    p.status[320 + 26] = (int)(__builtin_readcyclecounter() - clk0);
    p.status[320 + 27] = (int)(wall_clock64() - wall0);
  }
}

// ===========================================================================
// backward
template <int H>
struct MxBwdLds {
  static constexpr int DROWB = 64 * 2 + 16;                  // bytes per slot row of dz planes: 64 columns bf16 + pad
  static constexpr int DZ = 0;                               // [2][24][DROWB] bytes
  static constexpr int XST = (2 * 24 * DROWB + 15) / 16 * 4; // floats: [2][4][256] prefetched saved values
  static constexpr int RED = XST + 2 * 4 * 256;              // [8 rows][64] floats, final reductions
  static constexpr int FLAG = RED + 8 * 64;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H, bool DBG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mx_bwd_kernel(PersistArgs p) {
  const int dbg = DBG ? p.dbg : 0;
  using L = MxBwdLds<H>;
  constexpr int P = H / UC;
  constexpr int NT = P / 4;          // 16-k output tiles (= destination workgroups) per wave
  constexpr int NQ = P / 8;          // source pieces per lane
  static_assert(NT >= 2 && NQ >= 1, "mx backward: H >= 128");
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
  const int U0 = slot * UC, b0 = shard * MXR;
  const int T = p.T;
  const int n = lane & 15, q = lane >> 4;                 // matrix-phase identity
  static_assert(NQ <= 4, "mx backward: the slot hand-back rides in the second half's 8 groups");
  // exchange / gate identity: source group s8, k quad kq, row 2 w + r2; after the butterfly: unit 4 kq + (s8 >> 1),
  // gate pair dup (0: i, j; 1: f, o)
  const int s8 = lane & 7, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int grow = 2 * w + r2, gb = b0 + grow;
  const int gu = 4 * kq + (s8 >> 1), dup = s8 & 1;
  const int n_g = gb < p.B ? p.len[gb] : 0;

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
  float dc_state = 0.f;
  float db0 = 0.f, db1 = 0.f, am0 = 0.f, am1 = 0.f;   // bias gradient / largest |dz| of my two gate columns, my row
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;

  // ring slot = [dest P][src P][8 rows][4 k quads] x 16 bytes
  const size_t piece_bytes = (size_t)MXR * UC * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * MXRINGB * slot_bytes, 0, (int)(MXRINGB * slot_bytes), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};
  const unsigned in_off = (unsigned)((size_t)slot * block_bytes + ((size_t)s8 * MXR + grow) * 64 + kq * 16);

  // saved forward values of step s, one step ahead: A, B = the activations of my two gates, C = c (dup 0) / c_prev
  // (dup 1), D = dout (dup 0); the pair exchanges what the other needs
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gb * T * 4 * H + (size_t)(2 * dup) * H + U0 + gu) * 4);
  const unsigned coff = (unsigned)(((size_t)gb * T * H + U0 + gu) * 4);
  const unsigned doff = (unsigned)(((size_t)gb * T * 2 * H + (size_t)dir * H + U0 + gu) * 4);
  auto fetch_part = [&](int s, int part) {
    const bool act = s >= 0 && s < n_g;
    const int t = dir ? n_g - 1 - s : s;
    const int tc = dup == 0 ? t : (dir ? t + 1 : t - 1);
    const bool want_c = act && (dup == 0 || s > 0);
    float *st = xst + (s & 1) * 1024 + 64 * w;
    if (part == 0) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    if (part == 1) prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) + (unsigned)(4 * H) : OOB, smem, st + 256);
    if (part == 2) prefetch_lds_b32(rc, want_c ? coff + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + 512);
    if (part == 3) prefetch_lds_b32(rd, (act && dup == 0) ? doff + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 768);
  };
  auto fetch = [&](int s) {
    for (int i = 0; i < 4; ++i) fetch_part(s, i);
  };
  fetch(p.max_len - 1);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  // dz of step s goes to HBM at the top of step s - 1, behind that step's exchange loads (see the forward kernel:
  // between a publish and the next poll the wave's memory queue holds exchange traffic only); always issued
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  const bool st_ok = gb < p.B && !(dbg & 128);
  float d_0 = 0.f, d_1 = 0.f;
  int d_t = 0;
  bool d_any = false;
  auto dz_stores = [&]() {
    const unsigned o = (d_any && st_ok) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0), rsg, o, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  for (int s = p.max_len - 1; s >= 0; --s) {
    MX_STAMP(1, 0);
    // (a) reduce-scatter input: the partial products of step s + 1 addressed to my units
    u32x4 v[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) v[i] = zero4;
    const unsigned base = (unsigned)(((s + 1) % MXRINGB) * slot_bytes) + in_off;
    const bool have_in = s + 1 < p.max_len && !(dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      bool first = true;
      // the pieces of this step cannot be there sooner than a hand-off after my own publish: a first round issued at
      // once fails and costs the memory queue a round trip (2.34 -> 2.27 us per step with ~110 ns of idling first)
      __builtin_amdgcn_s_sleep(4);
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int i = 0; i < NQ; ++i)
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(8 * i) * (unsigned)(MXR * 64), 0, 16);
        if (first) { dz_stores(); first = false; }
#pragma unroll
        for (int i = 0; i < NQ; ++i) mx = mx_max4(mx, v[i]);
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
      wait_vm<0>();     // (no exchange loads to order the prefetched values: first step, or the no-waiting experiment)
    }
    MX_STAMP(1, 1);
    if ((dbg & 4096) && (unit == 0 || unit == 4) && lane == 0 && s == p.max_len / 2 - 1)
      p.status[384 + 64 * (unit != 0) + 2 * slot + 1] = (int)wall_clock64() + (w << 28);   // poll done (last wave wins)
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

    // (b) gate gradients of (row, unit): the pair shares its saved values (prefetched a step ahead: older than the
    // exchange loads above, and vector-memory operations complete in issue order)
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
    char *const dzb = dzs + (s & 1) * (24 * L::DROWB);
    {
      unsigned ph, pm, pl;
      mx_split3x2(d0, d1, ph, pm, pl);
      const unsigned o = (unsigned)grow * L::DROWB + (unsigned)(4 * gu + 2 * dup) * 2;
      *reinterpret_cast<unsigned *>(dzb + o) = ph;
      *reinterpret_cast<unsigned *>(dzb + o + 8 * L::DROWB) = pm;
      *reinterpret_cast<unsigned *>(dzb + o + 16 * L::DROWB) = pl;
    }
    {   // dz of this step: stored at the top of the next one; padded frames get 0
      const int t_g = dir ? n_g - 1 - s : s;
      d_any = true; d_0 = d0; d_1 = d1; d_t = act_g ? t_g : s;
    }
    MX_STAMP(1, 2);
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    MX_STAMP(1, 3);
    if (s > 0) {
      // (c) partial dh of step s - 1: dz planes [16 slots x 64 columns] against W^T, tile t -> destination NT w + t,
      // in two halves of NT / 2 tiles; lanes n < 8 publish the first tiles of a half, the others (same sums) the
      // rest: piece (dest, me)[row n & 7][quad q].  The step's other memory instructions ride in the matrix stream:
      // next step's saved values (HBM latency: as early as possible) and the slot hand-back (I am the only reader of
      // my pieces) in the first half — a hand-back right in front of the publish delays it (ub/xchg_rs.hip).
      u32x4 b1[2], b2[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        b1[j] = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)n * L::DROWB + 64 * j + 16 * q);
        const u32x4 l = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)(16 + (n & 7)) * L::DROWB + 64 * j + 16 * q);
        b2[j] = n < 8 ? l : zero4;
      }
      constexpr int HT = NT / 2;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        mxf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int t = 0; t < HT; ++t)
              acc[t] = MX_MFMA(Wp[g == 0 ? 2 : g == 1 ? 1 : 0][hf * HT + t][j], g == 2 ? b2[j] : b1[j], acc[t]);
            const int slot_i = 4 * j + g;     // one memory instruction behind every group of matrix instructions
            if (hf == 0 && slot_i < 4) fetch_part(s - 1, slot_i);
            if (hf == 0 && slot_i >= 4 && slot_i - 4 < NQ)
              xstore(sent4, rs, have_in ? base + (unsigned)(8 * (slot_i - 4)) * (unsigned)(MXR * 64) : OOB, coloc);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          acc[t].x += mx_dpp<DPP_ROR8>(acc[t].x);
          acc[t].y += mx_dpp<DPP_ROR8>(acc[t].y);
          acc[t].z += mx_dpp<DPP_ROR8>(acc[t].z);
          acc[t].w += mx_dpp<DPP_ROR8>(acc[t].w);
        }
        if (hf == 0) MX_STAMP(1, 4);
        constexpr int QT = HT / 2 > 0 ? HT / 2 : 1;      // tiles per lane half and product half
        const int t0 = NT * w + hf * HT + (n < 8 ? 0 : HT / 2);
        const unsigned pbase = (unsigned)((s % MXRINGB) * slot_bytes + (size_t)t0 * block_bytes + (size_t)slot * piece_bytes +
                                          (size_t)(n & 7) * 64 + q * 16);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const mxf32x4 lo = acc[t], hi = acc[HT / 2 + t < HT ? HT / 2 + t : t];
          const mxf32x4 o = {n < 8 ? lo.x : hi.x, n < 8 ? lo.y : hi.y, n < 8 ? lo.z : hi.z, n < 8 ? lo.w : hi.w};
          // (HT = 1, H = 128: one tile per half, published by the lanes n < 8 only)
          xstore(__builtin_bit_cast(u32x4, o), rs, (HT >= 2 || n < 8) ? pbase + (unsigned)t * (unsigned)block_bytes : OOB, coloc);
        }
      }
      MX_STAMP(1, 9);
      if ((dbg & 4096) && (unit == 0 || unit == 4) && tid == 0 && s == p.max_len / 2)
        p.status[384 + 64 * (unit != 0) + 2 * slot] = (int)wall_clock64();     // last publish issued (wave 0)
    } else {
#pragma unroll
      for (int i = 0; i < NQ; ++i)
        xstore(sent4, rs, have_in ? base + (unsigned)(8 * i) * (unsigned)(MXR * 64) : OOB, coloc);
    }
    MX_STAMP(1, 5);
    MX_STAMP(1, 6);
  }
  dz_stores();
  // bias gradient / column maxima of my 64 gate columns over the unit's 8 rows
  __syncthreads();
  red[grow * 64 + (2 * dup) * 16 + gu] = db0;
  red[grow * 64 + (2 * dup + 1) * 16 + gu] = db1;
  __syncthreads();
  if (tid < 64) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) sum += red[r * 64 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 4) * H + U0 + (tid & 15)] = sum;
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

// ===========================================================================
// host side (called from lstm_persist.hip's run_chunk)
static int mx_env() {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_MX"); env = e ? atoi(e) : 1; }
  return env;
}

// the geometry needs a whole MI355X: 8 XCDs of 32 CUs, one workgroup per CU
static bool mx_device_ok() {
  static thread_local int cached_dev = -1;
  static thread_local bool ok = false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (dev != cached_dev) {
    int cus = 0;
    ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= NCU;
    if (!ok) (void)hipGetLastError();
    cached_dev = dev;
  }
  return ok;
}

bool lstm_mx_supported(int B, int H) {
  if (!mx_env() || lstm_persist_exact() || !mx_device_ok()) return false;
  return (H == 128 || H == 256 || H == 512) && B >= 1;
}
int lstm_mx_chunk_rows() { return MXR * MXNU / 2; }   // 32 batch rows per launch

size_t lstm_mx_ring_bytes(bool fwd, int H) {
  const size_t P = H / UC;
  return fwd ? (size_t)MXNU * RING * 24 * H * 2 : (size_t)MXNU * MXRINGB * P * P * MXR * UC * 4;
}

template <typename K>
static int mx_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[16] = {};
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
    return fail(NABU_EUNSUP, "persistent LSTM (mx): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;          // validation pass (lstm_persist.hip, run)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// one launch over B <= 32 rows; `a` comes filled from run_chunk (nshard = ceil(B / 8))
int lstm_mx_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
#define NABU_MX_CASE(h)                                                                                          \
  case h:                                                                                                        \
    if (a.dbg)                                                                                                   \
      return fwd ? mx_launch(lstm_mx_fwd_kernel<h, true>, a, grid, MxFwdLds<h>::TOTAL * sizeof(float), stream, dry) \
                 : mx_launch(lstm_mx_bwd_kernel<h, true>, a, grid, MxBwdLds<h>::TOTAL * sizeof(float), stream, dry); \
    return fwd ? mx_launch(lstm_mx_fwd_kernel<h, false>, a, grid, MxFwdLds<h>::TOTAL * sizeof(float), stream, dry) \
               : mx_launch(lstm_mx_bwd_kernel<h, false>, a, grid, MxBwdLds<h>::TOTAL * sizeof(float), stream, dry);
  switch (H) {
    NABU_MX_CASE(128)
    NABU_MX_CASE(256)
    NABU_MX_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mx): unsupported H=%d", H);
}

}  // namespace nabu
