// lstm_persist_mxh.hip — the persistent whole-sequence recurrence with its product as THREE fp16 plane
// products of row-scaled operands (the arithmetic of the step's dense products, gemm_pk.hip NP = 2 / DESIGN.md
// section 4.4) instead of seven bf16 plane products: half the matrix instructions per step.  Round 4.
//
// ARITHMETIC.  A factor that is constant along the reduction index factors out of a dot product, so every row of an
// operand carries its own power-of-two scale s = 2^(14 - floor(log2 amax)) (amax = the row's largest magnitude: the
// scaled row lies in (-2^15, 2^15), fp16 overflows at 65504) and x s = h + l with h = rne16(x s), l = rne16(x s - h):
// 22 significant bits for every element within 2^13 of its row's maximum, an absolute error <= 2^-40 of that maximum
// below.  Products kept: h.h, h.l, l.h (l.l rides along for free); accumulation in fp32 inside the matrix pipe, one
// rounding per 32-k instruction; the result is multiplied by 1 / (s_a s_b), exact.
//   forward:  W_h's gate columns are scaled once per launch (amax over all H rows of the column); h_(t-1) needs no
//             measurement: |h| = |o tanh c| < 1 + 2^-22 by construction, s = 2^14.
//   backward: W_h^T per output k over this workgroup's 64 gate columns, once per launch; dz per batch row over the same
//             64 columns EVERY STEP (a 32-lane maximum: four DPP steps and one swizzle) — the scale only has to be
//             constant inside one workgroup's reduction, the partial dh it publishes are plain fp32.
// Against float64 the recurrence with this product is as close as with the exact-fp32 kernels
// (tests/test_hip_fullsize.py::test_plane_recurrence_is_as_close_to_float64_as_the_fp32_kernels).
//
// GEOMETRY.  unit = (direction, 8 batch rows) = the 32 workgroups (256 threads, ONE per CU, one wave per SIMD) of ONE XCD:
// block b -> unit b % 8 -> XCD b % 8 (checked at run time: a unit that is not co-located publishes write-through); a
// workgroup owns 16 hidden units = 64 gate columns = a [H x 64] slice of W_h held as planes in registers for the whole
// sequence.  Forward exchange slot of a unit: cells of 16 bytes = 8 consecutive k of one (plane, row),
// [k / 8][16 = plane * 8 + row] — a lane's 16-byte load IS its B operand, the poll loop is the operand fetch; the N side
// of every instruction is B = [h | l] of 8 rows, 32 instructions per wave and step: W_l.B, W_h.B per (16 columns x 32 k).
// PROTOCOL ("the data is the flag", rings, in-order arguments, deferred result stores): header of lstm_persist.hip and
// lstm_persist_mxh.h; the bf16-plane predecessors of these kernels (rounds 4a: lstm_persist_mx.hip, lstm_persist_mx16.hip)
// are parked under tools/experiments/variants/.
#include "lstm_persist_mxh.h"

namespace nabu {

// experiment (off in the library build): the l plane of W_h in accumulation registers, fed to the matrix instruction
// from there (lstm_persist_mxh.h) — measured SLOWER in this kernel (1.35 against 1.27 us per forward step): hipcc then
// pairs the h-plane instruction of k-step j with the l-plane one of j + 1 on the SAME accumulators back to back
#ifndef MXH_ACC_L
#define MXH_ACC_L 0
#endif
#ifndef MXH_TAG_MODE
#define MXH_TAG_MODE 0
#endif
#ifndef MXH_BWD_ORDER
#define MXH_BWD_ORDER 0
#endif
// where the forward kernel issues the packed companions' stores of step s - 1 (EMIT): 2 = inside the matrix stream of step s,
// 1 = with the other result stores behind the first poll round's loads, 3 = at the end of step s - 1, behind its publish
#ifndef MXH_EMIT_AT
#define MXH_EMIT_AT 2
#endif
#if MXH_ACC_L
#define MXH_MFMA_L(a, b, c) mxf_mfma_acc(a, b, c)
#else
#define MXH_MFMA_L(a, b, c) MXH_MFMA(a, b, c)
#endif

#define MXH_STAMP(pass, i)                                                         \
  do {                                                                             \
    if constexpr (DBG) {                                                           \
      if ((dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)          \
        p.status[320 + 32 * (pass) + (i)] = (int)(wall_clock64());                 \
    }                                                                              \
  } while (0)

// ===========================================================================
// forward
template <int H>
struct MxhFwdLds {
  static constexpr int ROWF = 17 * 4;                        // floats per (wave, row): 16 units x 4 gates + pad
  static constexpr int PART = 0;                             // [2][4 waves][8 rows][ROWF]
  static constexpr int XST = PART + 2 * 4 * MXR * ROWF;      // [2][4 waves][64 lanes x 4] prefetched x-projection
                                                             // XIN: [3][4 waves][2 k-steps][64 lanes x 4] planes of x_t
  static constexpr int FLAG = XST + 3 * 4 * 512;
  static constexpr int TOTAL = FLAG + 4;
};

// XIN = true — the first layer's narrow input (D <= 64 features) is projected INSIDE the kernel (z = [x_t,
// h_(t-1)] . kernel + bias is ONE product in the reference's cell): x arrives as two fp16 planes per frame, scaled like h
// (lstm_mxh_prepare_x below), as two more k-steps of the same matrix stream — wave w multiplies them against gate w's
// columns of Wx into the accumulators its recurrent partial sums are in, so nothing downstream changes.  What this
// replaces: a [B T, 4H] product per direction in front of the kernel (gemm_smallk_kernel: write-bound, 0.26 ms per cfg2
// step for 8 % of the recurrent product's work) and its read-back, one float per thread and step.
// EMIT = true — the layer's output ALSO as packed f16x3 operands (lstm_persist.h, EmitArgs; include/nabu_hip.h, ABI version
// 3): the two fp16 planes of h 2^14 this kernel computes anyway for its exchange go, with the step's other results, to
// (1) the next layer's input operand (one 16-byte store per publishing lane: 8 units of one plane of one frame row),
// (2) its transposed operand and (3) this layer's own h_(t-1)^T operand (2-byte stores, one plane per lane half: the
// reduction index of both is the frame).  What that replaces: pk_pack_rows / pk_pack_cols / the h^T part of
// pk_pack_cols_multi — 0.44 ms of passes over `out` per cfg2 step.  Frozen rows write zeros at the frames they do not
// have, exactly as `out` does; the launch requires max_len == T (every frame of every row is written by the kernel).
template <int H, bool DBG, bool XIN = false, int EM = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mxh_fwd_kernel(PersistArgs p) {
  const int dbg = DBG ? p.dbg : 0;
  constexpr bool EMIT = EM != 0;     // EM: bit 0 rows, bit 1 transposed (both or neither), bit 2 h^T
  using L = MxhFwdLds<H>;
  constexpr int P = H / UC;
  constexpr int KW = H / 4;          // k values multiplied by one wave
  constexpr int NKS = KW / 32;       // k-steps of 32 per wave
  static_assert(NKS >= 1, "mxh forward: H >= 128");
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

  // this lane's slice of W_h as two scaled fp16 planes, A operands: column (gate c, unit U0 + n), k = w KW + 32 j + 8 q + e.
  // The column's scale needs its largest magnitude over ALL k: lanes q (shuffles), then the four waves (LDS).
  // inv[c]: what the finished sum of column (c, U0 + n) is multiplied with — n = u16: the lane that multiplies column
  // n is the lane that finishes unit n.
  u32x4 Wp[2][4][NKS];
  u32x4 Wxp[2][2];        // XIN: gate w's columns of Wx, rows 32 j + 8 q + e, times the input's range g (a power of two)
  float inv[4], bias_f[4] = {0.f, 0.f, 0.f, 0.f};
  {
    const float *Wh = p.kernel[dir] + ((size_t)p.D + (size_t)w * KW + 8 * q) * 4 * H + U0 + n;
    // XIN: x is held as x 2^14 / g (|x| <= g: lstm_mxh_prepare_x), i.e. scaled like h, so that x . Wx and h . W_h share
    // accumulators and descale; what remains of g goes into the weights: the column's scale covers g Wx as well
    const float xg = XIN ? p.xscale[0] : 0.f;
    const float *Wx = p.kernel[dir] + (size_t)(8 * q) * 4 * H + U0 + n;
    float mx[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      mx[c] = 0.f;
#pragma unroll
      for (int j = 0; j < NKS; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[c] = fmaxf(mx[c], fabsf(Wh[((size_t)32 * j + e) * 4 * H + (size_t)c * H]));
      if constexpr (XIN) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = 32 * j + 8 * q + e;
            if (k < p.D) mx[c] = fmaxf(mx[c], xg * fabsf(Wx[((size_t)32 * j + e) * 4 * H + (size_t)c * H]));
          }
      }
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 16));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], 32));
      if (q == 0) part[w * 64 + c * 16 + n] = mx[c];
    }
    __syncthreads();
    float sc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float m = fmaxf(fmaxf(part[c * 16 + n], part[64 + c * 16 + n]), fmaxf(part[128 + c * 16 + n], part[192 + c * 16 + n]));
      sc[c] = mxh_scale_of(m);
      inv[c] = mxh_inv_scale_of(m) * MXH_HINV;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[((size_t)32 * j + e) * 4 * H + (size_t)c * H] * sc[c];
        mxh_split8(x, Wp[0][c][j], Wp[1][c][j]);
#if MXH_ACC_L
        mxf_pin_acc(Wp[1][c][j]);        // the l plane: accumulation registers (lstm_persist_mxh.h)
#endif
      }
    if constexpr (XIN) {
      const float scw = sel4(w, sc[0], sc[1], sc[2], sc[3]) * xg;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * j + 8 * q + e;
          x[e] = k < p.D ? Wx[((size_t)32 * j + e) * 4 * H + (size_t)w * H] * scw : 0.f;
        }
        mxh_split8(x, Wxp[0][j], Wxp[1][j]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) bias_f[c] = p.bias[dir][(size_t)c * H + U0 + (lane & 15)];
    }
  }
  float c_state = 0.f, h_state = 0.f;
  if (!unit_handshake(p, unit, slot, MXNU, P, flag)) return;
  const bool coloc = flag[1] != 0;
  clock_stamp(p, 0, 0);

  // exchange slot of a unit: cells of 16 bytes = 8 consecutive k of one (plane, row): [k / 8][16 = plane * 8 + row]
  // — a k group's 16 cells are 256 contiguous bytes, a wave's k range 4 KiB of full 128-byte lines
  const size_t slot_bytes = (size_t)16 * H * 2;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  constexpr int KGW = KW / 8;                      // k groups per wave
  constexpr unsigned KSTEP_BYTES = 4 * 16 * 16;    // 4 k groups
  const unsigned off1 = (unsigned)((((size_t)w * KGW + q) * 16 + n) * 16);
  // my published piece (finishing half, lanes u16 & 7 = plane 0, 1): units U0 + (u16 & 8) .. + 7 of row frow
  const int ppl = u16 & 7;
  const bool pub_lane = fin && ppl < 2;
  const unsigned pub_off = (unsigned)((((size_t)(U0 >> 3) + (u16 >> 3)) * 16 + ppl * 8 + frow) * 16);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // x-projection of step s (bias included), one step ahead, by LDS-DMA (lstm_persist.hip: PER-STEP PREFETCH):
  // lane (u16, r2, gp) fetches gates 2 gp and 2 gp + 1 of (row frow, unit u16); the finishing lane reads all four
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const unsigned goff = (unsigned)(((size_t)fb * T * 4 * H + (size_t)(2 * gp) * H + U0 + u16) * 4);
  // ONE 16-byte LDS-DMA load per lane and step (two 4-byte loads per lane before: 1.35 -> 1.31 us per step): lane (r =
  // lane >> 4, gate g, unit quad uq) of the lower half-wave fetches units 4 uq .. 4 uq + 3 of gate g of row 2 w + r — the
  // row this lane also finishes, so its length is n_f.  (The same for the backward kernel's four loads — three tensors,
  // hence global_load_lds_dwordx4 with 64-bit addresses — was built and measured slower: 2.06 against 2.03 us.)
  const unsigned goff4 = (unsigned)(((size_t)fb * T * 4 * H + (size_t)((lane >> 2) & 3) * H + U0 + 4 * (lane & 3)) * 4);
  // XIN: the planes of x_s, [b][t][plane][64 k] fp16 (lstm_mxh_prepare_x), one step ahead, EVERY wave its own copy (the
  // claim stays what it is for the projection: older than the exchange loads the step waits for): lane (n, q) of k-step
  // j fetches its B operand — 8 k of (plane n >> 3, row n & 7) — into [j][lane]: a lane reads back what it fetched
  const i32x4 rxp = raw_rsrc(XIN ? p.xplanes : nullptr, XIN ? (unsigned)((size_t)p.B * T * 256) : 0u);
  const int xrow = b0 + (n & 7);
  const int n_x = (XIN && xrow < p.B) ? p.len[xrow] : 0;
  const unsigned xoff = (unsigned)(((size_t)xrow * T * 2 + (n >> 3)) * 128 + 16 * q);
  auto fetch_x = [&](int s) {
    if constexpr (XIN) {
      const int t = dir ? n_x - 1 - s : s;
      const bool act = s < n_x && !(dbg & (64 | 1024));
      float *st = xst + (s % 3) * 2048 + 512 * w;
      prefetch_lds_b128(rxp, act ? xoff + (unsigned)t * 256u : OOB, smem, st);
      prefetch_lds_b128(rxp, act ? xoff + (unsigned)t * 256u + 64u : OOB, smem, st + 256);
    } else {
      const int t = dir ? n_f - 1 - s : s;
      const bool act = fin && s < n_f && !(dbg & 64);
      prefetch_lds_b128(rg, act ? goff4 + (unsigned)t * (unsigned)(16 * H) : OOB, smem, xst + (s & 1) * 1024 + 256 * w);
    }
  };
  // XIN: the planes are fetched TWO steps ahead (ring of three staging buffers per wave): they are multiplied at the very
  // top of a step, and one step (1.3 us) is less than an HBM round trip behind a cold start — a late fetch stalled the
  // wave in front of its first poll round (1.62 us per step inside the training step against 1.47 alone)
  fetch_x(0);
  if constexpr (XIN) fetch_x(1);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // RESULT STORES ARE DEFERRED to the top of the next step, behind its exchange loads; always
  // issued, inactive lanes out of range: the wait counts of the loads in front stay exact
  float d_g0 = 0.f, d_g1 = 0.f, d_v = 0.f;
  int d_t = 0, d_to = 0;
  bool d_act = false, d_any = false;
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.cs[dir], 0, (int)((size_t)p.B * T * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.B * T * 2 * H * 4), 0x00020000);
  const unsigned coff = (unsigned)(((size_t)fb * T * H + U0 + u16) * 4);
  const unsigned ooff = (unsigned)(((size_t)fb * T * 2 * H + (size_t)dir * H + U0 + u16) * 4);
  const bool st_ok = fb < p.B && !(dbg & 128);
  // EMIT: packed companions (always three stores per step; a companion that is not asked for has an empty descriptor)
  u32x4 e_pv = {0u, 0u, 0u, 0u};     // publishing lanes: my 16-byte piece of the step's planes (zero for a frozen row)
  unsigned e_w = 0u;                 // my plane's fp16 of (row frow, unit u16): plane gp
  const int ess = EMIT ? p.emit.stack_shift : 0;
  const int erow0 = EMIT ? (p.emit.b0 + fb) * (T >> ess) : 0;          // first packed row / reduction index of my batch row
  const int ek0 = EMIT ? (p.emit.b0 + fb) * T : 0;                     // ... of h^T
  const int ecol = dir * H + U0 + u16;                                 // my unit's k (x_rows) / row (x_cols) inside a frame
  const int ehrow = EMIT ? p.emit.hT_row0 + U0 + u16 : 0;
  const unsigned e_xr_lane = EMIT ? (unsigned)((((dir * H + U0) >> 4) * 2 + (u16 & 7)) * p.emit.x_rows_pad) * 32u : 0u;   // (u16 & 7 = plane of a publishing lane)
  const unsigned e_xr_par = EMIT ? (unsigned)(((2 * H) >> 4) * 2 * p.emit.x_rows_pad) * 32u : 0u;      // the odd frame of a stacked pair: k + 2H
  __amdgpu_buffer_rsrc_t rex = __builtin_amdgcn_make_buffer_rsrc(EMIT ? p.emit.x_rows : nullptr, 0, (EMIT && p.emit.x_rows) ? 0x7FFFFFF0 : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rec = __builtin_amdgcn_make_buffer_rsrc(EMIT ? p.emit.x_cols : nullptr, 0, (EMIT && p.emit.x_cols) ? 0x7FFFFFF0 : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t reh = __builtin_amdgcn_make_buffer_rsrc(EMIT ? p.emit.hT[dir] : nullptr, 0, (EMIT && p.emit.hT[dir]) ? 0x7FFFFFF0 : 0, 0x00020000);
  auto result_stores = [&]() {
    const bool on = d_any && st_ok;
    const unsigned go_ = (on && d_act) ? goff + (unsigned)d_t * (unsigned)(16 * H) : OOB;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_g0), rsg, go_, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_g1), rsg, go_ == OOB ? OOB : go_ + (unsigned)(4 * H), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_v), rsc, (on && d_act && !gp) ? coff + (unsigned)d_t * (unsigned)(4 * H) : OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_v), rso, (on && gp) ? ooff + (unsigned)d_to * (unsigned)(8 * H) : OOB, 0, 0);
  };
  // EMIT: the companions of step s - 1 are stored from INSIDE the matrix stream of step s (emit_part 0 / 1 / 2 between
  // groups of matrix instructions: the wave has issue slots to spare there, and the address path is done with the
  // scattered 2-byte stores long before the next poll round needs it), the last step's behind the loop
  int e_to = 0;
  bool e_any = false;
  // (address arithmetic: 24-bit multiplies — full rate — of a k-block number (< 2^24) with a uniform block stride (< 2^24),
  // everything that depends on the lane alone folded into constants in front of the loop)
  const unsigned e_ckb = EMIT ? 2u * p.emit.x_cols_pad * 32u : 0u, e_hkb = EMIT ? 2u * p.emit.hT_rows_pad * 32u : 0u;
  const unsigned e_clane = EMIT ? ((unsigned)gp * p.emit.x_cols_pad + (unsigned)ecol) * 32u : 0u;
  const unsigned e_cpar = (unsigned)(2 * H) * 32u;                       // the odd frame of a stacked pair: row + 2H
  const unsigned e_csw = ((unsigned)u16 >> 3) & 1u;                      // bit 3 of my row (2H, H, U0 are multiples of 16)
  const unsigned e_hlane = EMIT ? ((unsigned)gp * p.emit.hT_rows_pad + (unsigned)ehrow) * 32u : 0u;
  const unsigned e_hsw = ((unsigned)ehrow >> 3) & 1u;
  auto emit_part = [&](int part) {
    if constexpr (EMIT) {
      // the address arithmetic must be ISSUED here, between the matrix instructions: pure arithmetic floats to the top of
      // the basic block otherwise (in front of the first matrix instruction, i.e. onto the step's critical chain) — its
      // input passes through an opaque asm that the scheduling barrier orders
      int to = e_to;
      asm volatile("" : "+v"(to));
      const bool on = e_any && st_ok;
      // frame `to` of my batch row (a frozen row: the frame it does not have, value 0 — as `out`)
      const unsigned fr = (unsigned)to >> ess, par = (unsigned)to & (unsigned)ess;     // (ess is 0 or 1: the mask of the parity)
      const unsigned r = (unsigned)erow0 + fr;
      if (part == 0 && (EM & 1)) {
        // (1) next layer's input operand: record (k-block, plane, row r), my 8 units = one 16-byte half of it
        const unsigned xr = e_xr_lane + (par ? e_xr_par : 0u) + r * 32u + ((e_csw ^ ((r >> 3) & 1u)) << 4);
        __builtin_amdgcn_raw_buffer_store_b128(e_pv, rex, (on && pub_lane) ? xr : OOB, 0, 0);
      }
      if (part == 1 && (EM & 2)) {
        // (2) its transpose: row = my k, reduction index = r; plane gp
        const unsigned xc = __umul24(r >> 4, e_ckb) + e_clane + (par ? e_cpar : 0u) + ((((r >> 3) & 1u) ^ e_csw) << 4) + (r & 7u) * 2u;
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)e_w, rec, on ? xc : OOB, 0, 0);
      }
      if (part == 2 && (EM & 4)) {
        // (3) h_(t-1)^T of my cell: the forward cell pairs dz[t] with out[t - 1], the backward cell with out[t + 1]; the
        // frame that has no neighbour gets 0
        int q = dir ? to - 1 : to + 1;
        const bool edge = dir ? q < 0 : q >= T;
        q = edge ? (dir ? T - 1 : 0) : q;
        const unsigned kq = (unsigned)(ek0 + q);
        const unsigned hc = __umul24(kq >> 4, e_hkb) + e_hlane + ((((kq >> 3) & 1u) ^ e_hsw) << 4) + (kq & 7u) * 2u;
        __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(edge ? 0u : e_w), reh, on ? hc : OOB, 0, 0);
      }
    }
  };
  constexpr int EM_N = (EM & 1) + ((EM >> 1) & 1) + ((EM >> 2) & 1);
  // (EMIT: the companions' stores of two steps, issued behind the fetch; placement 1: the second step's come behind the claim)
  constexpr int XVM_AFTER = 2 * NKS + 14 + (MXH_EMIT_AT == 1 ? 1 : 2) * EM_N;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  mxf32x4 ax = {0.f, 0.f, 0.f, 0.f};
  auto x_product = [&](int s) {
    if constexpr (XIN) {
      asm volatile("" ::: "memory");
      const float *xs = xst + (s % 3) * 2048 + 512 * w + 4 * lane;
      const u32x4 bx0 = *reinterpret_cast<const u32x4 *>(xs), bx1 = *reinterpret_cast<const u32x4 *>(xs + 256);
      mxf32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      if (!(dbg & (2 | 512))) {
        a0 = MXH_MFMA(Wxp[1][0], bx0, a0);
        a1 = MXH_MFMA(Wxp[1][1], bx1, a1);
        a0 = MXH_MFMA(Wxp[0][0], bx0, a0);
        a1 = MXH_MFMA(Wxp[0][1], bx1, a1);
      }
      ax = a0 + a1;
    }
  };

  for (int s = 0; s < p.max_len; ++s) {
    MXH_STAMP(0, 0);
    mxf32x4 acc[4];
    unsigned long long t_fail = 0;
    int fails = 0;
    // (a) h_{s-1} as planes: the poll loop IS the operand fetch — the loads of the wave's k range (4 KiB of full lines)
    // are repeated until no word holds the sentinel
    u32x4 b1[NKS];
#pragma unroll
    for (int j = 0; j < NKS; ++j) b1[j] = zero4;
    if (s > 0 && !(dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes);
      bool first = true;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j) b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off1 + j * KSTEP_BYTES, 0, 16);
        if (first) {     // step s - 1's results, behind the loads
          result_stores();
          if (MXH_EMIT_AT == 1) { emit_part(0); emit_part(1); emit_part(2); }
          if constexpr (XIN) { wait_vm<XVM_AFTER>(); x_product(s); }
          first = false;
        }
#pragma unroll
        for (int j = 0; j < NKS; ++j) mx = mx_max4(mx, b1[j]);
        if (__all(mx != SENT)) break;
        // a failed round: bounded-spin bookkeeping (the clock is first read here)
        if (poll_round_failed(p, flag, lane, fails, t_fail, 1)) break;
      }
    } else {
      result_stores();
      if (MXH_EMIT_AT == 1) { emit_part(0); emit_part(1); emit_part(2); }
      wait_vm<0>();        // (no exchange loads to order the prefetch: s = 0, or the no-waiting experiment)
      x_product(s);
    }
    MXH_STAMP(0, 1);
    // (b) product: 4 column tiles (gate c) x NKS k-steps x {W_l.B, W_h.B}, small terms first.  Next step's
    // x-projection (HBM latency: as early as possible) is requested from inside the matrix stream.
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !(dbg & 2)) {
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MXH_MFMA_L(Wp[1][c][j], b1[j], acc[c]);
        if (j == 0) {
          fetch_x(XIN ? s + 2 : s + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (MXH_EMIT_AT == 2) { emit_part(0); emit_part(1); emit_part(2); }     // EMIT: interleaved with the matrix instructions below
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = MXH_MFMA(Wp[0][c][j], b1[j], acc[c]);
      }
      if constexpr (EMIT && MXH_EMIT_AT == 2) {
        // the scheduling region from the barrier above to here holds 8 NKS - 4 matrix instructions, the companions' address
        // arithmetic (~25 vector instructions each) and their stores: three vector instructions in the shadow of every matrix
        // instruction (16 cycles of the matrix pipe each), a store after every eighth
#pragma unroll
        for (int i = 0; i < 8 * NKS - 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
          if (i % 8 == 7) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
      }
    } else {
      fetch_x(XIN ? s + 2 : s + 1);
      if (MXH_EMIT_AT == 2) { emit_part(0); emit_part(1); emit_part(2); }
    }
    if constexpr (XIN) {      // the input's part joins gate w's sums (same scales: see the weights above)
      if (wu == 0) acc[0] += ax;
      else if (wu == 1) acc[1] += ax;
      else if (wu == 2) acc[2] += ax;
      else acc[3] += ax;
    }
    MXH_STAMP(0, 2);
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
    MXH_STAMP(0, 3);

    // (c) gates of (row frow, unit u16): both lane halves compute the same; the four waves' partial sums are added
    // first, then descaled (exact: powers of two), then the x-projection
    mxf32x4 z;
    {
      // staged as [row r2][gate][unit quad][4 units]
      const float *xs = xst + (s & 1) * 1024 + 256 * w + r2 * 64 + (u16 >> 2) * 4 + (u16 & 3);
      const float *pr = pbuf + (size_t)frow * L::ROWF + u16 * 4;
      mxf32x4 sum = *reinterpret_cast<const mxf32x4 *>(pr);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) sum += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR * L::ROWF);
      if constexpr (XIN)
        z = (mxf32x4){bias_f[0] + sum.x * inv[0], bias_f[1] + sum.y * inv[1], bias_f[2] + sum.z * inv[2], bias_f[3] + sum.w * inv[3]};
      else
        z = (mxf32x4){xs[0] + sum.x * inv[0], xs[16] + sum.y * inv[1], xs[32] + sum.z * inv[2], xs[48] + sum.w * inv[3]};
    }
    const float gi = fast_sigmoid(z.x), gj = fast_tanh(z.y), gf = fast_sigmoid(z.z + 1.0f), go = fast_sigmoid(z.w);
    const bool act = s < n_f;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act) { c_state = c_new; h_state = h_new; }

    // (d) publish h_s as two planes (frozen rows republish): the pair words of a plane sit in the even lanes of an
    // 8-lane group; lane 8 g + pl collects the four words of plane pl -> ONE 16-byte store instruction per wave
    {
      const float hs = h_state * MXH_HSCALE;
      const unsigned w0 = mxh_cvt2(hs, 0.f) & 0xFFFFu;
      const float r = hs - (float)__builtin_bit_cast(mxh16x2, w0).x;
      const unsigned w1 = mxh_cvt2(r, 0.f) & 0xFFFFu;
      const unsigned pr0 = w0 | (mx_dppu<DPP_XOR1>(w0) << 16), pr1 = w1 | (mx_dppu<DPP_XOR1>(w1) << 16);   // even lanes: units u, u + 1
      const u32x4 v0 = {pr0, mx_dppu<0x102>(pr0), mx_dppu<0x104>(pr0), mx_dppu<0x106>(pr0)};   // lane 8 g
      const u32x4 v1 = {mx_dppu<0x111>(pr1), mx_dppu<0x101>(pr1), mx_dppu<0x103>(pr1), mx_dppu<0x105>(pr1)};   // 8 g + 1
      const u32x4 pv = ppl == 0 ? v0 : v1;
      xstore(pv, rs, (pub_lane && s + 1 < p.max_len) ? (unsigned)((s % RING) * slot_bytes) + pub_off : OOB, coloc);
      // hand back my pieces of h_{s-2} (ordering: lstm_persist.hip, forward (d))
      xstore(sent4, rs, (pub_lane && s >= 2) ? (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off : OOB, coloc);
      if constexpr (EMIT) {
        e_pv = act ? pv : zero4;
        e_w = act ? (gp ? w1 : w0) : 0u;
        e_any = true;
        e_to = act ? (dir ? n_f - 1 - s : s) : s;
        if (MXH_EMIT_AT == 3) { emit_part(0); emit_part(1); emit_part(2); }
      }
    }
    MXH_STAMP(0, 4);
    // (e) results of this step: stored at the top of the next one (see result_stores)
    {
      const int t_g = dir ? n_f - 1 - s : s;
      d_any = true; d_act = act; d_t = t_g; d_to = act ? t_g : s;
      d_g0 = gp ? gf : gi;
      d_g1 = gp ? go : gj;
      d_v = gp ? (act ? h_new : 0.f) : c_new;
    }
    MXH_STAMP(0, 5);
  }
  result_stores();
  if (MXH_EMIT_AT != 3) { emit_part(0); emit_part(1); emit_part(2); }
  clock_stamp(p, 0, 1);
}

// ===========================================================================
// XIN: the layer input as the forward kernel's operand.  Workspace: [0] g (float, a power of two >= max |x|), [16 .. 16 + D)
// the columns' largest magnitudes (bit patterns, zeroed by the caller's fill), +1024 bytes: planes [B][T][2][64 k] fp16 of
// x 2^14 / g, zero beyond D.  One scale for the whole tensor: a scale that differed between frames could not share the
// recurrent product's accumulators; what it costs is range at the bottom only — an element below 2^-17 g keeps an absolute
// error of 2^-40 g (the convention of every f16x3 operand), far below one rounding of the pre-activation it is added to.
__device__ __forceinline__ float mxh_x_range(const unsigned *cols, int D) {
  unsigned m = 0;
  for (int i = 0; i < D; ++i) m = max(m, cols[i]);
  unsigned e = (m >> 23) & 0xFFu;
  if ((m & 0x7FFFFFFFu) == 0) e = 126;                 // all zero: g = 1
  e = e < 15 ? 15 : (e > 252 ? 252 : e);
  return __builtin_bit_cast(float, (e + 1u) << 23);   // 2^(floor(log2 max) + 1) > max
}
__global__ __launch_bounds__(256) void mxh_xplanes_kernel(int frames, int D, const float *__restrict__ x, char *ws) {
  const unsigned *cols = reinterpret_cast<const unsigned *>(ws) + 16;
  const float g = mxh_x_range(cols, D);
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float *>(ws) = g;
  const float sc = MXH_HSCALE / g;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t f = i >> 3;
  const int k0 = 8 * (int)(i & 7);
  if (f >= (size_t)frames) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = k0 + e < D ? x[f * D + k0 + e] * sc : 0.f;
  u32x4 h, l;
  mxh_split8(v, h, l);
  char *dst = ws + 1024 + f * 256 + 2 * k0;
  *reinterpret_cast<u32x4 *>(dst) = h;
  *reinterpret_cast<u32x4 *>(dst + 128) = l;
}
size_t lstm_mxh_xws_bytes(int B, int T) { return 1024 + (size_t)B * T * 256; }
// x [B, T, D] -> ws (lstm_mxh_xws_bytes): three small launches (zero the maxima, measure, convert)
int lstm_mxh_prepare_x(int B, int T, int D, const float *x, void *ws, hipStream_t stream, const FillSeg *also) {
  if (D > 64 || D % 8) return fail(NABU_EUNSUP, "persistent LSTM (mxh): in-kernel input projection takes D <= 64, D %% 8 == 0 (lstm_persist_fuses_input)");
  const FillSeg seg[2] = {{ws, 256, 0u}, also ? *also : FillSeg{nullptr, 0, 0u}};    // (also: the exchange ring)
  if (int e = multi_fill(seg, 2, stream)) return e;
  if (int e = nabu_pk_amax(x, D, B * T, D, nullptr, static_cast<uint32_t *>(ws) + 16, stream)) return e;
  const size_t n = (size_t)B * T * 8;
  hipLaunchKernelGGL(mxh_xplanes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, B * T, D, x, static_cast<char *>(ws));
  NABU_LAUNCH_CHECK();
  return 0;
}

// ===========================================================================
// host side (called from lstm_persist.hip's run_chunk)
// NABU_PERSIST_MX=0: the exact-fp32 kernels of lstm_persist.hip for every shape (a per-call form of the same switch:
// nabu_blstm_desc.recurrent_precision = NABU_REC_F32, lstm_persist_set_exact)
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

// do the fp16-plane kernels (this file, lstm_persist_mxf.hip) take the shape?
bool lstm_mx_supported(int B, int H) {
  if (!mx_env() || lstm_persist_exact() || !mx_device_ok()) return false;
  return (H == 128 || H == 256 || H == 512) && B >= 1;
}
int lstm_mx_chunk_rows() { return MXR * MXNU / 2; }   // 32 batch rows per launch (8 per unit)

size_t lstm_mxh_ring_bytes(bool fwd, int H) {
  const size_t P = H / UC;
  return fwd ? (size_t)MXNU * RING * 16 * H * 2 : (size_t)MXNU * MXHRINGB * P * P * MXR * UC * 4;
}

// one launch over B <= 32 rows; `a` comes filled from run_chunk (nshard = ceil(B / 8))
int lstm_mxh_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  const int grid = MXNU * (H / UC);
#define NABU_MXH_CASE(h)                                                                                            \
  case h:                                                                                                           \
    if (emit && !a.dbg && (a.emit.x_rows || a.emit.x_cols))                                                         \
      return a.xplanes ? mxh_launch(lstm_mxh_fwd_kernel<h, false, true, 7>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry)  \
                       : mxh_launch(lstm_mxh_fwd_kernel<h, false, false, 7>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry); \
    if (emit && !a.dbg)                                                                                             \
      return a.xplanes ? mxh_launch(lstm_mxh_fwd_kernel<h, false, true, 4>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry)  \
                       : mxh_launch(lstm_mxh_fwd_kernel<h, false, false, 4>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry); \
    if (a.xplanes)                                                                                                  \
      return a.dbg ? mxh_launch(lstm_mxh_fwd_kernel<h, true, true>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry)  \
                   : mxh_launch(lstm_mxh_fwd_kernel<h, false, true>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry); \
    return a.dbg ? mxh_launch(lstm_mxh_fwd_kernel<h, true>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry) \
                 : mxh_launch(lstm_mxh_fwd_kernel<h, false>, a, grid, MxhFwdLds<h>::TOTAL * sizeof(float), stream, dry);
  const bool emit = a.emit.x_rows || a.emit.x_cols || a.emit.hT[0] || a.emit.hT[1];
  if (emit && (a.dbg || !fwd)) return fail(NABU_EINVAL, "persistent LSTM (mxh): packed companions are written by the plain forward kernel only");
  if (!fwd) return lstm_mxh_bwd_launch(H, a, stream, dry);      // lstm_persist_mxh_bwd.hip
  switch (H) {
    NABU_MXH_CASE(128)
    NABU_MXH_CASE(256)
    NABU_MXH_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM (mxh): unsupported H=%d", H);
}

}  // namespace nabu
