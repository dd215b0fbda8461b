// lstm_persist.hip — whole-sequence persistent recurrent kernels for one BLSTM
// layer (both directions in one launch), MI355X / gfx950.
//
// THIS FILE: the host side of every persistent recurrent kernel (geometry, workspace, validation, chunking: run /
// run_chunk at the end) and the exact-fp32 kernels of rounds 1-3 (v_mfma_f32_4x4x1, 4 or 8 rows per unit).
// DISPATCH (run_chunk), three kernel families since round 5:
//   1. lstm_persist_mxf.hip   fp16-plane product, 32 hidden units per workgroup: 33 .. 64 rows at H = 512 in one launch;
//   2. lstm_persist_mxh.hip   fp16-plane product, 16 hidden units per workgroup: launches of <= 32 rows, H in {128, 256, 512}
//                             (larger batches as consecutive launches) — the default on a whole MI355X;
//   3. this file              exact fp32: H = 64, devices with fewer than 256 CUs, NABU_PERSIST_MX=0, and per call
//                             nabu_blstm_desc.recurrent_precision = NABU_REC_F32 (bench.py's fp32_end_to_end leg).
// All share the exchange protocol described below (lstm_persist_dev.h); the bf16-plane kernels of round 4's first half and
// the 16-rows-per-unit kernels are parked under tools/experiments/variants/ (lstm_persist_mx.hip, lstm_persist_mx16.hip,
// lstm_persist_mxh16_fwd.inc).
//
// WHY: the recurrence is 2 x sum(T_l) strictly sequential steps per pass; one
// launch per timestep pays a kernel boundary (>=1.5 us) plus a cold re-read of
// W_h every step.  Here one launch covers the whole sequence and W_h never
// leaves the register file.
//
// DECOMPOSITION.  unit = (direction, shard of BS batch rows); a unit's P = H/16
// workgroups each own 16 hidden units (= 64 gate columns) of that direction, i.e.
// a [H x 64] slice of W_h (128 KiB at H=512) held in VGPRs for the whole sequence.
// Two geometries (template parameter BS):
//   BS = 4: 256-thread workgroups (one wave per SIMD, 128 weight registers per
//           lane), two workgroups per CU (cfg2: 16 units x 32 workgroups = 512);
//   BS = 8: 512-thread workgroups (two waves per SIMD, 64 weight registers per
//           lane), one per CU; used when the batch is too large for BS = 4.
// Block b belongs to unit b % NU, so with the observed round-robin block->XCD
// placement a unit lives on one XCD.
//
// PER-STEP EXCHANGE (the only inter-workgroup communication):
//   forward : all-gather of h_t      — every workgroup publishes its [16 x BS]
//             slice with 16-byte stores and reads the unit's whole [H x BS] vector;
//   backward: reduce-scatter of dh   — every workgroup publishes its partial
//             product [BS x H] cut into per-destination pieces and reads the P
//             pieces addressed to it, summing them in a fixed order.
//   THE DATA IS THE FLAG: exchange slots are pre-filled with the bit pattern
//   0xFFFFFFFF (a NaN no h or dh value can take); a consumer re-loads a slot
//   until no word holds the sentinel.  No flags, no fences, no atomics; every
//   32-bit word is individually valid or sentinel, so torn 16-byte stores are
//   harmless.  Slots form a ring (forward 4 deep, backward 2); a slot is reset to
//   the sentinel after it was consumed (forward: by its producer, backward: by
//   its single reader).  No timing assumption orders a reset against the
//   slot's next use: a workgroup publishes only after loads it issued behind
//   its reset stores have returned (vector-memory operations complete in
//   issue order); the backward kernel, whose ring is only 2 deep, drains the
//   resets and meets at a second, execution-only barrier before it publishes
//   (DESIGN.md section 5).
//   PLACEMENT-INDEPENDENT CORRECTNESS: consumers always use sc1 loads (bypass
//   the per-CU L1).  At kernel start the workgroups of a unit exchange their
//   XCC ids (through sc1 stores); only if ALL of them sit on one XCD do they
//   publish with plain stores (the data then stays in that XCD's L2 and never
//   touches the fabric); otherwise they publish with write-through sc1 stores.
//   A different placement therefore changes speed, never results.
//   Every spin is bounded by a wall-clock timeout; a timeout sets a status
//   word, makes every workgroup leave, and is reported to the host.
//
// MATH per workgroup and step: [BS x H] x [H x 64] in exact fp32 on the MATRIX pipe with
// v_mfma_f32_4x4x1_16b_f32: one instruction multiplies, for 16 blocks, a [4 x 1] column by a
// [1 x 4] row.  Forward: block = hidden unit, A = h[4 rows][k], B = W[k][4 gates of the unit];
// backward: block = 4 consecutive k, A = dz[4 rows][c], B = W[4 k][c].  Four batch rows fill
// the instruction exactly (the 16x16 / 32x32 shapes would waste 3/4 of it), the k reduction
// stays inside the accumulators, a lane ends with the 4 rows of one (unit, gate) resp. one k —
// the layout the gate phase resp. the exchange wants — and the VALU only does the gate math.
// (The first versions ran the product as 256 v_pk_fma_f32 per lane plus DPP reductions: at the
// same FLOP rate, but ~3x the VALU instructions, fighting the other workgroup of the CU.)
#include "lstm_persist.h"
#include "lstm_persist_dev.h"

namespace nabu {

// ===========================================================================
// forward.  KPL = k values per lane, BS = batch rows per unit, threads = 64*BS.
// One __syncthreads per step: wave w gathers and multiplies ITS k range only (h staged in a
// wave-private LDS region), the waves' partial sums meet in a double-buffered LDS tile.
template <int KPL, int BS>
struct FwdLds {
  static constexpr int NW = BS, RG = BS / 4;     // waves, groups of 4 batch rows
  static constexpr int KW = 4 * KPL;             // k values multiplied by one wave
  static constexpr int H = NW * KW;
  static constexpr int ROW = KW + 4;             // padded row of the staged h: [RG][4 rows][KW] per wave
  static constexpr int HS = 0;
  static constexpr int PART = HS + NW * RG * 4 * ROW;   // [2][NW][RG][64 lanes][4 rows]
  static constexpr int XST = PART + 2 * NW * RG * 256;  // [2][64*BS] prefetched x-projection ([3][64*BS]: the input rows, XK > 0)
  static constexpr int FLAG = XST + 3 * 64 * BS;
  static constexpr int TOTAL = FLAG + 4;
};

typedef float mf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row_shl_f(float v, const int n) {   // lane i <- lane i+n of its 16-lane row
  const int x = __builtin_bit_cast(int, v);
  int r;
  switch (n) {
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0xF, true); break;
    case 8: r = __builtin_amdgcn_update_dpp(0, x, 0x108, 0xF, 0xF, true); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x10C, 0xF, 0xF, true); break;
  }
  return __builtin_bit_cast(float, r);
}

// XK > 0 (narrow input, D = 4 XK = 40; BS = 4): the input projection x_t . Wx + b is part of the step — XK more
// instructions per wave on top of the recurrent product's KW, Wx's slice in XK more registers — instead of a GEMM that
// writes [B T, 4H] per direction (524 MB at cfg2's first layer) for this kernel to read back.  x_t of the four rows
// (4 D floats) is fetched two steps ahead into a ring of three LDS buffers: the wave that issued a piece waits for it
// at the end of the step, the step's barrier orders it before the product two steps later.
template <int KPL, int BS, int XK = 0>
__global__ __launch_bounds__(64 * BS) __attribute__((amdgpu_waves_per_eu(BS == 8 ? 4 : 2))) void lstm_persist_fwd_kernel(PersistArgs p) {
  static_assert(XK == 0 || (BS == 4 && 16 * XK <= 64 * BS), "in-kernel input projection: 4 rows per unit, one prefetch lane per input element");
  using L = FwdLds<KPL, BS>;
  constexpr int PT = 64 * BS, NW = BS;   // threads, waves
  constexpr int H = L::H;
  constexpr int P = H / UC;
  constexpr int WP = H / 4;                        // 16-byte pieces of h gathered by one wave
  constexpr int NQ = (WP + 63) / 64;               // ... per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *hs = smem + L::HS, *part = smem + L::PART, *xst = smem + L::XST;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  block_identity(NU, &unit, &slot, (p.dbg & 8192) != 0);
  // Experiments kept behind NABU_PERSIST_DEBUG: 8192 = the two workgroups of a CU belong to different
  // units (rotation in block_identity) and 2048 = one of them runs at raised wave priority.  With the
  // product on the matrix pipe neither beats the plain layout (same unit, lockstep) by more than
  // run-to-run noise, so both are off.
  const bool hi_prio = BS == 4 && unit < NU / 2 && (p.dbg & 2048) && (p.dbg & 8192);
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * BS;
  const int T = p.T;
  // matrix-phase identity (v_mfma_f32_4x4x1_16b_f32: 16 blocks of [4 rows] x [4 columns]): lane =
  // (hidden unit mu = block, gate mg = column); the A operand of a lane is h[row = lane & 3][k]
  const int mu = lane >> 2, mg = lane & 3;
  // gate-phase identity: (gate, batch row, hidden unit); 4 adjacent lanes = 4 gates.  Loads and
  // stores of the per-step tensors use it directly (16-byte runs per gate and row).
  const int gg = tid & 3, gb = (tid >> 2) & (BS - 1), gu = tid / (4 * BS);
  const int gbg = b0 + gb;
  const int n_g = gbg < p.B ? p.len[gbg] : 0;

  // this lane's slice of W_h stays in registers for the whole sequence: column (gate mg, unit mu),
  // the KW rows k of this wave
  constexpr int KW = L::KW, RG = L::RG;
  constexpr int NACC = (BS == 4 && NABU_FWD_NACC == 4) ? 4 : 2;   // independent accumulator chains of the recurrent product
  float Wr[KW];
  {
    const float *Wh = p.kernel[dir] + ((size_t)p.D + (size_t)w * KW) * 4 * H + (size_t)mg * H + U0 + mu;
#pragma unroll
    for (int j = 0; j < KW; ++j) Wr[j] = Wh[(size_t)j * 4 * H];
  }
  float Wxr[XK > 0 ? XK : 1];      // (XK > 0) rows w*XK .. of Wx, my column
  float bias_r = 0.f;              // ... bias of my gate-phase column
  if (XK > 0) {
    const float *Wx = p.kernel[dir] + (size_t)w * XK * 4 * H + (size_t)mg * H + U0 + mu;
#pragma unroll
    for (int j = 0; j < XK; ++j) Wxr[j] = Wx[(size_t)j * 4 * H];
    bias_r = p.bias[dir][(size_t)gg * H + U0 + gu];
  }
  float c_state = 0.f, h_state = 0.f;
  if (!unit_handshake(p, unit, slot, NU, P, flag)) return;
  const bool coloc = flag[1] != 0;
  // Two workgroups of different units share a CU (BS = 4): start the upper half of the
  // units half a step late so that one multiplies while the other waits for its exchange.
  if (BS == 4 && unit >= NU / 2 && (p.dbg & 8192) && !(p.dbg & 32)) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 100) __builtin_amdgcn_s_sleep(4);
  }

  const size_t slot_bytes = (size_t)H * BS * 4;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  const bool pub_lane = (gg == 0) && ((gb & 3) == 0);
  const unsigned pub_off = (unsigned)(((U0 + gu) * BS + gb) * 4);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // x-projection of step s (GEMM output, bias included), fetched one step ahead so that its
  // HBM latency never sits in front of the exchange loads (vector loads return in order)
  // (Branch-free: a buffer load with an out-of-range offset returns 0, which is the value an
  // inactive row needs.  Straight-line code lets hipcc count outstanding operations exactly
  // instead of draining the whole queue at control-flow joins.)
  float *const gbase = p.gates[dir] + (size_t)gbg * T * 4 * H + (size_t)gg * H + U0 + gu;
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gbg * T * 4 * H + (size_t)gg * H + U0 + gu) * 4);
  auto fetch_x = [&](int s) {   // -> xst[s & 1][tid]
    const int t = dir ? n_g - 1 - s : s;
    const unsigned off = (s < n_g && !(p.dbg & 64)) ? goff + (unsigned)t * (unsigned)(16 * H) : OOB;
    prefetch_lds_b32(rg, off, smem, xst + (s & 1) * PT + 64 * w);
  };
  // (XK > 0) input rows of step s: lane L = 64 w + lane < 4 D is element (row L / D, k L % D); buffer s % 3
  constexpr int XD = 4 * XK;
  const i32x4 rx = raw_rsrc(p.x, XK > 0 ? (unsigned)((size_t)p.B * T * XD * 4) : 0u);
  const int xr_row = XK > 0 ? tid / (XK > 0 ? XD : 1) : 0, xr_k = XK > 0 ? tid % (XK > 0 ? XD : 1) : 0;
  const int xr_b = b0 + xr_row;
  const int n_x = (XK > 0 && tid < 4 * XD && xr_b < p.B) ? p.len[xr_b] : 0;
  auto fetch_xt = [&](int s) {
    const int t = dir ? n_x - 1 - s : s;
    const unsigned off = s < n_x ? (unsigned)((((size_t)xr_b * T + t) * XD + xr_k) * 4) : OOB;
    prefetch_lds_b32(rx, off, smem, xst + (s % 3) * PT + 64 * w);
  };
  if (XK > 0) {
    fetch_xt(0);
    fetch_xt(1);
  } else {
    fetch_x(0);
  }
  // all prologue loads (W_h, the first prefetch) are complete before the loop
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to the compiler's bookkeeping
  if (XK > 0) __syncthreads();          // the input rows were fetched by other waves than their readers
  float xnext = XK > 0 ? 0.f : xst[tid];

  if (hi_prio) __builtin_amdgcn_s_setprio(2);   // for the whole sequence (a change inside the loop body
                                                // split its basic blocks and cost the FMA loop its registers)
  for (int s = 0; s < p.max_len; ++s) {
    NABU_STAMP(0, 0);
    // (a) wait for h_{s-1}: wave w gathers the k range it multiplies, nothing else
    if (s > 0 && !(p.dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes) + (unsigned)(w * WP) * 16u;
      u32x4 v[NQ] = {};
      SpinGuard guard;
      guard.start();
      for (;;) {
        // every lane always loads a valid piece (surplus lanes re-read the last one): values
        // that are only conditionally defined across this loop were miscompiled by hipcc 7.2
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int qr = lane + i * 64, q = min(qr, WP - 1);
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)q * 16u, 0, 16);
          ok = ok && (qr >= WP || !has_sentinel(v[i]));
        }
        if (__all(ok)) break;
        if (guard.expired(p)) {
          if (lane == 0) {
            flag[0] = 1;
            __hip_atomic_store(p.status, 1 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
      // stage transposed: piece qr = (k, row group) -> [row group][row][k], k contiguous, so that the
      // A operands of 4 consecutive k are one 16-byte read
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int qr = lane + i * 64;
        if (qr < WP) {
          const f32x4 fv = __builtin_bit_cast(f32x4, v[i]);
          float *d = hs + ((size_t)(w * RG + qr % RG) * 4) * L::ROW + qr / RG;
          d[0] = fv.x; d[L::ROW] = fv.y; d[2 * L::ROW] = fv.z; d[3 * L::ROW] = fv.w;
        }
      }
    }
    NABU_STAMP(0, 1);
    if ((p.dbg & 4096) && (unit == 0 || unit == NU / 2) && tid == 0 && s == p.max_len / 2 + 1)
      p.status[384 + 64 * (unit != 0) + 2 * slot + 1] = (int)wall_clock64();   // poll done (wave 0)
    const float xg = XK > 0 ? bias_r : xnext;
    if (XK > 0) fetch_xt(s + 2);
    else fetch_x(s + 1);

    // (b) recurrent product on the matrix pipe, exact fp32: v_mfma_f32_4x4x1_16b_f32 multiplies, for
    // each of 16 hidden units, [4 rows x 1] (h) by [1 x 4 gates] (W): no padding waste at 4 batch
    // rows, the k reduction stays inside the accumulators (no cross-lane sum), and the VALU is free
    // for the other workgroup of the CU.  Two accumulators per row group hide the dependent-issue
    // latency.  The staged h is read back by the wave that wrote it (in-order LDS, no barrier).
    mf32x4 acc[RG][NACC];
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[g][a] = (mf32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !(p.dbg & 2)) {
      const float *hrow = hs + ((size_t)w * RG * 4 + (lane & 3)) * L::ROW;
      constexpr int NCH = KW / 4;
      float4 hq[2][RG];
#pragma unroll
      for (int g = 0; g < RG; ++g) hq[0][g] = *reinterpret_cast<const float4 *>(hrow + (size_t)g * 4 * L::ROW);
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        if ((p.dbg & 256) && ch >= NCH * 3 / 4) break;      // timing experiment: three quarters of the product
        if (ch + 1 < NCH) {
#pragma unroll
          for (int g = 0; g < RG; ++g)
            hq[(ch + 1) & 1][g] = *reinterpret_cast<const float4 *>(hrow + (size_t)g * 4 * L::ROW + 4 * (ch + 1));
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const float4 h4 = hq[ch & 1][g];
          acc[g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.x, Wr[4 * ch + 0], acc[g][0], 0, 0, 0);
          acc[g][1 % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.y, Wr[4 * ch + 1], acc[g][1 % NACC], 0, 0, 0);
          acc[g][2 % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.z, Wr[4 * ch + 2], acc[g][2 % NACC], 0, 0, 0);
          acc[g][3 % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(h4.w, Wr[4 * ch + 3], acc[g][3 % NACC], 0, 0, 0);
        }
      }
    }
    if (XK > 0) {
      // + x_s . Wx: my XK input indices, row lane & 3 (the staged rows are [row][D])
      const float *xrow = xst + (s % 3) * PT + (lane & 3) * XD + w * XK;
#pragma unroll
      for (int j = 0; j < XK; ++j) acc[0][j % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(xrow[j], Wxr[j], acc[0][j % NACC], 0, 0, 0);
    }
    NABU_STAMP(0, 2);
    // hand the wave's partial sums to the gate phase: [wave][row group][lane][4 rows]
    float *const pbuf = part + (s & 1) * (NW * RG * 256);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      mf32x4 t = acc[g][0] + acc[g][1];
      if (NACC == 4) t += acc[g][2] + acc[g][3];
      *reinterpret_cast<mf32x4 *>(pbuf + ((size_t)(w * RG + g) * 64 + lane) * 4) = t;
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    NABU_STAMP(0, 3);

    // (c) gates: thread = (gate gg, row gb, unit gu); its partials sit at lane (gu, gg), row gb of
    // every wave's tile
    float z = xg;
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) z += pbuf[((size_t)(ww * RG + (gb >> 2)) * 64 + gu * 4 + gg) * 4 + (gb & 3)];
    const float a = (gg == 1) ? fast_tanh(z) : fast_sigmoid(gg == 2 ? z + 1.0f : z);
    const float gi = QUAD_BCAST(a, 0), gj = QUAD_BCAST(a, 1), gf = QUAD_BCAST(a, 2), go = QUAD_BCAST(a, 3);
    const bool act_g = s < n_g;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act_g) { c_state = c_new; h_state = h_new; }

    // (d) publish h_s (frozen rows republish their state): 4 rows -> one 16-byte store.
    // Ordering of my slot reset (issued two steps ago) before this store: same lane, same address.
    // Ordering of my slot reset of the PREVIOUS step (h_{s-3}) before this publish — consumers poll that
    // slot for h_{s+1} once they have seen h_s and must not find h_{s-3} there: vector-memory operations
    // complete in issue order and this step's exchange loads, issued after that reset, have been consumed.
    {
      const float h1 = row_shl_f(h_state, 4), h2 = row_shl_f(h_state, 8), h3 = row_shl_f(h_state, 12);
      u32x4 pv;
      pv.x = __builtin_bit_cast(unsigned, h_state);
      pv.y = __builtin_bit_cast(unsigned, h1);
      pv.z = __builtin_bit_cast(unsigned, h2);
      pv.w = __builtin_bit_cast(unsigned, h3);
      // (stores of the other lanes go out of range and are dropped by the buffer bounds check)
      xstore(pv, rs, (pub_lane && s + 1 < p.max_len) ? (unsigned)((s % RING) * slot_bytes) + pub_off : OOB, coloc);
      // hand back my piece of h_{s-2}: every wave of this workgroup has seen h_{s-1} of every
      // producer (barrier above), and a producer publishes h_{s-1} only after all its waves
      // finished reading h_{s-2}
      xstore(sent4, rs, (pub_lane && s >= 2) ? (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off : OOB, coloc);
    }
    // claim the prefetched x-projection: 2 = the publish and the reset store above.  (Here, in
    // front of the result stores: a later wait would also wait for those.)
    wait_vm<2>();
    if (XK == 0) xnext = xst[((s + 1) & 1) * PT + tid];
    NABU_STAMP(0, 4);
    if ((p.dbg & 4096) && (unit == 0 || unit == NU / 2) && tid == 0 && s == p.max_len / 2)
      p.status[384 + 64 * (unit != 0) + 2 * slot] = (int)wall_clock64();

    // (e) off the critical path: activations (in place over the x-projection), cell state, output
    if (gbg < p.B && !(p.dbg & 128)) {
      const int t_g = dir ? n_g - 1 - s : s;
      if (act_g) {
        gbase[(size_t)t_g * 4 * H] = a;
        if (gg == 0) p.cs[dir][((size_t)gbg * T + t_g) * H + U0 + gu] = c_new;
      }
      if (gg == 1)
        p.out[((size_t)gbg * T + (act_g ? t_g : s)) * 2 * H + (size_t)dir * H + U0 + gu] = act_g ? h_new : 0.f;
    }
    NABU_STAMP(0, 5);
  }
}

// ===========================================================================
// backward.  Lanes are (gate quarter cq, k-quad kq); NKQ k-quads per lane.  One __syncthreads
// per step: wave w reduces exactly the pieces its own gate threads need (wave-private LDS
// transpose), dz meets in a double-buffered LDS tile.
template <int BS>
struct BwdLds {
  static constexpr int RG = BS / 4;                // groups of 4 batch rows
  static constexpr int DROW = 64 + 4;              // padded row of dz: [RG][4 rows][64 gate columns]
  static constexpr int DZ = 0;                     // double buffered
  static constexpr int RED = DZ + 2 * RG * 4 * DROW;   // [16 groups][16*BS] partial dh sums
  static constexpr int XST = RED + 16 * 16 * BS;   // [2][3][64*BS] prefetched saved values
  static constexpr int FLAG = XST + 2 * 3 * 64 * BS;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H, int BS>
__global__ __launch_bounds__(64 * BS) __attribute__((amdgpu_waves_per_eu(BS == 8 ? 4 : 2))) void lstm_persist_bwd_kernel(PersistArgs p) {
  using L = BwdLds<BS>;
  constexpr int PT = 64 * BS;
  constexpr int P = H / UC;
  constexpr int NW = BS, RG = L::RG;
  constexpr int NG = H / 64;                        // output groups of 64 k (one MFMA covers 16 x 4 k)
  constexpr int GPW = NG >= NW ? NG / NW : 1;       // groups per wave (waves >= NG idle in the product)
  constexpr int PPB = UC * BS / 4;                  // 16-byte pieces per source piece
  constexpr int NQ = (P + 15) / 16;                 // sources per lane (16 source groups per wave)
  static_assert(GPW * 64 <= 128, "backward kernel holds at most 128 weight registers per lane");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *dzs = smem + L::DZ, *red = smem + L::RED, *xst = smem + L::XST;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  block_identity(NU, &unit, &slot, (p.dbg & 8192) != 0);
  // The two workgroups of a CU (units u and u + NU/2) must not multiply at the same time: the product
  // of one alone takes 0.5 us, two at once take 1 us each, and the slowest workgroup sets the step of
  // its whole unit.  The lower unit's product runs at raised wave priority: it is never slowed, the
  // other one is pushed back whenever they overlap and so slides into the gaps (self-stabilising).
  const bool hi_prio = BS == 4 && unit < NU / 2 && (p.dbg & 2048) && (p.dbg & 8192);
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * UC, b0 = shard * BS;
  const int T = p.T;
  // matrix-phase identity (v_mfma_f32_4x4x1_16b_f32): lane = one output k of a 64-k group; its A
  // operand is dz[row = lane & 3][column c]
  const bool mfma_wave = w * GPW < NG;
  // gate-phase identity as in the forward kernel; also used for every per-step load and store
  const int gg = tid & 3, gb = (tid >> 2) & (BS - 1), gu = tid / (4 * BS);
  const int gbg = b0 + gb;
  const int n_g = gbg < p.B ? p.len[gbg] : 0;
  // exchange identity: wave w sums, over all sources, the 4 pieces (pos) its own gate threads
  // consume; lane = (source group, piece)
  const int xgrp = lane >> 2, xpos = 4 * w + (lane & 3);

  // Wr[m][c] = W_h[k][column c of my workgroup], k = (w*GPW + m)*64 + lane, c = gate*16 + unit
  float Wr[GPW][64];
#pragma unroll
  for (int m = 0; m < GPW; ++m) {
    const int k = min((w * GPW + m) * 64 + lane, H - 1);
    const float *Wh = p.kernel[dir] + ((size_t)p.D + k) * 4 * H + U0;
#pragma unroll
    for (int c = 0; c < 64; ++c) Wr[m][c] = Wh[(size_t)(c >> 4) * H + (c & 15)];
  }
  float dc_state = 0.f;
  float db_acc = 0.f;   // bias gradient: my (gate, row, unit) dz summed over the sequence
  float am_acc = 0.f;   // ... and its largest magnitude (the column scales of the f16x3 weight-gradient products)
  if (!unit_handshake(p, unit, slot, NU, P, flag)) return;
  const bool coloc = flag[1] != 0;
  if (BS == 4 && unit >= NU / 2 && (p.dbg & 8192) && !(p.dbg & 32)) {   // see the forward kernel
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 100) __builtin_amdgcn_s_sleep(4);
  }

  // ring slot = [dest P][src P][16 u][BS b] floats
  const size_t piece_bytes = (size_t)UC * BS * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;      // what one destination reads
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RINGB * slot_bytes, 0, (int)(RINGB * slot_bytes), 0x00020000);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // saved forward values of step s, fetched one step ahead: av = my gate's activation,
  // xv = c (gate lane 0) / c_prev (lane 1) / dout (lane 2), shared across the quad below
  // (Branch-free as in the forward kernel: out-of-range buffer loads return 0.)
  float *const gbase = p.gates[dir] + (size_t)gbg * T * 4 * H + (size_t)gg * H + U0 + gu;
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  const unsigned goff = (unsigned)(((size_t)gbg * T * 4 * H + (size_t)gg * H + U0 + gu) * 4);
  const unsigned coff = (unsigned)(((size_t)gbg * T * H + U0 + gu) * 4);
  const unsigned doff = (unsigned)(((size_t)gbg * T * 2 * H + (size_t)dir * H + U0 + gu) * 4);
  // av = my gate's activation; xc = c (gate lane 0) / c_prev (lane 1), xd = dout (lane 2), else 0
  auto fetch = [&](int s) {   // -> xst[s & 1][0..2][tid]; only called with s >= 0
    const bool act = s >= 0 && s < n_g && !(p.dbg & 64);
    const int t = dir ? n_g - 1 - s : s;
    const int tc = gg == 0 ? t : (dir ? t + 1 : t - 1);     // c of this step / of the previous one
    const bool want_c = act && (gg == 0 || (gg == 1 && s > 0));
    float *st = xst + (s & 1) * 3 * PT + 64 * w;
    prefetch_lds_b32(rg, act ? goff + (unsigned)t * (unsigned)(16 * H) : OOB, smem, st);
    prefetch_lds_b32(rc, want_c ? coff + (unsigned)tc * (unsigned)(4 * H) : OOB, smem, st + PT);
    prefetch_lds_b32(rd, (act && gg == 2) ? doff + (unsigned)t * (unsigned)(8 * H) : OOB, smem, st + 2 * PT);
  };
  auto fetched = [&](int s, float &av, float &xv) {
    const float *st = xst + (s & 1) * 3 * PT + tid;
    av = st[0];
    xv = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, st[PT]) | __builtin_bit_cast(unsigned, st[2 * PT]));
  };
  float av_next, xv_next;
  fetch(p.max_len - 1);
  wait_vm<0>();
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): see the forward kernel
  fetched(p.max_len - 1, av_next, xv_next);

  if (hi_prio) __builtin_amdgcn_s_setprio(2);
  // where my partial products of step s go: piece (dest, me) of slot s % RINGB; OOB = nothing to publish
  auto out_off = [&](int s, int m) -> unsigned {
    const int k = (w * GPW + m) * 64 + lane;
    const int dest = k / UC, ul = k % UC;
    return (s > 0 && mfma_wave && k < H)
               ? (unsigned)((s % RINGB) * slot_bytes + (size_t)dest * block_bytes + (size_t)slot * piece_bytes +
                            (size_t)ul * BS * 4)
               : OOB;
  };
  for (int s = p.max_len - 1; s >= 0; --s) {
    NABU_STAMP(1, 0);
    // (a) reduce-scatter input: the partial products of step s+1 addressed to me
    float4 psum = make_float4(0.f, 0.f, 0.f, 0.f);
    u32x4 v[NQ] = {};
    const unsigned base = (unsigned)(((s + 1) % RINGB) * slot_bytes + (size_t)slot * block_bytes);
    const bool have_in = s + 1 < p.max_len && !(p.dbg & 1);
    if (have_in) {
      SpinGuard guard;
      guard.start();
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
          const int sr = xgrp + 16 * i, src = min(sr, P - 1);   // see the forward kernel
          v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + (unsigned)(src * PPB + xpos) * 16u, 0, 16);
          // a surplus lane's piece may already have been handed back by its owner: ignore it
          ok = ok && (sr >= P || !has_sentinel(v[i]));
        }
        if (__all(ok)) break;
        if (guard.expired(p)) {
          if (lane == 0) {
            flag[0] = 1;
            __hip_atomic_store(p.status, 2 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          break;
        }
      }
    }
    NABU_STAMP(1, 1);
    if (have_in) {
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int src = xgrp + 16 * i;
        if (src < P) {
          const f32x4 fv = __builtin_bit_cast(f32x4, v[i]);
          psum.x += fv.x;
          psum.y += fv.y;
          psum.z += fv.z;
          psum.w += fv.w;
        }
      }
    }
    // wave-private transpose: group xgrp holds the sum over the sources {xgrp + 16 i}
    *reinterpret_cast<float4 *>(red + xgrp * (UC * BS) + xpos * 4) = psum;
    const float a = av_next, xv = xv_next;

    // (b) gate gradients: thread = (gate gg, row gb, unit gu)
    float dh = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) dh += red[(gg + 4 * m) * (UC * BS) + gu * BS + gb];
    dh += QUAD_XOR1(dh);
    dh += QUAD_XOR2(dh);
    const float gi = QUAD_BCAST(a, 0), gj = QUAD_BCAST(a, 1), gf = QUAD_BCAST(a, 2), go = QUAD_BCAST(a, 3);
    const float c = QUAD_BCAST(xv, 0), cprev = QUAD_BCAST(xv, 1), dout = QUAD_BCAST(xv, 2);
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
    db_acc += dz;
    am_acc = fmaxf(am_acc, fabsf(dz));
    float *const dzb = dzs + (s & 1) * (RG * 4 * L::DROW);
    dzb[(size_t)gb * L::DROW + gg * 16 + gu] = dz;      // [row group][row][gate*16 + unit] = [gb][c]
    NABU_STAMP(1, 2);
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;
    NABU_STAMP(1, 3);
    // next step's saved values travel while this one multiplies
    // I am the only reader of my pieces: hand the slot back
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int src = xgrp + 16 * i;
      xstore(sent4, rs, (have_in && src < P) ? base + (unsigned)(src * PPB + xpos) * 16u : OOB, coloc);
    }
    if (s > 0) fetch(s - 1);

    // (c) partial product for step s-1 on the matrix pipe (exact fp32, see the forward kernel):
    // block = 4 consecutive k, D[row][k] += dz[row][c] * W[k][c] over my 64 gate columns c.  A lane
    // ends with the 4 rows of ITS k: exactly one 16-byte piece of the reduce-scatter — no cross-lane sum.
    if (s > 0) {
      mf32x4 acc[GPW][RG][2];
#pragma unroll
      for (int m = 0; m < GPW; ++m)
#pragma unroll
        for (int g = 0; g < RG; ++g) acc[m][g][0] = acc[m][g][1] = (mf32x4){0.f, 0.f, 0.f, 0.f};
      if (!(p.dbg & 2) && mfma_wave) {
        const float *drow = dzb + (size_t)(lane & 3) * L::DROW;
        float4 dq[2][RG];
#pragma unroll
        for (int g = 0; g < RG; ++g) dq[0][g] = *reinterpret_cast<const float4 *>(drow + (size_t)g * 4 * L::DROW);
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
          if (ch + 1 < 16) {
#pragma unroll
            for (int g = 0; g < RG; ++g)
              dq[(ch + 1) & 1][g] = *reinterpret_cast<const float4 *>(drow + (size_t)g * 4 * L::DROW + 4 * (ch + 1));
          }
#pragma unroll
          for (int g = 0; g < RG; ++g) {
            const float4 d4 = dq[ch & 1][g];
#pragma unroll
            for (int m = 0; m < GPW; ++m) {
              acc[m][g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(d4.x, Wr[m][4 * ch + 0], acc[m][g][0], 0, 0, 0);
              acc[m][g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(d4.y, Wr[m][4 * ch + 1], acc[m][g][1], 0, 0, 0);
              acc[m][g][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(d4.z, Wr[m][4 * ch + 2], acc[m][g][0], 0, 0, 0);
              acc[m][g][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(d4.w, Wr[m][4 * ch + 3], acc[m][g][1], 0, 0, 0);
            }
          }
        }
      }
      NABU_STAMP(1, 4);
      // publish.  The pieces I write into (slot s % RINGB) were handed back by their readers RINGB - 1
      // steps ago; what makes that reset PERFORMED before this store arrives:
      //   RINGB >= 3: I have polled the reader's publish of the step AFTER its reset, and every wave of the
      //     reader completed a poll loop in between (loads issued after the reset stores returned:
      //     vector-memory operations complete in issue order);
      //   RINGB == 2 (the ring of an XCD then is 1 MB and stays in its L2): I have only
      //     polled the publish of the reset's own step, issued by other waves than the resetting ones — so
      //     every wave drains its reset stores (all but the 3 prefetch loads issued after them) and the
      //     workgroup meets at an execution barrier before anybody publishes.
      if (RINGB == 2) {
        wait_vm<3>();
        NABU_STAMP(1, 7);
        __builtin_amdgcn_s_barrier();
        NABU_STAMP(1, 8);
      }
#pragma unroll
      for (int m = 0; m < GPW; ++m) {
        const unsigned off = out_off(s, m);
#pragma unroll
        for (int g = 0; g < RG; ++g) {
          const mf32x4 t = acc[m][g][0] + acc[m][g][1];
          xstore(__builtin_bit_cast(u32x4, t), rs, off == OOB ? OOB : off + 16 * g, coloc);
        }
      }
    }
    NABU_STAMP(1, 9);
    // claim the prefetched values here, in front of the dz store (see the forward kernel)
    if (s > 0) {
      wait_vm<GPW * RG>();   // N = the publish stores above
      fetched(s - 1, av_next, xv_next);
    }
    NABU_STAMP(1, 5);
    // (d) dz to HBM (in place over the activations); padded frames get 0
    if (gbg < p.B && !(p.dbg & 128)) {
      const int t_g = dir ? n_g - 1 - s : s;
      gbase[(size_t)(act_g ? t_g : s) * 4 * H] = dz;
    }
    NABU_STAMP(1, 6);
  }
  // bias gradient of my 64 gate columns, summed over my BS rows: one partial row per unit (the host
  // adds the shards with the deterministic column-sum kernel) — saves re-reading dz from HBM
  __syncthreads();
  red[tid] = db_acc;
  __syncthreads();
  if (gb == 0) {
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < BS; ++r) sum += red[gg + 4 * r + 4 * BS * gu];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)gg * H + U0 + gu] = sum;
  }
  __syncthreads();
  red[tid] = am_acc;
  __syncthreads();
  if (gb == 0) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < BS; ++r) m = fmaxf(m, red[gg + 4 * r + 4 * BS * gu]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)gg * H + U0 + gu] = m;
  }
}

// ===========================================================================
// bound of every in-kernel spin, wall_clock64 ticks (100 MHz); process-wide setting
static unsigned long long g_timeout_ticks = 20000000ull;   // 0.2 s: a step takes microseconds
unsigned long long lstm_persist_timeout_ticks() { return g_timeout_ticks; }
void lstm_persist_set_timeout_us(long long us) {
  g_timeout_ticks = us > 0 ? (unsigned long long)us * 100ull : 20000000ull;
}

// compute units of the CURRENT device (partitioned / CPX modes expose fewer than the 256 of a whole
// MI355X); every geometry decision below uses it, so that a shape whose grid cannot be co-resident
// is reported as unsupported — LSTM_AUTO then takes the step-wise kernels instead of failing at launch
static int cu_count() {
  static thread_local int cached_dev = -1, cached = NCU;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return NCU; }
  if (dev != cached_dev) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
      (void)hipGetLastError();
      v = NCU;
    }
    cached = v > NCU ? NCU : v;     // block -> CU bookkeeping (block_identity) is written for <= 256 CUs
    cached_dev = dev;
  }
  return cached;
}

// lstm_persist_mxh.hip: the fp16-plane kernels — three fp16 plane products of row-scaled operands, 8 rows per unit, launches
// of <= 32 rows (the default wherever H is 128, 256 or 512 on a whole MI355X; NABU_PERSIST_MX=0 or
// nabu_blstm_desc.recurrent_precision = NABU_REC_F32: the exact-fp32 kernels of this file)
bool lstm_mx_supported(int B, int H);
int lstm_mx_chunk_rows();
size_t lstm_mxh_ring_bytes(bool fwd, int H);
int lstm_mxh_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry);
size_t lstm_mxh_xws_bytes(int B, int T);
int lstm_mxh_prepare_x(int B, int T, int D, const float *x, void *ws, hipStream_t stream, const FillSeg *also);
// lstm_persist_mxf.hip: the same arithmetic with 32 hidden units per workgroup, two units of 8 rows per XCD: 33 .. 64 batch
// rows at H = 512 in one launch (NABU_PERSIST_MXF=0: launches of <= 32 rows on the kernels above)
bool lstm_mxf_supported(int B, int H);
size_t lstm_mxf_ring_bytes(bool fwd, int H);
int lstm_mxf_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry);

// geometry: BS = 4 (two 256-thread workgroups per CU) when the batch fits, else BS = 8
bool lstm_persist_fuses_input(int B, int T, int D, int H);
size_t lstm_persist_db_floats(int B, int H) { return (size_t)((B + 3) / 4) * 2 * 4 * H; }
static int pick_bs(int B, int H, bool fwd) {
  const int P = H / UC, ncu = cu_count();
  if (2 * ((B + 3) / 4) * P <= 2 * ncu) return 4;
  if (2 * ((B + 7) / 8) * P <= ncu) return 8;          // one 512-thread workgroup per CU
  // two 512-thread workgroups per CU (<= 128 VGPRs): cfg5's B = 64 at H = 512 in ONE launch.  Forward
  // only: measured 4.04 us per step against 2 x 2.41 for two launches of 32 rows; the backward kernel,
  // whose reduce-scatter volume doubles with the rows, takes 5.69 against 2 x 2.47 (5.02 against 2 x 2.41
  // with the 2-deep ring) and stays chunked.
  if (fwd && 2 * ((B + 7) / 8) * P <= 2 * ncu) return 8;
  return 0;
}

// Batches that need more workgroups than the chip holds run as consecutive launches over
// chunks of batch rows (the tensors are batch-major, a chunk is a contiguous slab): at H = 512 a
// launch takes up to 64 rows (BS = 8, two workgroups per CU), B = 96 is a launch of 64 and one of 32.
static int chunk_rows(int B, int H, bool fwd, int T = 0) {
  if (lstm_mx_supported(B, H)) {
    // up to 32 rows: one launch of 8-row units; 33 .. 64 rows at H = 512: one launch of lstm_persist_mxf.hip (unless a slab
    // of 64 rows x T frames of gates is beyond the 32-bit buffer offsets of one launch); otherwise launches of 32 rows
    int c = lstm_mx_chunk_rows();
    if (B > c && lstm_mxf_supported(B > 2 * c ? 2 * c : B, H) && !(T > 0 && (size_t)2 * c * T * 4 * H * 4 >= 0x80000000ull)) c *= 2;
    return B < c ? B : c;
  }
  if (pick_bs(B, H, fwd)) return B;
  int c = (fwd ? 8 : 4) * (2 * cu_count() / (2 * (H / UC)));   // largest batch of one launch
  return c < 4 ? 4 : c;
}

bool lstm_persist_supported(int B, int T, int H) {
  if (!(H == 64 || H == 128 || H == 256 || H == 512)) return false;
  if (B <= 0 || T <= 0) return false;
  if (lstm_mx_supported(B, H)) return (size_t)chunk_rows(B, H, true, T) * T * 4 * H * 4 < 0x80000000ull;
  if ((size_t)B * T * 4 * H * 4 >= 0x80000000ull && (size_t)chunk_rows(B, H, true) * T * 4 * H * 4 >= 0x80000000ull)
    return false;   // 32-bit buffer offsets inside one launch
  return pick_bs(chunk_rows(B, H, true), H, true) != 0 && pick_bs(chunk_rows(B, H, false), H, false) != 0;
}

// bias-gradient partials of the backward kernel: one row of 2 x 4H per shard (BS = 4 gives the most)
static size_t db_part_bytes(int B, int H) { return 2 * (size_t)((B + 3) / 4) * 2 * 4 * H * sizeof(float); }   // sums, maxima

static size_t ring_bytes(bool fwd, int BS, int nshard, int H) {
  const size_t NU = 2 * (size_t)nshard, P = H / UC;
  return fwd ? NU * RING * (size_t)H * BS * 4 : NU * RINGB * P * P * UC * BS * 4;
}

size_t lstm_persist_ws_bytes(int B, int T, int H) {
  if (!lstm_persist_supported(B, T, H)) return 0;
  size_t m = 0;
  for (int BS = 4; BS <= 8; BS += 4) {   // either geometry may be selected at run time
    for (int f = 0; f < 2; ++f) {
      const int Bc = chunk_rows(B, H, f != 0, T);
      const int ns = (Bc + BS - 1) / BS;
      const size_t r = ring_bytes(f != 0, BS, ns, H);
      if (r > m) m = r;
    }
  }
  if (lstm_mx_supported(B, H))
    for (int f = 0; f < 2; ++f) {
      if (lstm_mxh_ring_bytes(f != 0, H) > m) m = lstm_mxh_ring_bytes(f != 0, H);
      // (a batch of more than 64 rows runs as launches of 64 rows and a remainder)
      if (B > lstm_mx_chunk_rows() && lstm_mxf_supported(B > 64 ? 64 : B, H) && lstm_mxf_ring_bytes(f != 0, H) > m)
        m = lstm_mxf_ring_bytes(f != 0, H);
    }
  return TABLE_BYTES + m + db_part_bytes(B, H);
}

// Co-residency is a REQUIREMENT of these kernels (every workgroup of a unit polls its peers): the grid is
// validated against the runtime's occupancy answer for this kernel, block size and LDS request on the current
// device before it is launched — what hipLaunchCooperativeKernel would check, without its +15-19 us per launch
// (MI355X_MICROARCH.md, coop-launch row).  The answer can be one block per CU too high at 81-112 SGPRs with
// 256-thread blocks (same guide); the grids here need at most 2 per CU where the register file admits 2 and the
// LDS request is sized for exactly that, and every spin is bounded, so an optimistic answer costs a time-out,
// never a hang.  NABU_EUNSUP makes LSTM_AUTO take the step-wise kernels.
// dry: validate only (every chunk of a call is validated BEFORE the first one is enqueued: a later chunk that cannot be
// co-resident must not leave the earlier chunks' in-place updates of the gate buffers behind — the step-wise fallback
// of LSTM_AUTO restarts from the untouched buffers)
template <typename K>
static int launch(K kernel, const PersistArgs &a, int grid, int threads, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, threads, blocks; size_t lds; };
  static thread_local Seen seen[32] = {};
  int dev = 0;
  NABU_HIP(hipGetDevice(&dev));
  int blocks = -1;
  for (const Seen &c : seen)
    if (c.fn == fn && c.dev == dev && c.threads == threads && c.lds == lds) blocks = c.blocks;
  if (blocks < 0) {
    NABU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    NABU_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, fn, threads, lds));
    for (Seen &c : seen)
      if (!c.fn) { c = Seen{fn, dev, threads, blocks, lds}; break; }
  }
  if ((long long)blocks * cu_count() < grid)
    return fail(NABU_EUNSUP, "persistent LSTM: %d workgroups cannot be co-resident (%d per CU x %d CUs on this device)",
                grid, blocks, cu_count());
  if (dry) return 0;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

struct RowMax { unsigned *part; unsigned stride; bool kept; };   // (lstm_persist.h: rowmax) of the chunk at hand
static int run_chunk(bool fwd, int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2], float *out,
                     const float *dout, int *status, void *ws, size_t ws_bytes, float *db_part, float *amax_part,
                     int *shard_base, hipStream_t stream, const float *x, const float *const bias[2], bool dry, RowMax *rm,
                     const void *xws, int xrow0, const EmitArgs *emit);

// exchange ring + XCC table back to 0xFF bytes: one small kernel (a hipMemsetAsync is its own kind of dispatch and
// costs ~6 us of queue gap in front of every recurrent launch)
// The caller of the forward pass may clear the ring in a fill of its own (it has one anyway: lstm.hip, the maxima of
// the input projection): lstm_persist_ring_seg names the region the launch will want cleared, and
// lstm_persist_ring_cleared says that it has been.  The note is consumed by the next reset on this host thread, whatever
// that launch is, and honoured only if it names the same workspace, stream and at least as many bytes.
static thread_local struct { void *ws; size_t bytes; hipStream_t stream; } g_ring_cleared = {nullptr, 0, nullptr};
static int ring_reset(void *ws, size_t bytes, hipStream_t stream) {
  const bool cleared = g_ring_cleared.ws == ws && g_ring_cleared.stream == stream && g_ring_cleared.bytes >= bytes;
  g_ring_cleared.ws = nullptr;
  if (cleared) return 0;
  const FillSeg seg = {ws, (bytes + 3) / 4, 0xFFFFFFFFu};
  return multi_fill(&seg, 1, stream);
}
bool lstm_persist_ring_seg(bool fwd, int B, int T, int H, void *ws, FillSeg *seg) {
  if (!ws || !lstm_mx_supported(B, H)) return false;      // (the fp32 kernels' rings depend on the chunking: not offered)
  if (chunk_rows(B, H, fwd, T) < B) return false;          // several launches share the ring: each clears it
  size_t bytes;
  if (lstm_mxf_supported(B, H)) bytes = TABLE_BYTES + lstm_mxf_ring_bytes(fwd, H);
  else if (B <= lstm_mx_chunk_rows()) bytes = TABLE_BYTES + lstm_mxh_ring_bytes(fwd, H);
  else return false;
  *seg = FillSeg{ws, (bytes + 3) / 4, 0xFFFFFFFFu};
  return true;
}
void lstm_persist_ring_cleared(const FillSeg *seg, hipStream_t stream) {
  g_ring_cleared.ws = seg ? seg->ptr : nullptr;
  g_ring_cleared.bytes = seg ? seg->words * 4 : 0;
  g_ring_cleared.stream = stream;
}

static thread_local bool g_exact = false;
void lstm_persist_set_exact(bool exact) { g_exact = exact; }
bool lstm_persist_exact() { return g_exact; }

static int run(bool fwd, int B, int T, int D, int H, int max_len, const int32_t *len,
               const float *const kernel[2], float *const gates[2], float *const cs[2], float *out,
               const float *dout, int *status, void *ws, size_t ws_bytes, float **db_part_out, int *db_rows_out,
               hipStream_t stream, const float *x = nullptr, const float *const bias[2] = nullptr, uint32_t *rowmax = nullptr,
               bool *rowmax_done = nullptr, void *xws = nullptr, const EmitArgs *emit = nullptr) {
  if (!lstm_persist_supported(B, T, H)) return fail(NABU_EUNSUP, "persistent LSTM: unsupported B=%d H=%d", B, H);
  const size_t need = lstm_persist_ws_bytes(B, T, H);
  if (ws_bytes < need) return fail(NABU_EWS, "persistent LSTM: workspace %zu < %zu", ws_bytes, need);
  float *db_part = reinterpret_cast<float *>(static_cast<char *>(ws) + need - db_part_bytes(B, H));
  float *amax_part = db_part + db_part_bytes(B, H) / (2 * sizeof(float));
  int shards = 0;
  const int Bc = chunk_rows(B, H, fwd, T);
  bool kept = rowmax != nullptr;
  for (int pass = 0; pass < 2; ++pass) {     // pass 0 validates every chunk, pass 1 enqueues them
    shards = 0;
    if (pass == 1 && fwd && x && lstm_mx_supported(B, H)) {
      FillSeg ring;      // cleared by prepare_x's own fill
      const bool with_ring = lstm_persist_ring_seg(true, B, T, H, ws, &ring);
      if (int e = lstm_mxh_prepare_x(B, T, D, x, xws, stream, with_ring ? &ring : nullptr)) return e;
      if (with_ring) lstm_persist_ring_cleared(&ring, stream);
    }
    for (int b0 = 0; b0 < B; b0 += Bc) {
      RowMax rm = {rowmax ? rowmax + (size_t)b0 * T : nullptr, (unsigned)((size_t)B * T), false};
      const int nb = B - b0 < Bc ? B - b0 : Bc;
      float *g2[2] = {gates[0] + (size_t)b0 * T * 4 * H, gates[1] + (size_t)b0 * T * 4 * H};
      float *c2[2] = {cs[0] + (size_t)b0 * T * H, cs[1] + (size_t)b0 * T * H};
      const int e = run_chunk(fwd, nb, T, D, H, max_len, len + b0, kernel, g2, c2,
                              out ? out + (size_t)b0 * T * 2 * H : nullptr,
                              dout ? dout + (size_t)b0 * T * 2 * H : nullptr, status, ws, ws_bytes, db_part, amax_part, &shards,
                              stream, x ? x + (size_t)b0 * T * D : nullptr, bias, pass == 0, &rm, xws, b0, emit);
      if (e) return e;
      kept = kept && rm.kept;
    }
  }
  if (rowmax_done) *rowmax_done = kept;
  if (db_part_out) *db_part_out = db_part;      // (the maxima follow at db_part + lstm_persist_db_floats(B, H))
  if (db_rows_out) *db_rows_out = shards;
  return 0;
}

static int run_chunk(bool fwd, int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2], float *out,
                     const float *dout, int *status, void *ws, size_t ws_bytes, float *db_part, float *amax_part,
                     int *shard_base, hipStream_t stream, const float *x, const float *const bias[2], bool dry, RowMax *rm,
                     const void *xws, int xrow0, const EmitArgs *emit) {
  PersistArgs a;
  a.emit = EmitArgs{};
  if (emit && fwd) {
    // only the fp16-plane kernels with 16 units per workgroup write companions (the caller asked lstm_persist_emits)
    if (!lstm_persist_emits(B, T, H, max_len) || xrow0 != 0)
      return fail(NABU_EINVAL, "persistent LSTM: this launch cannot write the packed companions");
    a.emit = *emit;
    a.emit.b0 = xrow0;
  }
  a.rowmax_part = nullptr; a.rowmax_stride = 0;
  { const char *e = getenv("NABU_PERSIST_DEBUG"); a.dbg = e ? atoi(e) : 0; }
  if (lstm_mx_supported(B, H)) {
    const bool xin = fwd && x != nullptr;
    if (xin && (lstm_mxf_supported(B, H) || B > lstm_mx_chunk_rows() || !bias || !xws || D > 64 || D % 8))
      return fail(NABU_EINVAL, "persistent LSTM (mxh): the in-kernel input projection does not take this shape");
    a.B = B; a.T = T; a.D = D; a.H = H; a.max_len = max_len; a.nshard = (B + 7) / 8;
    a.len = len;
    for (int i = 0; i < 2; ++i) { a.kernel[i] = kernel[i]; a.gates[i] = gates[i]; a.cs[i] = cs[i]; a.bias[i] = nullptr; }
    a.out = out; a.dout = dout; a.x = nullptr;
    a.xplanes = nullptr; a.xscale = nullptr;
    if (xin) {       // (the planes of the whole batch were written by run(); this chunk's rows)
      a.xscale = static_cast<const float *>(xws);
      a.xplanes = static_cast<const char *>(xws) + 1024 + (size_t)xrow0 * T * 256;
      a.bias[0] = bias[0]; a.bias[1] = bias[1];
    }
    a.db_part = db_part; a.amax_part = amax_part; a.shard_base = *shard_base;
    *shard_base += a.nshard;
    a.status = status;
    a.table = static_cast<unsigned *>(ws);
    a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
    a.timeout_ticks = g_timeout_ticks;
    if (lstm_mxf_supported(B, H)) {     // 33 .. 64 rows at H = 512: sixteen units of 8 rows, 128 gate columns per workgroup
      if (!dry)
        if (int e = ring_reset(ws, TABLE_BYTES + lstm_mxf_ring_bytes(fwd, H), stream)) return e;
      return lstm_mxf_launch(fwd, H, a, stream, dry);
    }
    if (B > lstm_mx_chunk_rows()) return fail(NABU_EINVAL, "persistent LSTM (mxh): a launch takes <= %d rows", lstm_mx_chunk_rows());
    if (!dry)
      if (int e = ring_reset(ws, TABLE_BYTES + lstm_mxh_ring_bytes(fwd, H), stream)) return e;
    if (!fwd && rm->part) {     // the backward kernel keeps the frames' maxima of dz
      a.rowmax_part = rm->part; a.rowmax_stride = rm->stride;
      rm->kept = true;
    }
    return lstm_mxh_launch(fwd, H, a, stream, dry);
  }
  int BS = pick_bs(B, H, fwd);
  if ((a.dbg & 16) && 2 * ((B + 7) / 8) * (H / UC) <= cu_count()) BS = 8;
  a.B = B; a.T = T; a.D = D; a.H = H; a.max_len = max_len; a.nshard = (B + BS - 1) / BS;
  a.len = len;
  for (int i = 0; i < 2; ++i) { a.kernel[i] = kernel[i]; a.gates[i] = gates[i]; a.cs[i] = cs[i]; }
  a.out = out; a.dout = dout;
  a.x = x; a.bias[0] = bias ? bias[0] : nullptr; a.bias[1] = bias ? bias[1] : nullptr;
  a.xplanes = nullptr; a.xscale = nullptr;
  const int XK = (fwd && x && bias && BS == 4 && lstm_persist_fuses_input(B, T, D, H)) ? D / 4 : 0;
  if (fwd && x && !XK) return fail(NABU_EINVAL, "persistent LSTM: the in-kernel input projection does not take this shape");
  a.db_part = db_part; a.shard_base = *shard_base;
  a.amax_part = amax_part;
  *shard_base += a.nshard;
  a.status = status;
  a.table = static_cast<unsigned *>(ws);
  a.xbuf = static_cast<char *>(ws) + TABLE_BYTES;
  a.timeout_ticks = g_timeout_ticks;
  const int NU = 2 * a.nshard, P = H / UC;
  const int grid = NU * P;
  const int per_cu = (BS == 4 || grid > cu_count()) ? 2 : 1;
  if (grid > per_cu * cu_count())
    return fail(NABU_EUNSUP, "persistent LSTM: %d workgroups > %d x %d CUs", grid, per_cu, cu_count());
  if (!dry)
    if (int e = ring_reset(ws, TABLE_BYTES + ring_bytes(fwd, BS, a.nshard, H), stream)) return e;
  // dynamic LDS chosen so that exactly `per_cu` workgroups fit on a CU (160 KiB)
  const size_t lds = BS == 4 ? 64 * 1024 : (grid > cu_count() ? 72 * 1024 : 96 * 1024);
#define NABU_PERSIST_CASE(h)                                                                         \
  case h:                                                                                            \
    if (XK == 10) return launch(lstm_persist_fwd_kernel<h / 16, 4, 10>, a, grid, 256, lds, stream, dry);    \
    if (BS == 4)                                                                                     \
      return fwd ? launch(lstm_persist_fwd_kernel<h / 16, 4>, a, grid, 256, lds, stream, dry)        \
                 : launch(lstm_persist_bwd_kernel<h, 4>, a, grid, 256, lds, stream, dry);            \
    return fwd ? launch(lstm_persist_fwd_kernel<h / 32, 8>, a, grid, 512, lds, stream, dry)          \
               : launch(lstm_persist_bwd_kernel<h, 8>, a, grid, 512, lds, stream, dry);
  switch (H) {
    NABU_PERSIST_CASE(64)
    NABU_PERSIST_CASE(128)
    NABU_PERSIST_CASE(256)
    NABU_PERSIST_CASE(512)
  }
  return fail(NABU_EUNSUP, "persistent LSTM: unsupported H=%d", H);
}

int lstm_persist_fwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     float *out, int *status, void *ws, size_t ws_bytes, hipStream_t stream, const float *x,
                     const float *const bias[2], void *xws, const EmitArgs *emit) {
  return run(true, B, T, D, H, max_len, len, kernel, gates, cs, out, nullptr, status, ws, ws_bytes, nullptr, nullptr,
             stream, x, bias, nullptr, nullptr, xws, emit);
}
// companions come out of the kernel itself only on the fp16-plane kernels with 16 units per workgroup, ONE launch of
// <= 32 rows (lstm_persist_mxh.hip), when the recurrence visits every frame (a frame it never visits is never written)
// and no debug variant is selected
bool lstm_persist_emits(int B, int T, int H, int max_len) {
  if (getenv("NABU_PERSIST_DEBUG") && atoi(getenv("NABU_PERSIST_DEBUG"))) return false;
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_EMIT"); env = e ? atoi(e) : 1; }
  if (!env || max_len != T || !lstm_persist_supported(B, T, H) || !lstm_mx_supported(B, H)) return false;
  return B <= lstm_mx_chunk_rows() && !lstm_mxf_supported(B, H);
}
// (sized from the ONE eligibility test, lstm_persist_fuses_input: no workspace where the projection is not taken)
size_t lstm_persist_xws_bytes(int B, int T, int D, int H) {
  return (lstm_mx_supported(B, H) && lstm_persist_fuses_input(B, T, D, H)) ? lstm_mxh_xws_bytes(B, T) : 0;
}

// narrow input (D = 40: 4 rows x D elements fit the 256 lanes of one prefetch), every launch of the forward pass on the
// 4-row geometry: the kernel projects the input itself (no x . Wx GEMM in front of it)
bool lstm_persist_fuses_input(int B, int T, int D, int H) {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_FUSE_INPUT"); env = e ? atoi(e) : 1; }
  if (!env || !lstm_persist_supported(B, T, H)) return false;
  if (lstm_mx_supported(B, H))     // fp16-plane kernels: any narrow input of <= 64 features, one launch of <= 32 rows
    return D <= 64 && D % 8 == 0 && B <= lstm_mx_chunk_rows() && !lstm_mxf_supported(B, H);
  if (D != 40) return false;
  if ((size_t)B * T * D * 4 >= 0x80000000ull) return false;
  const int Bc = chunk_rows(B, H, true, T);
  for (int b0 = 0; b0 < B; b0 += Bc)
    if (pick_bs(B - b0 < Bc ? B - b0 : Bc, H, true) != 4) return false;
  return true;
}

int lstm_persist_bwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     const float *dout, int *status, void *ws, size_t ws_bytes, float **db_part, int *db_rows,
                     hipStream_t stream, uint32_t *rowmax, bool *rowmax_done) {
  return run(false, B, T, D, H, max_len, len, kernel, gates, cs, nullptr, dout, status, ws, ws_bytes, db_part, db_rows,
             stream, nullptr, nullptr, rowmax, rowmax_done);
}

}  // namespace nabu
