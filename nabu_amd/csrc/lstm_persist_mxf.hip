// lstm_persist_mxf.hip — the fp16-plane persistent recurrence (lstm_persist_mxh.hip) with THIRTY-TWO hidden units per
// workgroup: batches of 33 .. 64 rows (cfg5: 64) as up to SIXTEEN units of 8 rows, two units per XCD, 16 workgroups per
// unit.  Round 4.
//
// WHY.  With fp16 planes a workgroup's slice of W_h takes 128 registers per lane where the bf16 planes took 192: there is
// room for twice the slice — 128 gate columns x H (256 registers).  Fewer, fatter workgroups per unit halve what a unit
// exchanges per batch row: the backward reduce-scatter moves rows x H x 4 bytes per WORKGROUP of the unit, so 16
// workgroups of 128 columns move half of what 32 workgroups of 64 columns do; and the instruction's N side is full with
// 8 rows x 2 planes, which the 16-rows-per-unit kernels (one plane per instruction) pay three instructions per tile
// for.  Against those kernels at the cfg5 layer shape (64 x 400 x 2048, H = 512): see DESIGN.md section 5.2.
//
// GEOMETRY.  block b -> XCD b % 8 (checked at run time as before), local = b / 8 in [0, 32): unit = XCD + 8 (local / P),
// slot = local % P, P = H / 32 workgroups per unit; unit = (direction, 8 batch rows) as in lstm_persist_mxh.hip.  A
// workgroup owns 32 hidden units = 128 gate columns = 8 column tiles (gate c, unit half).  Protocol, rings, deferred
// stores, row scales, tag bits: lstm_persist_mxh.hip / lstm_persist_mxh.h — what differs is spelled out below.
#include "lstm_persist_mxh.h"

namespace nabu {

constexpr int MXF_UC = 32;             // hidden units per workgroup
constexpr int MXF_NU = 16;             // units per launch: two per XCD

// W_h as planes takes 256 registers per lane here: the l plane lives in the ACCUMULATION registers (the other half of the
// unified file) and is fed to the matrix instruction from there — hipcc keeps matrix operands in ordinary registers and
// otherwise copies four of them in front of every instruction (v_accvgpr_read: 0.2 us per step).  An l-plane instruction
// is always followed, at least four matrix instructions later, by the compiler's own h-plane instruction on the same
// accumulators, so every result the vector ALU reads comes out of an instruction whose hazards hipcc pads itself.
// (mxf_pin_acc / mxf_mfma_acc: lstm_persist_mxh.h)

__device__ __forceinline__ void mxf_identity(int P, int *unit, int *slot) {
  const int xcd = blockIdx.x % 8, local = blockIdx.x / 8;
  *unit = xcd + 8 * (local / P);
  *slot = local % P;
}

// ===========================================================================
// forward
template <int H>
struct MxfFwdLds {
  static constexpr int ROWF = 33 * 4;                        // floats per (wave, row): 32 units x 4 gates + pad
  static constexpr int PART = 0;                             // [2][4 waves][8 rows][ROWF]
  static constexpr int XST = PART + 2 * 4 * MXR * ROWF;      // [2][4 gates][256] prefetched x-projection
  static constexpr int FLAG = XST + 2 * 4 * 256;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mxf_fwd_kernel(PersistArgs p) {
  using L = MxfFwdLds<H>;
  constexpr int P = H / MXF_UC;
  constexpr int KW = H / 4;          // k values multiplied by one wave
  constexpr int NKS = KW / 32;       // k-steps of 32 per wave
  static_assert(NKS >= 1 && 2 * P <= 32, "mxf forward: 128 <= H <= 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *part = smem + L::PART, *xst = smem + L::XST;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mxf_identity(P, &unit, &slot);
  if (unit >= NU || blockIdx.x / 8 >= 2 * P) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * MXF_UC, b0 = shard * MXR;
  const int T = p.T;
  // matrix-phase identity: n = N index (plane half, row) / M index (unit inside a 16-unit half), q = k group / column group
  const int n = lane & 15, q = lane >> 4;
  // finishing identity: (row 2 w + r2, unit u32), one per lane
  const int u32 = lane & 31, r2 = lane >> 5;
  const int frow = 2 * w + r2, fb = b0 + frow;
  const int n_f = fb < p.B ? p.len[fb] : 0;

  // this lane's slice of W_h as two scaled fp16 planes, A operands: tile (gate c, unit half uh) = column (c, U0 + 16 uh +
  // n), k = w KW + 32 j + 8 q + e.  Column scale: largest magnitude over all k (lanes q: shuffles; the four waves: LDS).
  // inv[c]: of the column (c, U0 + u32) this lane FINISHES — fetched from the lane that multiplies it
  u32x4 Wp[2][8][NKS];
  float inv[4];
  {
    const float *Wh = p.kernel[dir] + ((size_t)p.D + (size_t)w * KW + 8 * q) * 4 * H + U0 + n;
    float mx[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const int c = ct >> 1, uh = ct & 1;
      mx[ct] = 0.f;
#pragma unroll
      for (int j = 0; j < NKS; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) mx[ct] = fmaxf(mx[ct], fabsf(Wh[((size_t)32 * j + e) * 4 * H + (size_t)c * H + 16 * uh]));
      mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 16));
      mx[ct] = fmaxf(mx[ct], __shfl_xor(mx[ct], 32));
      if (q == 0) part[w * 128 + ct * 16 + n] = mx[ct];
    }
    __syncthreads();
    float sc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      const float m = fmaxf(fmaxf(part[ct * 16 + n], part[128 + ct * 16 + n]), fmaxf(part[256 + ct * 16 + n], part[384 + ct * 16 + n]));
      sc[ct] = mxh_scale_of(m);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ct = 2 * c + (u32 >> 4);
      const float m = fmaxf(fmaxf(part[ct * 16 + (u32 & 15)], part[128 + ct * 16 + (u32 & 15)]),
                            fmaxf(part[256 + ct * 16 + (u32 & 15)], part[384 + ct * 16 + (u32 & 15)]));
      inv[c] = mxh_inv_scale_of(m) * MXH_HINV;
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
        const int c = ct >> 1, uh = ct & 1;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[((size_t)32 * j + e) * 4 * H + (size_t)c * H + 16 * uh] * sc[ct];
        mxh_split8(x, Wp[0][ct][j], Wp[1][ct][j]);
        mxf_pin_acc(Wp[1][ct][j]);
      }
  }
  float c_state = 0.f, h_state = 0.f;
  if (!unit_handshake(p, unit, slot, MXF_NU, P, flag)) return;
  const bool coloc = flag[1] != 0;
  clock_stamp(p, 0, 0);

  // exchange slot of a unit: cells of 16 bytes = 8 consecutive k of one (plane, row): [k / 8][16 = plane * 8 + row]
  const size_t slot_bytes = (size_t)16 * H * 2;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * RING * slot_bytes, 0, (int)(RING * slot_bytes), 0x00020000);
  constexpr int KGW = KW / 8;                      // k groups per wave
  constexpr unsigned KSTEP_BYTES = 4 * 16 * 16;    // 4 k groups
  const unsigned off1 = (unsigned)((((size_t)w * KGW + q) * 16 + n) * 16);
  // my published piece (lanes u32 & 7 = plane 0, 1): units U0 + (u32 & 24) .. + 7 of row frow
  const int ppl = u32 & 7;
  const bool pub_lane = ppl < 2;
  const unsigned pub_off = (unsigned)((((size_t)(U0 >> 3) + (u32 >> 3)) * 16 + ppl * 8 + frow) * 16);
  const u32x4 sent4 = {SENT, SENT, SENT, SENT};

  // x-projection of step s (bias included): the four gates of (row frow, unit u32), one step ahead by LDS-DMA
  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const unsigned goff = (unsigned)(((size_t)fb * T * 4 * H + U0 + u32) * 4);
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
  // results of step s go to HBM at the top of step s + 1, behind that step's exchange loads (lstm_persist_mxh.hip)
  float d_g[4] = {0.f, 0.f, 0.f, 0.f}, d_c = 0.f, d_h = 0.f;
  int d_t = 0, d_to = 0;
  bool d_act = false, d_any = false;
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc(p.gates[dir], 0, (int)((size_t)p.B * T * 4 * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.cs[dir], 0, (int)((size_t)p.B * T * H * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)p.B * T * 2 * H * 4), 0x00020000);
  const unsigned coff = (unsigned)(((size_t)fb * T * H + U0 + u32) * 4);
  const unsigned ooff = (unsigned)(((size_t)fb * T * 2 * H + (size_t)dir * H + U0 + u32) * 4);
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
    mxf32x4 acc[8];
    unsigned long long t_fail = 0;
    int fails = 0;
    // (a) h_{s-1} as planes: the poll loop IS the operand fetch (4 KiB of full lines per wave)
    u32x4 b1[NKS];
#pragma unroll
    for (int j = 0; j < NKS; ++j) b1[j] = zero4;
    if (s > 0 && !(p.dbg & 1)) {
      const unsigned base = (unsigned)(((s - 1) % RING) * slot_bytes);
      bool first = true;
      for (;;) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < NKS; ++j) b1[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + off1 + j * KSTEP_BYTES, 0, 16);
        if (first) { result_stores(); first = false; }
#pragma unroll
        for (int j = 0; j < NKS; ++j) mx = mx_max4(mx, b1[j]);
        if (__all(mx != SENT)) break;
        if (poll_round_failed(p, flag, lane, fails, t_fail, 1)) break;
      }
    } else {
      result_stores();
      wait_vm<0>();
    }
    // (b) product: 8 column tiles x NKS k-steps x {W_l.B, W_h.B}, small terms first; next step's x-projection is
    // requested from inside the matrix stream
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
    if (s > 0 && !(p.dbg & 2)) {
#pragma unroll
      for (int j = 0; j < NKS; ++j) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = mxf_mfma_acc(Wp[1][ct][j], b1[j], acc[ct]);
        if (j == 0) { fetch_x_part(s + 1, 0); __builtin_amdgcn_sched_barrier(0); }
        if (j == 1 || NKS == 1) { fetch_x_part(s + 1, 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) acc[ct] = MXH_MFMA(Wp[0][ct][j], b1[j], acc[ct]);
        if (j == 0) { fetch_x_part(s + 1, 1); __builtin_amdgcn_sched_barrier(0); }
        if (j == 1 || NKS == 1) { fetch_x_part(s + 1, 3); __builtin_amdgcn_sched_barrier(0); }
      }
    } else {
      fetch_x(s + 1);
    }
    // the two plane halves of N: lanes n and n ^ 8 end with the same sums (row n & 7; units 16 uh + 4 q + i, gate c)
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
      acc[ct].x += mx_dpp<DPP_ROR8>(acc[ct].x);
      acc[ct].y += mx_dpp<DPP_ROR8>(acc[ct].y);
      acc[ct].z += mx_dpp<DPP_ROR8>(acc[ct].z);
      acc[ct].w += mx_dpp<DPP_ROR8>(acc[ct].w);
    }
    // partial sums -> LDS [wave][row][unit][4 gates]: lanes n < 8 write units 16 uh + 4 q + {0, 1}, the others + {2, 3}
    float *const pbuf = part + (s & 1) * (4 * MXR * L::ROWF);
    {
      const bool lo = n < 8;
#pragma unroll
      for (int uh = 0; uh < 2; ++uh) {
        float *d = pbuf + ((size_t)(w * MXR + (n & 7))) * L::ROWF + (16 * uh + 4 * q + (lo ? 0 : 2)) * 4;
        const mxf32x4 v0 = {lo ? acc[uh].x : acc[uh].z, lo ? acc[2 + uh].x : acc[2 + uh].z, lo ? acc[4 + uh].x : acc[4 + uh].z,
                            lo ? acc[6 + uh].x : acc[6 + uh].z};
        const mxf32x4 v1 = {lo ? acc[uh].y : acc[uh].w, lo ? acc[2 + uh].y : acc[2 + uh].w, lo ? acc[4 + uh].y : acc[4 + uh].w,
                            lo ? acc[6 + uh].y : acc[6 + uh].w};
        *reinterpret_cast<mxf32x4 *>(d) = v0;
        *reinterpret_cast<mxf32x4 *>(d + 4) = v1;
      }
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;

    // (c) gates of (row frow, unit u32): the four waves' partial sums, descaled (exact), plus the x-projection
    mxf32x4 z;
    {
      const float *xs = xst + (s & 1) * 1024 + tid;
      const float *pr = pbuf + (size_t)frow * L::ROWF + u32 * 4;
      mxf32x4 sum = *reinterpret_cast<const mxf32x4 *>(pr);
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) sum += *reinterpret_cast<const mxf32x4 *>(pr + (size_t)ww * MXR * L::ROWF);
      z = (mxf32x4){xs[0] + sum.x * inv[0], xs[256] + sum.y * inv[1], xs[512] + sum.z * inv[2], xs[768] + sum.w * inv[3]};
    }
    const float gi = fast_sigmoid(z.x), gj = fast_tanh(z.y), gf = fast_sigmoid(z.z + 1.0f), go = fast_sigmoid(z.w);
    const bool act = s < n_f;
    const float c_new = c_state * gf + gi * gj;
    const float h_new = fast_tanh(c_new) * go;
    if (act) { c_state = c_new; h_state = h_new; }

    // (d) publish h_s as two planes: lane 8 g + pl collects the four pair words of plane pl -> one 16-byte store
    {
      const float hs = h_state * MXH_HSCALE;
      const unsigned w0 = mxh_cvt2(hs, 0.f) & 0xFFFFu;
      const float r = hs - (float)__builtin_bit_cast(mxh16x2, w0).x;
      const unsigned w1 = mxh_cvt2(r, 0.f) & 0xFFFFu;
      const unsigned pr0 = w0 | (mx_dppu<DPP_XOR1>(w0) << 16), pr1 = w1 | (mx_dppu<DPP_XOR1>(w1) << 16);
      const u32x4 v0 = {pr0, mx_dppu<0x102>(pr0), mx_dppu<0x104>(pr0), mx_dppu<0x106>(pr0)};
      const u32x4 v1 = {mx_dppu<0x111>(pr1), mx_dppu<0x101>(pr1), mx_dppu<0x103>(pr1), mx_dppu<0x105>(pr1)};
      const u32x4 pv = ppl == 0 ? v0 : v1;
      xstore(pv, rs, (pub_lane && s + 1 < p.max_len) ? (unsigned)((s % RING) * slot_bytes) + pub_off : OOB, coloc);
      xstore(sent4, rs, (pub_lane && s >= 2) ? (unsigned)(((s - 2) % RING) * slot_bytes) + pub_off : OOB, coloc);
    }
    {
      const int t_g = dir ? n_f - 1 - s : s;
      d_any = true; d_act = act; d_t = t_g; d_to = act ? t_g : s;
      d_g[0] = gi; d_g[1] = gj; d_g[2] = gf; d_g[3] = go;
      d_c = c_new;
      d_h = act ? h_new : 0.f;
    }
  }
  result_stores();
  clock_stamp(p, 0, 1);
}

// ===========================================================================
// backward
template <int H>
struct MxfBwdLds {
  static constexpr int DROWB = 128 * 2 + 16;                 // bytes per slot row of dz planes: 128 columns fp16 + pad
  static constexpr int DZ = 0;                               // [2][16 = plane * 8 + row][DROWB] bytes
  static constexpr int INVD = (2 * 16 * DROWB + 15) / 16 * 4;   // floats: [2][8] inverse row scales of dz
  static constexpr int XST = INVD + 16;                      // [2 parities][2 passes][4][256] prefetched saved values
  static constexpr int RED = XST + 2 * 2 * 4 * 256;          // [8 rows][128] floats, final reductions
  static constexpr int FLAG = RED + 8 * 128;
  static constexpr int TOTAL = FLAG + 4;
};

template <int H>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_mxf_bwd_kernel(PersistArgs p) {
  using L = MxfBwdLds<H>;
  constexpr int P = H / MXF_UC;      // workgroups per unit = sources = destinations
  constexpr int NT = H / 16 / 4;     // 16-k output tiles per wave (two per destination workgroup)
  constexpr int NQ = P / 8;          // source pieces per lane and pass
  static_assert(NT >= 2 && NT % 2 == 0 && NQ >= 1 && 2 * P <= 32, "mxf backward: 256 <= H <= 512");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char *dzs = reinterpret_cast<char *>(smem) + L::DZ;
  float *invd = smem + L::INVD, *xst = smem + L::XST, *red = smem + L::RED;
  int *flag = reinterpret_cast<int *>(smem + L::FLAG);

  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int NU = 2 * p.nshard;
  int unit, slot;
  mxf_identity(P, &unit, &slot);
  if (unit >= NU || blockIdx.x / 8 >= 2 * P) return;
  const int dir = unit & 1, shard = unit >> 1;
  const int U0 = slot * MXF_UC, b0 = shard * MXR;
  const int T = p.T;
  const int n = lane & 15, q = lane >> 4;                 // matrix-phase identity
  // exchange / gate identity, two passes over the unit halves (units 16 ps + ...): source group s8, k quad kq, row
  // 2 w + r2; after the butterfly: unit 16 ps + 4 kq + (s8 >> 1), gate pair dup.  The 32 lanes of a row are half a wave.
  const int s8 = lane & 7, kq = (lane >> 3) & 3, r2 = lane >> 5;
  const int grow = 2 * w + r2, gb = b0 + grow;
  const int dup = s8 & 1;
  const int n_g = gb < p.B ? p.len[gb] : 0;
  int gu[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) gu[ps] = 16 * ps + 4 * kq + (s8 >> 1);
  constexpr int HT = NT / 2;                              // tiles per product half
  constexpr int QT = HT / 2;                              // tiles per lane half and product half

  // A operands: W^T as two scaled fp16 planes.  Row m = output k = 16 (NT w + t) + n; reduction index c' = 32 j + 8 q + e
  // = 4 unit + gate over the workgroup's 128 columns (4 k-steps).  Row scale over those 128 columns (lanes q: shuffles).
  u32x4 Wp[2][NT][4];
  float inv_sel[2][QT][4];
  {
    float inv_lane[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float *Wh = p.kernel[dir] + ((size_t)p.D + 16 * (NT * w + t) + n) * 4 * H + U0;
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(Wh[(size_t)(e & 3) * H + 8 * j + 2 * q + (e >> 2)]));
      m = fmaxf(m, __shfl_xor(m, 16));
      m = fmaxf(m, __shfl_xor(m, 32));
      const float sc = mxh_scale_of(m);
      inv_lane[t] = mxh_inv_scale_of(m);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = Wh[(size_t)(e & 3) * H + 8 * j + 2 * q + (e >> 2)] * sc;
        mxh_split8(x, Wp[0][t][j], Wp[1][t][j]);
        mxf_pin_acc(Wp[1][t][j]);
      }
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = __shfl(inv_lane[hf * HT + t], 4 * q + i);
          const float hi = __shfl(inv_lane[hf * HT + QT + t], 4 * q + i);
          inv_sel[hf][t][i] = n < 8 ? lo : hi;
        }
  }
  float dc_state[2] = {0.f, 0.f};
  // bias-gradient sums in float64, added behind the publish (lstm_persist_mxh.hip, backward kernel)
  double db0[2] = {0.0, 0.0}, db1[2] = {0.0, 0.0};
  float am0[2] = {0.f, 0.f}, am1[2] = {0.f, 0.f};
  if (!unit_handshake(p, unit, slot, MXF_NU, P, flag)) return;
  const bool coloc = flag[1] != 0;
  clock_stamp(p, 1, 0);

  // ring slot = [dest P][src P][8 rows][8 k quads] x 16 bytes: a piece = the 32 units of its destination
  const size_t piece_bytes = (size_t)MXR * MXF_UC * 4;
  const size_t block_bytes = (size_t)P * piece_bytes;
  const size_t slot_bytes = (size_t)P * block_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      p.xbuf + (size_t)unit * MXHRINGB * slot_bytes, 0, (int)(MXHRINGB * slot_bytes), 0x00020000);
  unsigned in_off[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps)
    in_off[ps] = (unsigned)((size_t)slot * block_bytes + ((size_t)s8 * MXR + grow) * 128 + (4 * ps + kq) * 16);
  constexpr unsigned SRC8 = 8 * MXR * 128;       // 8 sources further

  const i32x4 rg = raw_rsrc(p.gates[dir], (unsigned)((size_t)p.B * T * 4 * H * 4));
  const i32x4 rc = raw_rsrc(p.cs[dir], (unsigned)((size_t)p.B * T * H * 4));
  const i32x4 rd = raw_rsrc(p.dout, (unsigned)((size_t)p.B * T * 2 * H * 4));
  unsigned goff[2], coff[2], doff[2];
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    goff[ps] = (unsigned)(((size_t)gb * T * 4 * H + (size_t)(2 * dup) * H + U0 + gu[ps]) * 4);
    coff[ps] = (unsigned)(((size_t)gb * T * H + U0 + gu[ps]) * 4);
    doff[ps] = (unsigned)(((size_t)gb * T * 2 * H + (size_t)dir * H + U0 + gu[ps]) * 4);
  }
  // saved forward values of step s (pass ps), one step ahead: A, B = activations of my two gates, C = c / c_prev,
  // D = dout (dup 0)
  auto fetch_part = [&](int s, int idx) {
    const int ps = idx >> 2, part_i = idx & 3;
    const bool act = s >= 0 && s < n_g;
    const int t = dir ? n_g - 1 - s : s;
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
  const bool st_ok = gb < p.B && !(p.dbg & 128);
  float d_0[2] = {0.f, 0.f}, d_1[2] = {0.f, 0.f};
  int d_t = 0;
  bool d_any = false;
  auto dz_stores = [&]() {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const unsigned o = (d_any && st_ok) ? goff[ps] + (unsigned)d_t * (unsigned)(16 * H) : OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_0[ps]), rsg, o, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, d_1[ps]), rsg, o == OOB ? OOB : o + (unsigned)(4 * H), 0, 0);
    }
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  // GATE FACTORS AHEAD OF THE EXCHANGE (lstm_persist_mxh.hip, backward kernel): the saved values of the step were
  // prefetched a step ahead; claimed by a counted wait behind the first poll round's loads (everything but the VM_AFTER
  // operations issued since the prefetch: last step's publishes, this round's loads, the result stores) they become the
  // factors A, F0, F1, G of  dct = dc + (dout + dh) A,  dz = dct F0, (dct | dht) F1,  dc' = dct G  while the exchange is in flight
  constexpr int VM_AFTER = 2 * QT + 2 * NQ + 4;
  float fA[2], f0[2], f1[2], fG[2], f_dout[2];
  auto gate_factors = [&](int s) {
    asm volatile("" ::: "memory");
    const bool act = s < n_g;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const float *st = xst + ((s & 1) * 2 + ps) * 1024 + tid;
      const float sA = st[0], sB = st[256], sC = st[512], sD = st[768];
      const float pA = mx_dpp<DPP_XOR1>(sA), pB = mx_dpp<DPP_XOR1>(sB), pC = mx_dpp<DPP_XOR1>(sC), pD = mx_dpp<DPP_XOR1>(sD);
      const float gi = dup ? pA : sA, gj = dup ? pB : sB, gf = dup ? sA : pA, go = dup ? sB : pB;
      const float c = dup ? pC : sC, cprev = dup ? sC : pC;
      f_dout[ps] = dup ? pD : sD;
      const float tc = fast_tanh(c);
      fA[ps] = go * (1.f - tc * tc);
      const float a0 = dup ? cprev * gf * (1.f - gf) : gj * gi * (1.f - gi);
      const float a1 = dup ? tc * go * (1.f - go) : gi * (1.f - gj * gj);
      f0[ps] = act ? a0 : 0.f;
      f1[ps] = act ? a1 : 0.f;
      fG[ps] = gf;
    }
  };

  for (int s = p.max_len - 1; s >= 0; --s) {
    // (a) reduce-scatter input: the partial products of the previous iteration addressed to my 32 units, two passes
    u32x4 v[2][NQ];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int i = 0; i < NQ; ++i) v[ps][i] = zero4;
    const int it = p.max_len - 1 - s;                       // iteration count: slot it & 1, generation it >> 1
    const unsigned sbase = (unsigned)(((it - 1) & 1) * slot_bytes);
    const bool have_in = it > 0 && !(p.dbg & 1);
    if (have_in) {
      unsigned long long t_fail = 0;
      int fails = 0;
      bool first = true;
      const bool want1 = (((it - 1) >> 1) & 1) != 0;        // the tag of the pieces published in iteration it - 1
      __builtin_amdgcn_s_sleep(4);      // (a first round issued at once fails)
      for (;;) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int i = 0; i < NQ; ++i)
            v[ps][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, sbase + in_off[ps] + (unsigned)i * SRC8, 0, 16);
        if (first) {
          dz_stores();
          wait_vm<VM_AFTER>();
          gate_factors(s);
          first = false;
        }
        unsigned a = ~0u, o = 0u;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            a &= v[ps][i].x & v[ps][i].y & v[ps][i].z & v[ps][i].w;
            o |= v[ps][i].x | v[ps][i].y | v[ps][i].z | v[ps][i].w;
          }
        if (__all(want1 ? (a & 1u) != 0 : (o & 1u) == 0)) break;
        if (poll_round_failed(p, flag, lane, fails, t_fail, 2)) break;
      }
    } else {
      dz_stores();
      wait_vm<0>();
      gate_factors(s);
    }
    char *const dzb = dzs + (s & 1) * (16 * L::DROWB);
    const bool act_g = s < n_g;
    float d0[2], d1[2];
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

      // (b) gate gradients of (row, unit) from the factors computed ahead of the exchange
      const float dht = f_dout[ps] + dh;
      const float dct = dc_state[ps] + dht * fA[ps];
      d0[ps] = dct * f0[ps];
      d1[ps] = (dup ? dht : dct) * f1[ps];
      if (act_g) dc_state[ps] = dct * fG[ps];
    }
    {
      // this row's largest |dz| over the workgroup's 128 columns = both passes of the row's 32 lanes (bit patterns)
      unsigned mb = max(max(__builtin_bit_cast(unsigned, d0[0]) & 0x7FFFFFFFu, __builtin_bit_cast(unsigned, d1[0]) & 0x7FFFFFFFu),
                        max(__builtin_bit_cast(unsigned, d0[1]) & 0x7FFFFFFFu, __builtin_bit_cast(unsigned, d1[1]) & 0x7FFFFFFFu));
      mb = max(mb, mx_dppu<DPP_XOR1>(mb));
      mb = max(mb, mx_dppu<DPP_XOR2>(mb));
      mb = max(mb, mx_dppu<DPP_HALF_MIRROR>(mb));
      mb = max(mb, mx_dppu<DPP_ROW_MIRROR>(mb));
      mb = max(mb, (unsigned)__builtin_amdgcn_ds_swizzle((int)mb, 0x401F));
      const unsigned ex = min(max(mb >> 23, 15u), 253u);
      const float sc = __builtin_bit_cast(float, (268u - ex) << 23);
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        unsigned ph, pl;
        mxh_split2x2(d0[ps] * sc, d1[ps] * sc, ph, pl);
        const unsigned o = (unsigned)grow * L::DROWB + (unsigned)(4 * gu[ps] + 2 * dup) * 2;
        *reinterpret_cast<unsigned *>(dzb + o) = ph;
        *reinterpret_cast<unsigned *>(dzb + o + 8 * L::DROWB) = pl;
        d_0[ps] = d0[ps]; d_1[ps] = d1[ps];
      }
      if ((lane & 31) == 0) invd[(s & 1) * 8 + grow] = __builtin_bit_cast(float, (ex - 14u) << 23);
      const int t_g = dir ? n_g - 1 - s : s;
      d_any = true; d_t = act_g ? t_g : s;
    }
    __syncthreads();                                            // the step's only barrier
    if (flag[0]) return;

    if (s > 0) {
      // (c) partial dh of step s - 1: dz planes [16 slots x 128 columns] against W^T, tile t -> 16 k of destination
      // (NT w + t) / 2, in two halves of NT / 2 tiles; lanes n < 8 publish the first tiles of a half, the others (same
      // sums) the rest; next step's saved values are requested from inside the matrix stream
      u32x4 b1[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b1[j] = *reinterpret_cast<const u32x4 *>(dzb + (unsigned)n * L::DROWB + 64 * j + 16 * q);
      const float idz = invd[(s & 1) * 8 + (n & 7)];
      const unsigned tag = (unsigned)(it >> 1) & 1u;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        mxf32x4 acc[HT];
#pragma unroll
        for (int t = 0; t < HT; ++t) acc[t] = (mxf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
#ifndef MXF_EXP_NOPROD
#pragma unroll
            for (int t = 0; t < HT; ++t)
              acc[t] = g == 0 ? mxf_mfma_acc(Wp[1][hf * HT + t][j], b1[j], acc[t]) : MXH_MFMA(Wp[0][hf * HT + t][j], b1[j], acc[t]);
#endif
            if (hf == 0) fetch_part(s - 1, 2 * j + g);     // one memory instruction behind every group
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
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          // tile -> (destination, half of its 32 units): piece (dest, me)[row n & 7][k quad 4 th + q]
          const int tile = NT * w + hf * HT + (n < 8 ? 0 : QT) + t;
          const unsigned po = (unsigned)((it & 1) * slot_bytes + (size_t)(tile >> 1) * block_bytes + (size_t)slot * piece_bytes +
                                         (size_t)(n & 7) * 128 + (4 * (tile & 1) + q) * 16);
          const mxf32x4 lo = acc[t], hi = acc[QT + t];
          const mxf32x4 o = {(n < 8 ? lo.x : hi.x) * inv_sel[hf][t][0] * idz, (n < 8 ? lo.y : hi.y) * inv_sel[hf][t][1] * idz,
                             (n < 8 ? lo.z : hi.z) * inv_sel[hf][t][2] * idz, (n < 8 ? lo.w : hi.w) * inv_sel[hf][t][3] * idz};
          const u32x4 ob = __builtin_bit_cast(u32x4, o);
          const u32x4 ot = {(ob.x & ~1u) | tag, (ob.y & ~1u) | tag, (ob.z & ~1u) | tag, (ob.w & ~1u) | tag};
          xstore(ot, rs, po, coloc);
        }
      }
    }
    // (behind the publish: nothing waits for these)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      db0[ps] += (double)d_0[ps]; db1[ps] += (double)d_1[ps];
      am0[ps] = fmaxf(am0[ps], fabsf(d_0[ps])); am1[ps] = fmaxf(am1[ps], fabsf(d_1[ps]));
    }
  }
  dz_stores();
  clock_stamp(p, 1, 1);
  // bias gradient / column maxima of my 128 gate columns over the unit's 8 rows
  __syncthreads();
  double *redd = reinterpret_cast<double *>(smem);      // [8 rows][128] doubles over the dz plane / staging area
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    redd[grow * 128 + (2 * dup) * 32 + gu[ps]] = db0[ps];
    redd[grow * 128 + (2 * dup + 1) * 32 + gu[ps]] = db1[ps];
  }
  __syncthreads();
  if (tid < 128) {
    double sum = 0.0;
#pragma unroll
    for (int r = 0; r < MXR; ++r) sum += redd[r * 128 + tid];
    p.db_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 5) * H + U0 + (tid & 31)] = (float)sum;
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    red[grow * 128 + (2 * dup) * 32 + gu[ps]] = am0[ps];
    red[grow * 128 + (2 * dup + 1) * 32 + gu[ps]] = am1[ps];
  }
  __syncthreads();
  if (tid < 128) {
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < MXR; ++r) m = fmaxf(m, red[r * 128 + tid]);
    p.amax_part[((size_t)(p.shard_base + shard) * 2 + dir) * 4 * H + (size_t)(tid >> 5) * H + U0 + (tid & 31)] = m;
  }
}

// ===========================================================================
// host side (called from lstm_persist.hip's run_chunk)
bool lstm_mxf_supported(int B, int H) {
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PERSIST_MXF"); env = e ? atoi(e) : 1; }
  return env != 0 && H == 512 && (B > 32 || env == 2) && B <= 64;     // 2: also batches of <= 32 rows (measurements)
}

size_t lstm_mxf_ring_bytes(bool fwd, int H) {
  const size_t P = H / MXF_UC;
  return fwd ? (size_t)MXF_NU * RING * 16 * H * 2 : (size_t)MXF_NU * MXHRINGB * P * P * MXR * MXF_UC * 4;
}

template <typename K>
static int mxf_launch(K kernel, const PersistArgs &a, int grid, size_t lds, hipStream_t stream, bool dry) {
  const void *fn = reinterpret_cast<const void *>(kernel);
  struct Seen { const void *fn; int dev, blocks; };
  static thread_local Seen seen[8] = {};
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
    return fail(NABU_EUNSUP, "persistent LSTM (mxf): %d workgroups cannot be co-resident (%d per CU)", grid, blocks);
  if (dry) return 0;          // validation pass (lstm_persist.hip, run)
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, stream, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

// one launch over 33 .. 64 rows; `a` comes filled from run_chunk (nshard = ceil(B / 8))
int lstm_mxf_launch(bool fwd, int H, const PersistArgs &a, hipStream_t stream, bool dry) {
  if (H != 512) return fail(NABU_EUNSUP, "persistent LSTM (mxf): unsupported H=%d", H);
  const int grid = 8 * 2 * (H / MXF_UC);
  return fwd ? mxf_launch(lstm_mxf_fwd_kernel<512>, a, grid, MxfFwdLds<512>::TOTAL * sizeof(float), stream, dry)
             : mxf_launch(lstm_mxf_bwd_kernel<512>, a, grid, MxfBwdLds<512>::TOTAL * sizeof(float), stream, dry);
}

}  // namespace nabu
