// lstm.hip — one bidirectional LSTM layer (layer.blstm of the reference):
// orchestration of the input-projection GEMMs, the recurrent kernels and the
// weight-gradient GEMMs, plus the one-launch-per-timestep recurrent kernels.
//
// Data layout in HBM (all fp32, batch-major):
//   x      [B,T,D]        layer input
//   out    [B,T,2H]       fw | bw hidden states, 0 for t >= len
//   reserve = gates_fw [B,T,4H] | gates_bw [B,T,4H] | cs_fw [B,T,H] | cs_bw [B,T,H]
//     gates_* first holds x·Wx+b (GEMM output), is overwritten in place by the
//     activations (i,g,f,o) in the forward recurrence and again in place by the
//     pre-activation gradients dz in the backward recurrence.
// The TF kernel [(D+H),4H] is used as stored: rows [0,D) = Wx, rows [D,D+H) = Wh.
#include "common.h"
#include "lstm_persist.h"

#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "gemm_args.h"

namespace nabu {

struct StepArgs {
  int B, T, D, H, max_len;
  const int32_t *len;
  const float *kernel[2];  // per direction
  float *gates[2];
  float *cs[2];
  float *out;          // fwd: written; bwd: unused
  const float *dout;   // bwd
  float *hstate;       // [2 pingpong][2 dir][B][H]
  float *cstate;       // fwd: c state [2][B][H]; bwd: dc carry [2][B][H]
};

constexpr int SB = 16;  // batch rows per block
constexpr int SU = 16;  // hidden units per block

// ---------------------------------------------------------------------------
// forward, one timestep, both directions.  grid (H/16, B/16, 2), 256 threads.
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(StepArgs p, int s) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int H = p.H, B = p.B, T = p.T;
  float *hs = smem;               // [SB][H]
  float *zs = smem + SB * H;      // [SB][4][SU]
  const int dir = blockIdx.z, u0 = blockIdx.x * SU, b0 = blockIdx.y * SB;
  const int tid = threadIdx.x;
  const float *hprev = p.hstate + ((size_t)((s & 1) * 2 + dir) * B) * H;
  float *hnext = p.hstate + ((size_t)(((s & 1) ^ 1) * 2 + dir) * B) * H;

  // stage h_{s-1} of this block's batch rows
  for (int i = tid; i < SB * H / 4; i += 256) {
    const int bl = i / (H / 4), k4 = i % (H / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b0 + bl < B) v = reinterpret_cast<const float4 *>(hprev + (size_t)(b0 + bl) * H)[k4];
    reinterpret_cast<float4 *>(hs + bl * H)[k4] = v;
  }
  __syncthreads();

  {  // recurrent product: thread = (batch row bl, gate g, unit quad q)
    const int q = tid & 3, g = (tid >> 2) & 3, bl = tid >> 4;
    const int ucol = u0 + 4 * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ucol < H) {
      const float *W = p.kernel[dir] + (size_t)p.D * 4 * H + (size_t)g * H + ucol;
      const float *hrow = hs + bl * H;
#pragma unroll 4
      for (int k = 0; k < H; ++k) {
        const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)k * 4 * H);
        const float hv = hrow[k];
        acc.x = fmaf(hv, w.x, acc.x);
        acc.y = fmaf(hv, w.y, acc.y);
        acc.z = fmaf(hv, w.z, acc.z);
        acc.w = fmaf(hv, w.w, acc.w);
      }
    }
    *reinterpret_cast<float4 *>(zs + (bl * 4 + g) * SU + 4 * q) = acc;
  }
  __syncthreads();

  {  // gates + state update: thread = (batch row bl, unit u)
    const int u = tid & 15, bl = tid >> 4;
    const int b = b0 + bl, hu = u0 + u;
    if (b < B && hu < H) {
      const int n = p.len[b];
      const size_t sidx = (size_t)b * H + hu;
      if (s < n) {
        const int t = dir ? n - 1 - s : s;
        float *gp = p.gates[dir] + ((size_t)b * T + t) * 4 * H + hu;
        const float zi = gp[0] + zs[(bl * 4 + 0) * SU + u];
        const float zj = gp[H] + zs[(bl * 4 + 1) * SU + u];
        const float zf = gp[2 * H] + zs[(bl * 4 + 2) * SU + u];
        const float zo = gp[3 * H] + zs[(bl * 4 + 3) * SU + u];
        const float i = sigmoidf_(zi), g = tanhf_(zj), f = sigmoidf_(zf + 1.0f), o = sigmoidf_(zo);
        float *cst = p.cstate + (size_t)dir * B * H + sidx;
        const float c = *cst * f + i * g;
        const float h = tanhf_(c) * o;
        gp[0] = i; gp[H] = g; gp[2 * H] = f; gp[3 * H] = o;
        p.cs[dir][((size_t)b * T + t) * H + hu] = c;
        p.out[((size_t)b * T + t) * 2 * H + (size_t)dir * H + hu] = h;
        *cst = c;
        hnext[sidx] = h;
      } else {
        // finished sequence: state frozen, output row s is zero (t >= len)
        hnext[sidx] = hs[bl * H + hu];
        p.out[((size_t)b * T + s) * 2 * H + (size_t)dir * H + hu] = 0.f;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// backward, one timestep (s descending), both directions.
//   dh_carry[b,u] = sum_col dz_{s+1}[b,col] * Wh[u,col]      (block-local)
//   dz_s from saved activations, written in place over the activations.
constexpr int DZC = 512;  // dz columns staged per LDS chunk

__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(StepArgs p, int s) {
  __shared__ __attribute__((aligned(16))) float dzs[SB][DZC];
  const int H = p.H, B = p.B, T = p.T;
  const int dir = blockIdx.z, u0 = blockIdx.x * SU, b0 = blockIdx.y * SB;
  const int tid = threadIdx.x;
  const int u = tid & 15, bl = tid >> 4;
  const int b = b0 + bl, hu = u0 + u;
  const bool valid = b < B && hu < H;
  const int n = b < B ? p.len[b] : 0;

  float dh = 0.f;
  if (s + 1 < p.max_len) {
    const float *Wrow = p.kernel[dir] + (size_t)(p.D + (hu < H ? hu : 0)) * 4 * H;
    for (int c0 = 0; c0 < 4 * H; c0 += DZC) {
      const int cw = min(DZC, 4 * H - c0);
      __syncthreads();
      for (int i = tid; i < SB * (DZC / 4); i += 256) {
        const int r = i / (DZC / 4), c4 = i % (DZC / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int bb = b0 + r;
        if (bb < B && 4 * c4 < cw) {
          const int nn = p.len[bb];
          if (s + 1 < nn) {
            const int t1 = dir ? nn - 2 - s : s + 1;
            v = *reinterpret_cast<const float4 *>(p.gates[dir] + ((size_t)bb * T + t1) * 4 * H + c0 + 4 * c4);
          }
        }
        *reinterpret_cast<float4 *>(&dzs[r][4 * c4]) = v;
      }
      __syncthreads();
      if (valid) {
#pragma unroll 4
        for (int c = 0; c < cw; c += 4) {
          const float4 w = *reinterpret_cast<const float4 *>(Wrow + c0 + c);
          const float4 d = *reinterpret_cast<const float4 *>(&dzs[bl][c]);
          dh = fmaf(d.x, w.x, dh);
          dh = fmaf(d.y, w.y, dh);
          dh = fmaf(d.z, w.z, dh);
          dh = fmaf(d.w, w.w, dh);
        }
      }
    }
  }
  if (!valid) return;
  if (s < n) {
    const int t = dir ? n - 1 - s : s;
    float *gp = p.gates[dir] + ((size_t)b * T + t) * 4 * H + hu;
    const float i = gp[0], g = gp[H], f = gp[2 * H], o = gp[3 * H];
    const float c = p.cs[dir][((size_t)b * T + t) * H + hu];
    float cprev = 0.f;
    if (s > 0) cprev = p.cs[dir][((size_t)b * T + (dir ? t + 1 : t - 1)) * H + hu];
    float *dcp = p.cstate + (size_t)dir * B * H + (size_t)b * H + hu;
    const float tc = tanhf_(c);
    const float dht = p.dout[((size_t)b * T + t) * 2 * H + (size_t)dir * H + hu] + dh;
    const float dct = *dcp + dht * o * (1.f - tc * tc);
    gp[0] = dct * g * i * (1.f - i);
    gp[H] = dct * i * (1.f - g * g);
    gp[2 * H] = dct * cprev * f * (1.f - f);
    gp[3 * H] = dht * tc * o * (1.f - o);
    *dcp = dct * f;
  } else {
    // padded frame s >= len: dz must be 0 for the weight-gradient GEMMs
    float *gp = p.gates[dir] + ((size_t)b * T + s) * 4 * H + hu;
    gp[0] = 0.f; gp[H] = 0.f; gp[2 * H] = 0.f; gp[3 * H] = 0.f;
  }
}

// ---------------------------------------------------------------------------
struct Layout {
  size_t gates_elems, cs_elems;
  size_t reserve_bytes;
  // workspace carve
  size_t hstate_off, cstate_off, gemm_off, gemm_bytes, persist_off, persist_bytes, xws_off, xws_bytes, total;
  // bf16-resident input products (gemm_precision = bf16): converted copies of the operands
  bool bf16_pre, bf16_fwd;
  size_t bf16_off, bf16_bytes;
  // packed bf16-plane operands (gemm_pk.hip): planes = 3 (bf16x6) or 1 (bf16); pk_in = the input products
  // X·Wx, dZ·Wx^T, X^T·dZ, pk_rec = the recurrent weight gradient h_{t-1}^T·dZ
  int pk_planes;
  bool pk_in, pk_rec;
  bool pk_whole;    // narrow input (D < 256): [x^T ; h^T] is ONE operand and the whole kernel gradient one product
  size_t pk_off, pk_bytes;
  // forward: X [BT, D], W^T of both cells [8H, D]; backward: dZ^T [8H, BT], X^T [D, BT], h^T per cell [H, BT],
  // dZ [BT, 8H], Wx of both cells [D, 8H] (byte offsets inside the pk region)
  size_t pk_x, pk_w, pk_xT, pk_hT[2], pk_dz, pk_w2;
  // planes = 2 (f16x3): the row maxima (uint32 per packed row) of the operands above; those of one pass are
  // contiguous (one memset): forward [x | w], backward [dz | w2 | xT | hT0 | hT1]; dZ^T's sit behind it in the reserve
  // the column maxima of x and the row maxima of Wx (the backward pass's x^T and Wx operands) are measured by the
  // forward pass in the same reads as its own and kept in the reserve (res_axT_off, res_aw2_off)
  size_t pk_ax, pk_aw, pk_adz, pk_ahT[2], pk_abwd_bytes, res_adzT_off, res_axT_off, res_aw2_off;
  size_t res_xT_off;         // 0: none.  x^T as a packed operand (the weight-gradient product's), written by the forward
                             // call from the read of x that packs its own operand (one pass over x less per step)
  size_t res_ahT_off;        // 0: none.  The known row bound of h^T (|h| <= 1), written by the forward call's one fill
  // per-workgroup row maxima of dz written by the persistent backward kernel ([2 directions x H / 16][BT] bit patterns;
  // 0 bytes where the layer has no input gradient): the row scales of dZ as [BT, 8H] without a pass over dz
  size_t pk_rowmax, pk_rowmax_bytes;
  bool fwd_only;    // NABU_BLSTM_FWD_ONLY: the reserve ends behind the activations
  // ABI version 3, packed companions (include/nabu_hip.h): x_pk = the input operands come packed from the caller;
  // hT_ext = h^T lives in the caller's hT_pk, written by the forward call; cmp_* = sizes (nabu_blstm_pk_bytes) and a
  // scratch array of row maxima (all 1.0f) for the pack-kernel fallback
  bool x_pk, hT_ext;
  int out_stack;
  size_t cmp_bytes[5];
  size_t cmp_amax_off, cmp_amax_bytes;
  // dZ^T packed [8H, BT] lives in the layer's RESERVE (behind the activations): it is written by the data part of
  // the backward pass and read by the weight-gradient part, which may run later (nabu_blstm_bwd_weights)
  size_t res_dzT_off, res_dzT_bytes;
};

// ABI version 1 callers pass the 32-byte descriptor (everything up to gemm_precision): the later fields read as 0
static int load_desc(const nabu_blstm_desc *in, nabu_blstm_desc *out) {
  constexpr uint32_t V1 = 8 * sizeof(int32_t), V2 = 11 * sizeof(int32_t);
  if (!in || (in->size != sizeof(nabu_blstm_desc) && in->size != V1 && in->size != V2))
    return fail(NABU_EINVAL, "blstm: bad descriptor size");
  *out = nabu_blstm_desc{};
  memcpy(out, in, in->size);
  out->size = sizeof(nabu_blstm_desc);
  if (!(out->x_bound >= 0.f) || out->x_bound > 3.0e38f) return fail(NABU_EINVAL, "blstm: x_bound must be finite and >= 0");
  if (out->flags & ~NABU_BLSTM_FWD_ONLY) return fail(NABU_EINVAL, "blstm: unknown flag bits %d", out->flags);
  if (out->recurrent_precision != NABU_REC_DEFAULT && out->recurrent_precision != NABU_REC_F32)
    return fail(NABU_EINVAL, "blstm: recurrent_precision must be NABU_REC_DEFAULT or NABU_REC_F32");
  if (out->out_stack < 0 || out->out_stack > 2) return fail(NABU_EINVAL, "blstm: out_stack must be 0, 1 or 2");
  if (out->out_stack == 0) out->out_stack = 1;
  if ((out->x_pk_rows == nullptr) != (out->x_pk_cols == nullptr) && !(out->flags & NABU_BLSTM_FWD_ONLY))
    return fail(NABU_EINVAL, "blstm: x_pk_rows and x_pk_cols come as a pair (a forward-only descriptor may give the rows alone)");
  return 0;
}
// every entry point works on the normalised copy and, for its duration, tells the persistent-kernel dispatch whether
// this call asked for the exact-fp32 recurrence (workspace sizes depend on it as well)
struct DescScope {
  nabu_blstm_desc d;
  int err;
  bool prev;
  explicit DescScope(const nabu_blstm_desc *in) : err(load_desc(in, &d)), prev(lstm_persist_exact()) {
    if (!err) lstm_persist_set_exact(d.recurrent_precision == NABU_REC_F32);
  }
  ~DescScope() { lstm_persist_set_exact(prev); }
};

// the input-to-hidden products X·Wx, dZ·Wx^T, X^T·dZ run on bf16 copies of their operands (converted once per
// call, gemm_bf16_pre.hip) when bf16 arithmetic is requested and every reduction length is a multiple of 64
static bool bf16_resident(const nabu_blstm_desc *d) {
  const int prec = d->gemm_precision == NABU_GEMM_DEFAULT ? nabu_gemm_get_default_precision() : d->gemm_precision;
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_BF16_RESIDENT"); env = e ? atoi(e) : 1; }
  const long long BT = (long long)d->B * d->T;
  return env && prec == NABU_GEMM_BF16 && d->D % 64 == 0 && (4 * d->H) % 64 == 0 && BT % 64 == 0 && BT < (1ll << 31);
}
// the forward product alone also takes an input width that is not a multiple of 64 (the first layer: D = 80):
// the bf16 copies of x and Wx^T are zero-padded to the next multiple of 64 along the reduction index
static bool bf16_resident_fwd(const nabu_blstm_desc *d) {
  if (bf16_resident(d)) return true;
  const int prec = d->gemm_precision == NABU_GEMM_DEFAULT ? nabu_gemm_get_default_precision() : d->gemm_precision;
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_BF16_RESIDENT"); env = e ? atoi(e) : 1; }
  const long long BT = (long long)d->B * d->T;
  return env && prec == NABU_GEMM_BF16 && d->D % 8 == 0 && (4 * d->H) % 64 == 0 && BT < (1ll << 31) && BT >= 2048;
}
static int pad64(int x) { return (x + 63) / 64 * 64; }

// planes of the packed-operand path for this layer (0 = not taken): bf16x6 -> 3, f16x3 -> 2, bf16 -> 1
static int pk_planes_of(const nabu_blstm_desc *d) {
  const int prec = d->gemm_precision == NABU_GEMM_DEFAULT ? nabu_gemm_get_default_precision() : d->gemm_precision;
  static int env = -1;
  if (env < 0) { const char *e = getenv("NABU_PK"); env = e ? atoi(e) : 1; }
  if (!env || !gemm_pk_device_ok()) return 0;
  const long long BT = (long long)d->B * d->T;
  if (BT < 1024 || BT >= (1ll << 31) - 512 || d->H % 64) return 0;   // n_split = 4H must be a multiple of 256
  // f16x3 pays for its row maxima with ~20 small launches per layer and step: below 2048 frames (cfg1: 1600) the
  // other fp32-equivalent arithmetic is faster (2.06 against 2.16 ms per cfg1 step)
  if (prec == NABU_GEMM_F16X3) return BT >= 2048 ? 2 : 3;
  return prec == NABU_GEMM_BF16X6 ? 3 : prec == NABU_GEMM_BF16 ? 1 : 0;
}
// the row "maximum" an a-priori bound stands for: the float just below it, so that a power-of-two bound (|h| <= 1) maps
// [0, bound] into (-2^15, 2^15] — 2^15 itself is an fp16 number — instead of giving a bit of range away; 0: no bound
static unsigned bound_bits(float bound) {
  if (!(bound > 0.f)) return 0u;
  unsigned b;
  memcpy(&b, &bound, 4);
  return b > 0x00800000u ? b - 1 : b;
}
// one entry point for the bf16-plane and the scaled-fp16-plane packs (amax: the row maxima of planes = 2)
static int pk_pack_any(int planes, int transposed, const float *src, long long ld, int R, int C, void *dst, int rows_pad,
                       int row_off, int kb_off, int fill_rows, int fill_kb, int period, int shift, const uint32_t *amax,
                       nabu_stream_t stream) {
  if (planes == 2)
    return nabu_pk_pack_f16(transposed, src, ld, R, C, dst, rows_pad, row_off, kb_off, fill_rows, fill_kb, period, shift,
                            amax, stream);
  return nabu_pk_pack(planes, transposed, src, ld, R, C, dst, rows_pad, row_off, kb_off, fill_rows, fill_kb, period, shift,
                      stream);
}
static nabu_pk_gemm_desc pk_desc(int planes, int M, int N, int nkb, const void *A, int a_rows_pad, const void *B,
                                 int b_rows_pad, float *C, int ldc) {
  nabu_pk_gemm_desc g = {};
  g.size = sizeof(g); g.planes = planes; g.M = M; g.N = N; g.nkb = nkb; g.nbatch = 1;
  g.A[0] = A; g.B[0] = B; g.a_rows_pad = a_rows_pad; g.b_rows_pad = b_rows_pad; g.a_planes = g.b_planes = planes;
  g.C[0] = C; g.ldc = ldc; g.alpha = 1.f; g.beta = 0.f;
  if (planes == 2) g.a_amax[0] = g.a_amax[1] = g.b_amax[0] = g.b_amax[1] = reinterpret_cast<const uint32_t *>(16);   // sizing calls
  return g;
}

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

static Layout make_layout(const nabu_blstm_desc *d) {
  Layout L;
  const size_t B = d->B, T = d->T, D = d->D, H = d->H;
  L.gates_elems = B * T * 4 * H;
  L.cs_elems = B * T * H;
  L.reserve_bytes = (2 * L.gates_elems + 2 * L.cs_elems) * sizeof(float);
  L.res_dzT_off = L.res_dzT_bytes = 0;
  L.res_adzT_off = L.res_axT_off = L.res_aw2_off = L.res_ahT_off = L.res_xT_off = 0;
  L.pk_rowmax = L.pk_rowmax_bytes = 0;
  L.fwd_only = (d->flags & NABU_BLSTM_FWD_ONLY) != 0;
  size_t off = 2048;  // ws[0..4): persistent kernels' status word (0 = ok), zeroed by the caller once;
                      // ws[64..64+4*grid): XCC id of every block of the last forward launch (diagnostic)
  L.hstate_off = off; off += align_up(4 * B * H * sizeof(float), 256);
  L.cstate_off = off; off += align_up(2 * B * H * sizeof(float), 256);
  size_t g = 0;
  const int M = (int)(B * T);
  g = max_sz(g, nabu_gemm_ws_bytes(M, (int)(4 * H), (int)D));            // x·Wx
  g = max_sz(g, nabu_gemm_ws_bytes(M, (int)D, (int)(4 * H)));            // dz·Wx^T
  g = max_sz(g, nabu_gemm_ws_bytes((int)D, (int)(4 * H), M));            // x^T·dz
  if (T > 1) g = max_sz(g, nabu_gemm_ws_bytes((int)H, (int)(4 * H), (int)(B * (T - 1))));
  g = max_sz(g, nabu_colsum_ws_bytes(M, (int)(4 * H)));
  if (bf16_resident_fwd(d)) g = max_sz(g, gemm_bf16_pre_ws_bytes(M, (int)(4 * H), pad64((int)D)));
  if (bf16_resident(d)) {
    g = max_sz(g, gemm_bf16_pre_ws_bytes(M, (int)(4 * H), (int)D));
    g = max_sz(g, gemm_bf16_pre_ws_bytes(M, (int)D, (int)(4 * H)));
    g = max_sz(g, gemm_bf16_pre_ws_bytes((int)D, (int)(4 * H), M));
  }
  L.gemm_off = off; L.gemm_bytes = align_up(g, 256); off += L.gemm_bytes;
  L.persist_bytes = align_up(lstm_persist_ws_bytes(d->B, d->T, d->H), 256);
  L.persist_off = off; off += L.persist_bytes;
  // narrow input projected inside the forward kernel (lstm_persist.h): its plane copy of x
  L.xws_bytes = align_up(lstm_persist_xws_bytes(d->B, d->T, d->D, d->H), 256);
  L.xws_off = off; off += L.xws_bytes;
  L.bf16_pre = bf16_resident(d);
  L.bf16_fwd = bf16_resident_fwd(d);
  L.bf16_off = off;
  L.bf16_bytes = 0;
  if (L.bf16_fwd) {
    const size_t BT = B * T, G = 4 * H, Dp = pad64((int)D);
    const size_t fwd = 2 * (BT * Dp + G * Dp), bwd = L.bf16_pre ? 2 * (BT * G + G * BT + D * BT + D * G) : 0;
    L.bf16_bytes = align_up(max_sz(fwd, bwd), 256);
    off += L.bf16_bytes;
  }
  L.pk_planes = pk_planes_of(d);
  L.pk_in = L.pk_planes && D >= 256 && D % 4 == 0;
  L.pk_rec = L.pk_planes && T > 1;
  L.pk_whole = L.pk_rec && !L.pk_in && D % 4 == 0;
  L.pk_off = off; L.pk_bytes = 0;
  if (L.pk_planes) {
    const int P = L.pk_planes, BT = (int)(B * T), G = (int)(4 * H);
    size_t fwd = 0, bwd = 0;
    auto take = [](size_t &o, size_t bytes) { const size_t at = o; o += align_up(bytes, 256); return at; };
    if (L.pk_in) { L.pk_x = take(fwd, nabu_pk_bytes(BT, (int)D, P)); L.pk_w = take(fwd, nabu_pk_bytes(2 * G, (int)D, P)); }
    if (!L.fwd_only) {
      L.res_dzT_off = align_up(L.reserve_bytes, 256);
      L.res_dzT_bytes = nabu_pk_bytes(2 * G, BT, P);
      L.reserve_bytes = L.res_dzT_off + L.res_dzT_bytes;
      if (L.pk_in) {
        L.res_xT_off = align_up(L.reserve_bytes, 256);
        L.reserve_bytes = L.res_xT_off + nabu_pk_bytes((int)D, BT, P);
      }
    }
    L.pk_abwd_bytes = 0;
    if (P == 2) {
      const size_t aBT = 4 * (size_t)nabu_pk_rows_pad(BT), aG = 4 * (size_t)nabu_pk_rows_pad(2 * G), aD = 4 * (size_t)nabu_pk_rows_pad((int)D);
      const size_t aW = 4 * (size_t)nabu_pk_rows_pad((int)(L.pk_whole ? D + H : H));
      if (!L.fwd_only) {
        L.res_adzT_off = align_up(L.reserve_bytes, 256);
        L.res_axT_off = L.res_adzT_off + aG;
        L.res_aw2_off = L.res_axT_off + aD;
        L.reserve_bytes = L.res_aw2_off + aD;
        if (L.pk_in && L.pk_rec) {     // (not the whole-kernel form: its leading rows are measured maxima of x)
          L.res_ahT_off = L.reserve_bytes;
          L.reserve_bytes += aW;
        }
      }
      L.pk_ax = take(fwd, aBT); L.pk_aw = take(fwd, aG);
      L.pk_adz = take(bwd, aBT);
      L.pk_ahT[0] = take(bwd, aW); L.pk_ahT[1] = take(bwd, aW);
      L.pk_abwd_bytes = bwd;
      if (L.pk_in && (size_t)2 * (H / 16) * BT * 4 < 0x80000000ull) {
        L.pk_rowmax_bytes = (size_t)2 * (H / 16) * BT * 4;
        L.pk_rowmax = take(bwd, L.pk_rowmax_bytes);
      }
    }
    if (L.pk_in) L.pk_xT = take(bwd, nabu_pk_bytes((int)D, BT, P));
    for (int dir = 0; dir < 2; ++dir)
      L.pk_hT[dir] = take(bwd, L.pk_rec ? nabu_pk_bytes((int)(L.pk_whole ? D + H : H), BT, P) : 0);
    if (L.pk_in) { L.pk_dz = take(bwd, nabu_pk_bytes(BT, 2 * G, P)); L.pk_w2 = take(bwd, nabu_pk_bytes((int)D, 2 * G, P)); }
    L.pk_bytes = max_sz(fwd, bwd);
    off += L.pk_bytes;
    // split-K slabs of the four products
    size_t gws = 0;
    nabu_pk_gemm_desc g;
    const int rpBT = nabu_pk_rows_pad(BT), rpG = nabu_pk_rows_pad(2 * G), rpD = nabu_pk_rows_pad((int)D);
    float *dummy = reinterpret_cast<float *>(16);
    if (L.pk_in) {
      g = pk_desc(P, BT, 2 * G, nabu_pk_kblocks((int)D, P), dummy, rpBT, dummy, rpG, dummy, G); g.n_split = G; g.C2[0] = dummy;
      gws = max_sz(gws, nabu_gemm_pk_ws_bytes(&g));
      g = pk_desc(P, (int)D, 2 * G, nabu_pk_kblocks(BT, P), dummy, rpD, dummy, rpG, dummy, G); g.n_split = G; g.C2[0] = dummy;
      gws = max_sz(gws, nabu_gemm_pk_ws_bytes(&g));
      g = pk_desc(P, BT, (int)D, nabu_pk_kblocks(2 * G, P), dummy, rpBT, dummy, rpD, dummy, (int)D);
      gws = max_sz(gws, nabu_gemm_pk_ws_bytes(&g));
    }
    if (L.pk_rec) {
      const int Mw = (int)(L.pk_whole ? D + H : H);
      g = pk_desc(P, Mw, G, nabu_pk_kblocks(BT, P), dummy, nabu_pk_rows_pad(Mw), dummy, rpG, dummy, G); g.nbatch = 2;
      g.A[1] = g.B[1] = dummy; g.C[1] = dummy;
      gws = max_sz(gws, nabu_gemm_pk_ws_bytes(&g));
    }
    if (gws > L.gemm_bytes) {   // the gemm region precedes the persist region: grow it in place
      const size_t grow = align_up(gws, 256) - L.gemm_bytes;
      L.gemm_bytes += grow; L.persist_off += grow; L.xws_off += grow; L.bf16_off += grow; L.pk_off += grow; off += grow;
    }
  }
  // packed companions (ABI version 3)
  {
    const int BT = (int)(B * T), S = d->out_stack > 0 ? d->out_stack : 1;
    const bool P2 = L.pk_planes == 2;
    const int Mw = (int)(L.pk_whole ? D + H : H);
    L.out_stack = S;
    L.cmp_bytes[0] = (P2 && L.pk_in) ? nabu_pk_bytes(BT, (int)D, 2) : 0;
    L.cmp_bytes[1] = (P2 && L.pk_in) ? nabu_pk_bytes((int)D, BT, 2) : 0;
    L.cmp_bytes[2] = (P2 && L.pk_rec) ? 2 * nabu_pk_bytes(Mw, BT, 2) : 0;
    L.cmp_bytes[3] = (T % S == 0 && BT / S >= 1) ? nabu_pk_bytes(BT / S, (int)(2 * H) * S, 2) : 0;
    L.cmp_bytes[4] = (T % S == 0 && BT / S >= 1) ? nabu_pk_bytes((int)(2 * H) * S, BT / S, 2) : 0;
    L.x_pk = d->x_pk_rows != nullptr && L.cmp_bytes[0] != 0 && (L.fwd_only || d->x_pk_cols != nullptr);
    L.hT_ext = d->hT_pk != nullptr && L.cmp_bytes[2] != 0 && !L.fwd_only;
    size_t rows = nabu_pk_rows_pad(BT);
    rows = max_sz(rows, nabu_pk_rows_pad((int)(4 * H)));
    rows = max_sz(rows, nabu_pk_rows_pad((int)(D + H)));
    L.cmp_amax_bytes = align_up(4 * rows, 256);
    L.cmp_amax_off = off; off += L.cmp_amax_bytes;
  }
  L.total = off;
  return L;
}

// the row "maximum" of a packed companion: |h| <= 1 at the scale 2^14 the recurrent kernel splits its planes at
static constexpr unsigned CMP_AMAX_BITS = 0x3F800000u;

static int check_desc(const nabu_blstm_desc *d) {
  if (d->B <= 0 || d->T <= 0 || d->D <= 0 || d->H <= 0) return fail(NABU_EINVAL, "blstm: non-positive dimension");
  if (d->H % 4 != 0) return fail(NABU_EUNSUP, "blstm: num_units must be a multiple of 4 (got %d)", d->H);
  if (d->max_len < 0 || d->max_len > d->T) return fail(NABU_EINVAL, "blstm: max_len out of range");
  return 0;
}

// optional profiling hook: caller-owned events recorded around the recurrent kernels
static thread_local hipEvent_t g_ev_begin = nullptr, g_ev_end = nullptr;
#define NABU_PROFILE_MARK(ev, s) do { if (ev) NABU_HIP(hipEventRecord(ev, s)); } while (0)
// optional hook between the recurrent kernel(s) and the dense products of nabu_blstm_bwd
static thread_local nabu_phase_hook_t g_phase_hook = nullptr;
static thread_local void *g_phase_user = nullptr;

static bool use_persistent(const nabu_blstm_desc *d) {
  if (d->mode == NABU_LSTM_STEPWISE) return false;
  return lstm_persist_supported(d->B, d->T, d->H);
}

// WHAT A RESERVE HOLDS is decided by make_layout from the descriptor AND from process state (the default GEMM
// precision, NABU_PK, the device's LDS class): the forward call records a fingerprint of the layout it wrote, keyed by
// the reserve's address, and the backward calls compare it with the layout THEY derive before they touch the buffer —
// a reserve that no forward call produced, or one produced under another layout (precision switched in between, a
// forward-only descriptor), is rejected with NABU_EINVAL instead of being read as something it is not.  Host memory
// only: no device round trip, and the check works (and is tested) without a GPU.
struct ReserveTag {
  const void *reserve;
  uint64_t serial;
  int32_t B, T, D, H, planes, flags, rec;
  uint8_t pk_in, pk_rec, pk_whole, bf16_pre;
  size_t reserve_bytes, res_dzT_off;
};
static std::mutex g_tag_mutex;
static ReserveTag g_tags[4096];   // (least recently written is replaced: a reserve whose tag fell out is rejected, see nabu_hip.h)
static uint64_t g_tag_serial = 0;
static ReserveTag tag_of(const nabu_blstm_desc *d, const Layout &L, const void *reserve) {
  ReserveTag t = {};
  t.reserve = reserve; t.B = d->B; t.T = d->T; t.D = d->D; t.H = d->H; t.planes = L.pk_planes; t.flags = d->flags;
  t.rec = d->recurrent_precision;
  t.pk_in = L.pk_in; t.pk_rec = L.pk_rec; t.pk_whole = L.pk_whole; t.bf16_pre = L.bf16_pre;
  t.reserve_bytes = L.reserve_bytes; t.res_dzT_off = L.res_dzT_off;
  return t;
}
static void tag_store(const ReserveTag &t) {
  std::lock_guard<std::mutex> lock(g_tag_mutex);
  ReserveTag *slot = &g_tags[0];
  for (ReserveTag &e : g_tags) {
    if (e.reserve == t.reserve) { slot = &e; break; }
    if (e.serial < slot->serial) slot = &e;          // least recently written
  }
  *slot = t;
  slot->serial = ++g_tag_serial;
}
static int tag_check(const nabu_blstm_desc *d, const Layout &L, const void *reserve, const char *who) {
  if (L.fwd_only) return fail(NABU_EINVAL, "%s: the descriptor says NABU_BLSTM_FWD_ONLY — its reserve has no room for a backward pass", who);
  const ReserveTag want = tag_of(d, L, reserve);
  std::lock_guard<std::mutex> lock(g_tag_mutex);
  for (const ReserveTag &e : g_tags) {
    if (e.reserve != reserve || !e.serial) continue;
    if (e.B == want.B && e.T == want.T && e.D == want.D && e.H == want.H && e.planes == want.planes && e.flags == want.flags &&
        e.rec == want.rec && e.pk_in == want.pk_in && e.pk_rec == want.pk_rec && e.pk_whole == want.pk_whole && e.bf16_pre == want.bf16_pre &&
        e.reserve_bytes == want.reserve_bytes && e.res_dzT_off == want.res_dzT_off)
      return 0;
    return fail(NABU_EINVAL, "%s: the reserve was written by nabu_blstm_fwd under another layout (B %d T %d D %d H %d, %d planes, "
                "%zu bytes, flags %d; this call: B %d T %d D %d H %d, %d planes, %zu bytes) — descriptor or process precision "
                "changed between the passes", who, e.B, e.T, e.D, e.H, e.planes, e.reserve_bytes, e.flags, want.B, want.T, want.D,
                want.H, want.planes, want.reserve_bytes);
  }
  return fail(NABU_EINVAL, "%s: no nabu_blstm_fwd call of this process wrote this reserve (%p)", who, reserve);
}

}  // namespace nabu

using namespace nabu;

extern "C" int nabu_blstm_set_profile_events(void *ev_begin, void *ev_end) {
  g_ev_begin = static_cast<hipEvent_t>(ev_begin);
  g_ev_end = static_cast<hipEvent_t>(ev_end);
  return 0;
}
extern "C" int nabu_blstm_set_phase_hook(nabu_phase_hook_t fn, void *user) {
  g_phase_hook = fn;
  g_phase_user = user;
  return 0;
}
extern "C" int nabu_persist_set_timeout_us(long long us) {
  nabu::lstm_persist_set_timeout_us(us);
  return 0;
}

extern "C" int nabu_blstm_uses_persistent(const nabu_blstm_desc *d_in) {
  DescScope scope(d_in);
  const nabu_blstm_desc *d = &scope.d;
  if (scope.err || check_desc(d)) return 0;
  return use_persistent(d) ? 1 : 0;
}

extern "C" int nabu_blstm_pk_bytes(const nabu_blstm_desc *d_in, size_t bytes[5]) {
  DescScope scope(d_in);
  if (scope.err) return scope.err;
  const nabu_blstm_desc *d = &scope.d;
  if (int e = check_desc(d)) return e;
  NABU_CHECK_ARG(bytes, "blstm_pk_bytes: null pointer");
  const Layout L = make_layout(d);
  for (int i = 0; i < 5; ++i) bytes[i] = L.cmp_bytes[i];
  return 0;
}
static bool wants_companions(const nabu_blstm_desc *d, const Layout &L) {
  return ((d->out_pk_rows || d->out_pk_cols) && L.cmp_bytes[3] != 0) || L.hT_ext;
}
static bool kernel_emits(const nabu_blstm_desc *d, const Layout &L) {
  const int max_len = d->max_len > 0 ? d->max_len : d->T;
  if (!wants_companions(d, L) || !use_persistent(d) || !lstm_persist_emits(d->B, d->T, d->H, max_len)) return false;
  for (int i = 2; i < 5; ++i)
    if (L.cmp_bytes[i] >= 0x7FFFFFF0ull) return false;      // 32-bit buffer offsets inside the kernel
  return true;
}
// which companions the recurrent kernel writes itself: bit 0 rows, bit 1 transposed, bit 2 h^T.  Default 4: measured on
// cfg2, the 2-byte stores of the transposed operand and the extra live state of the rows cost the forward kernel what the
// pack kernels they replace cost; NABU_PERSIST_EMIT_MASK=7 writes all three from the kernel, 0 none
// (... and not from the launch that also projects its input (the first layer, XIN): there the h^T stores cost 0.09 ms
// per cfg2 step against 0.05 for the pack they replace — per-launch events, LABNOTES.md section 9.  The switch, when set,
// is taken literally.)
static int emit_mask_env(bool *from_env = nullptr) {
  static int m = -1;
  static bool set = false;
  if (m < 0) { const char *e = getenv("NABU_PERSIST_EMIT_MASK"); set = e != nullptr; m = e ? (atoi(e) & 7) : 4; }
  if (from_env) *from_env = set;
  return m;
}
static int emitted_by_kernel(const nabu_blstm_desc *d, const Layout &L) {
  if (!kernel_emits(d, L)) return 0;
  bool from_env = false;
  int m = emit_mask_env(&from_env);
  if (!from_env && lstm_persist_fuses_input(d->B, d->T, d->D, d->H)) m &= ~4;
  const bool want_out = (d->out_pk_rows || d->out_pk_cols) && L.cmp_bytes[3] != 0;
  if (!want_out || !d->out_pk_rows) m &= ~1;
  if (!want_out || !d->out_pk_cols) m &= ~2;
  if (!L.hT_ext) m &= ~4;
  return m;
}
extern "C" int nabu_blstm_emits_packed(const nabu_blstm_desc *d_in) {
  DescScope scope(d_in);
  const nabu_blstm_desc *d = &scope.d;
  if (scope.err || check_desc(d)) return 0;
  return emitted_by_kernel(d, make_layout(d));
}

extern "C" size_t nabu_blstm_reserve_bytes(const nabu_blstm_desc *d_in) {
  DescScope scope(d_in);
  const nabu_blstm_desc *d = &scope.d;
  if (scope.err || check_desc(d)) return 0;
  return make_layout(d).reserve_bytes;
}
extern "C" size_t nabu_blstm_ws_bytes(const nabu_blstm_desc *d_in) {
  DescScope scope(d_in);
  const nabu_blstm_desc *d = &scope.d;
  if (scope.err || check_desc(d)) return 0;
  return make_layout(d).total;
}

extern "C" int nabu_blstm_fwd(const nabu_blstm_desc *d_in, const float *x, const int32_t *len,
                              const float *kernel_fw, const float *bias_fw,
                              const float *kernel_bw, const float *bias_bw, float *out,
                              void *reserve, void *ws, size_t ws_bytes, nabu_stream_t stream) {
  DescScope scope(d_in);
  if (scope.err) return scope.err;
  const nabu_blstm_desc *d = &scope.d;
  if (int e = check_desc(d)) return e;
  NABU_CHECK_ARG(x && len && kernel_fw && bias_fw && kernel_bw && bias_bw && out && reserve && ws,
                 "blstm_fwd: null pointer");
  const Layout L = make_layout(d);
  if (ws_bytes < L.total) return fail(NABU_EWS, "blstm_fwd: workspace %zu < %zu", ws_bytes, L.total);
  if (d->mode == NABU_LSTM_PERSISTENT && !lstm_persist_supported(d->B, d->T, d->H))
    return fail(NABU_EUNSUP, "blstm_fwd: persistent kernel does not support B=%d H=%d", d->B, d->H);
  tag_store(tag_of(d, L, reserve));
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int B = d->B, T = d->T, D = d->D, H = d->H;
  const int max_len = d->max_len > 0 ? d->max_len : T;
  float *r = static_cast<float *>(reserve);
  float *gates[2] = {r, r + L.gates_elems};
  float *cs[2] = {r + 2 * L.gates_elems, r + 2 * L.gates_elems + L.cs_elems};
  char *w = static_cast<char *>(ws);
  const float *kern[2] = {kernel_fw, kernel_bw};
  const float *bias[2] = {bias_fw, bias_bw};

  // narrow input (first layer): the persistent kernel projects its input itself (lstm_persist.hip, XK) — no product here
  const bool fuse_in = use_persistent(d) && lstm_persist_fuses_input(B, T, D, H);
  lstm_persist_ring_cleared(nullptr, s);              // (no note from an earlier call that failed half-way)
  bool ring_with_fill = use_persistent(d) && !fuse_in;   // the projection's fill also clears the exchange ring
  auto input_projection = [&]() -> int {
  // time-batched input projections (MFMA): gates_d = x·Wx_d + b_d
  if (L.pk_in) {
    // packed bf16-plane operands: X once, Wx^T of both cells as the rows of ONE operand; one product fills the
    // gate buffers of both directions
    const int P = L.pk_planes, BT = B * T, G = 4 * H;
    char *pk = w + L.pk_off;
    const int rpBT = nabu_pk_rows_pad(BT), rpG = nabu_pk_rows_pad(2 * G), nkb = nabu_pk_kblocks(D, P);
    uint32_t *ax = reinterpret_cast<uint32_t *>(pk + L.pk_ax), *aw = reinterpret_cast<uint32_t *>(pk + L.pk_aw);
    if (P == 2) {
      // f16x3: the frames' and the gate columns' largest magnitudes first.  The input: from the caller's bound where
      // one is given (x_bound: the previous layer's LSTM outputs — no pass over x), measured otherwise (the features).
      // The maxima the BACKWARD pass needs of the same tensors (columns of x, rows of Wx) come out of the same reads
      // and wait in the reserve — unless no backward pass follows.
      uint32_t *axT = L.fwd_only ? nullptr : reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_axT_off);
      uint32_t *aw2 = L.fwd_only ? nullptr : reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_aw2_off);
      const unsigned xb = L.x_pk ? CMP_AMAX_BITS : bound_bits(d->x_bound);     // (a packed companion: scale 2^14)
      const int rpD = nabu_pk_rows_pad(D);
      // (and the backward pass's row bound of h^T, |h| <= 1: a constant it would otherwise fill in a launch of its own)
      uint32_t *ahT = L.res_ahT_off ? reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_ahT_off) : nullptr;
      // and the recurrent launch's exchange ring (lstm_persist.h: lstm_persist_ring_seg) — nothing between here and that
      // launch writes the persistent kernels' part of the workspace
      FillSeg fill[6] = {{ax, (size_t)rpBT, xb}, {aw, (size_t)rpG, 0u}, {axT, axT ? (size_t)rpD : 0, xb}, {aw2, aw2 ? (size_t)rpD : 0, 0u},
                         {ahT, ahT ? (size_t)nabu_pk_rows_pad(H) : 0, L.hT_ext ? CMP_AMAX_BITS : bound_bits(1.0f)}, {nullptr, 0, 0u}};
      const bool with_ring = ring_with_fill && lstm_persist_ring_seg(true, B, T, H, w + L.persist_off, &fill[5]);
      if (int e = multi_fill(fill, 6, s)) return e;
      if (with_ring) lstm_persist_ring_cleared(&fill[5], s);
      if (!xb)
        if (int e = nabu_pk_amax(x, D, BT, D, ax, axT, stream)) return e;
      if (int e = pk_amax_pair(kern[0], kern[1], G, D, G, aw2, aw, aw + G, nullptr, s)) return e;
    }
    // the input operand: the producer layer's forward kernel wrote it (x_pk_rows) — or one pass over x here
    const void *xop = L.x_pk ? d->x_pk_rows : pk + L.pk_x;
    if (!L.x_pk && L.res_xT_off) {
      // ... and x^T for the backward pass's weight-gradient product out of the same read (kept in the reserve)
      const int rpDx = nabu_pk_rows_pad(D);
      if (int e = pk_pack_both(P, x, D, BT, D, pk + L.pk_x, rpBT, 0, rpBT, nkb, static_cast<char *>(reserve) + L.res_xT_off, rpDx, 0,
                               rpDx, nabu_pk_kblocks(BT, P), s, P == 2 ? ax : nullptr,
                               P == 2 ? reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_axT_off) : nullptr))
        return e;
    } else if (!L.x_pk) {
      if (int e = pk_pack_any(P, 0, x, D, BT, D, pk + L.pk_x, rpBT, 0, 0, rpBT, nkb, 0, 0, ax, stream)) return e;
    }
    {   // Wx^T of both cells: one launch
      PkPackReq rq[2];
      for (int dir = 0; dir < 2; ++dir)
        rq[dir] = PkPackReq{kern[dir], G, D, G, pk + L.pk_w, rpG, dir * G, 0, dir ? rpG - G : G, nkb, 0, 0, P == 2 ? aw : nullptr};
      if (int e = pk_pack_multi(P, 1, rq, 2, s)) return e;
    }
    nabu_pk_gemm_desc g = pk_desc(P, BT, 2 * G, nkb, xop, rpBT, pk + L.pk_w, rpG, gates[0], G);
    g.C2[0] = gates[1]; g.n_split = G; g.bias = bias[0]; g.bias2 = bias[1];
    if (P == 2) { g.a_amax[0] = ax; g.b_amax[0] = aw; g.direct = 2; }
    if (int e = nabu_gemm_pk(&g, w + L.gemm_off, L.gemm_bytes, stream)) return e;
  } else if (L.bf16_fwd) {
    // bf16 copies: x once, Wx_d transposed ([4H, Dp]: the reduction index contiguous, zero-padded to a
    // multiple of 64), then 2-byte operands
    const int Dp = pad64(D);
    unsigned short *xb = reinterpret_cast<unsigned short *>(w + L.bf16_off);
    unsigned short *wt = xb + (size_t)B * T * Dp;
    if (Dp != D) NABU_HIP(hipMemsetAsync(xb, 0, ((size_t)B * T + 4 * H) * Dp * 2, s));
    if (int e = cvt_bf16((size_t)B * T, D, x, D, xb, Dp, s)) return e;
    for (int dir = 0; dir < 2; ++dir) {
      if (int e = cvt_bf16_t(D, 4 * H, kern[dir], 4 * H, wt, Dp, s)) return e;
      if (int e = gemm_bf16_pre(B * T, 4 * H, Dp, 1.f, xb, Dp, wt, Dp, 0.f, gates[dir], 4 * H, bias[dir], w + L.gemm_off,
                                L.gemm_bytes, s))
        return e;
    }
  } else
  for (int dir = 0; dir < 2; ++dir) {
    int e = nabu_gemm_ex(d->gemm_precision, 0, 0, B * T, 4 * H, D, 1.f, x, D, kern[dir], 4 * H, 0.f, gates[dir],
                          4 * H, bias[dir], 0, 0, 0, w + L.gemm_off, L.gemm_bytes, stream);
    if (e) return e;
  }
  return 0;
  };
  if (!fuse_in)
    if (int e = input_projection()) return e;
  // frames t in [max_len, T) are never visited by the recurrence
  if (max_len < T)
    NABU_HIP(hipMemset2DAsync(out + (size_t)max_len * 2 * H, (size_t)T * 2 * H * sizeof(float), 0,
                              (size_t)(T - max_len) * 2 * H * sizeof(float), B, s));

  // packed companions of the output (ABI version 3): out of the recurrent kernel itself where that is possible, by the
  // pack kernels behind the recurrence otherwise — complete on return either way
  const bool want_cmp = wants_companions(d, L);
  const int S = L.out_stack, rpW = nabu_pk_rows_pad(L.pk_whole ? D + H : H), r0 = L.pk_whole ? D : 0;
  char *hTp[2] = {nullptr, nullptr};
  if (L.hT_ext) { hTp[0] = static_cast<char *>(d->hT_pk); hTp[1] = hTp[0] + L.cmp_bytes[2] / 2; }
  const bool want_out = (d->out_pk_rows || d->out_pk_cols) && L.cmp_bytes[3] != 0;
  const int by_kernel = emitted_by_kernel(d, L);
  const bool emit = by_kernel != 0;
  auto companions_by_pack_kernels = [&](int todo) -> int {
    todo &= ((want_out && d->out_pk_rows) ? 1 : 0) | ((want_out && d->out_pk_cols) ? 2 : 0) | (L.hT_ext ? 4 : 0);
    if (!todo) return 0;
    uint32_t *am = reinterpret_cast<uint32_t *>(w + L.cmp_amax_off);
    const FillSeg fill = {am, L.cmp_amax_bytes / 4, CMP_AMAX_BITS};
    if (int e = multi_fill(&fill, 1, s)) return e;
    const int R = B * T / S, C = 2 * H * S;
    if (want_out && d->out_pk_rows && (todo & 1))
      if (int e = pk_pack_any(2, 0, out, C, R, C, d->out_pk_rows, nabu_pk_rows_pad(R), 0, 0, nabu_pk_rows_pad(R), nabu_pk_kblocks(C, 2), 0, 0, am, stream)) return e;
    if (want_out && d->out_pk_cols && (todo & 2))
      if (int e = pk_pack_any(2, 1, out, C, R, C, d->out_pk_cols, nabu_pk_rows_pad(C), 0, 0, nabu_pk_rows_pad(C), nabu_pk_kblocks(R, 2), 0, 0, am, stream)) return e;
    if (L.hT_ext && (todo & 4)) {
      PkPackReq rq[2];
      for (int dir = 0; dir < 2; ++dir)
        rq[dir] = PkPackReq{out + (size_t)dir * H, 2 * H, B * T, H, hTp[dir], rpW, r0, 0, rpW - r0, nabu_pk_kblocks(B * T, 2), T, dir ? 1 : -1, am};
      if (int e = pk_pack_multi(2, 1, rq, 2, s)) return e;
    }
    return 0;
  };
  EmitArgs em = {};
  if (emit) {
    em.x_rows = (want_out && (by_kernel & 1)) ? static_cast<char *>(d->out_pk_rows) : nullptr;
    em.x_cols = (want_out && (by_kernel & 2)) ? static_cast<char *>(d->out_pk_cols) : nullptr;
    em.hT[0] = (by_kernel & 4) ? hTp[0] : nullptr; em.hT[1] = (by_kernel & 4) ? hTp[1] : nullptr;
    em.x_rows_pad = (unsigned)nabu_pk_rows_pad(B * T / S);
    em.x_cols_pad = (unsigned)nabu_pk_rows_pad(2 * H * S);
    em.hT_rows_pad = (unsigned)rpW;
    em.hT_row0 = r0;
    em.stack_shift = S == 2 ? 1 : 0;
    em.b0 = 0;
  }

  if (use_persistent(d)) {
    NABU_PROFILE_MARK(g_ev_begin, s);
    int e = lstm_persist_fwd(B, T, D, H, max_len, len, kern, gates, cs, out, reinterpret_cast<int *>(w), w + L.persist_off,
                             L.persist_bytes, s, fuse_in ? x : nullptr, fuse_in ? bias : nullptr,
                             L.xws_bytes ? w + L.xws_off : nullptr, emit ? &em : nullptr);
    // the grid cannot be co-resident on this device (occupancy check before the launch): LSTM_AUTO steps instead
    if (!(e == NABU_EUNSUP && d->mode == NABU_LSTM_AUTO)) {
      if (e) return e;
      NABU_PROFILE_MARK(g_ev_end, s);
      if (want_cmp) return companions_by_pack_kernels(7 & ~by_kernel);
      return 0;
    }
    lstm_persist_ring_cleared(nullptr, s);
    ring_with_fill = false;
    if (fuse_in)      // the step kernels read the projection from the gate buffers
      if (int e2 = input_projection()) return e2;
  }
  StepArgs p;
  p.B = B; p.T = T; p.D = D; p.H = H; p.max_len = max_len; p.len = len;
  for (int i = 0; i < 2; ++i) { p.kernel[i] = kern[i]; p.gates[i] = gates[i]; p.cs[i] = cs[i]; }
  p.out = out; p.dout = nullptr;
  p.hstate = reinterpret_cast<float *>(w + L.hstate_off);
  p.cstate = reinterpret_cast<float *>(w + L.cstate_off);
  NABU_HIP(hipMemsetAsync(p.hstate, 0, 4 * (size_t)B * H * sizeof(float), s));
  NABU_HIP(hipMemsetAsync(p.cstate, 0, 2 * (size_t)B * H * sizeof(float), s));
  const dim3 grid((H + SU - 1) / SU, (B + SB - 1) / SB, 2);
  const size_t shm = ((size_t)SB * H + SB * 4 * SU) * sizeof(float);
  if (shm > 64 * 1024)
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(lstm_step_fwd_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  NABU_PROFILE_MARK(g_ev_begin, s);
  for (int t = 0; t < max_len; ++t) {
    hipLaunchKernelGGL(lstm_step_fwd_kernel, grid, dim3(256), shm, s, p, t);
  }
  NABU_LAUNCH_CHECK();
  NABU_PROFILE_MARK(g_ev_end, s);
  if (want_cmp) return companions_by_pack_kernels(7);
  return 0;
}

// parts: 1 = data (recurrence backward, bias gradients, input gradient, the packs of dz), 2 = weights (dWx, dWh from
// the dz the data part left in the reserve), 3 = both (nabu_blstm_bwd)
static int blstm_bwd_parts(int parts, const nabu_blstm_desc *d, const float *x, const int32_t *len,
                           const float *kernel_fw, const float *kernel_bw, const float *out,
                           const float *d_out, void *reserve, float *d_x, float *dkernel_fw,
                           float *dbias_fw, float *dkernel_bw, float *dbias_bw, void *ws,
                           size_t ws_bytes, nabu_stream_t stream) {
  if (int e = check_desc(d)) return e;
  NABU_CHECK_ARG(x && len && out && reserve && ws, "blstm_bwd: null pointer");
  if (parts & 1) NABU_CHECK_ARG(kernel_fw && kernel_bw && d_out && dbias_fw && dbias_bw, "blstm_bwd: null pointer");
  if (parts & 2) NABU_CHECK_ARG(dkernel_fw && dkernel_bw, "blstm_bwd: null pointer");
  const Layout L = make_layout(d);
  if (int e = tag_check(d, L, reserve, parts == 3 ? "blstm_bwd" : parts == 1 ? "blstm_bwd_data" : "blstm_bwd_weights")) return e;
  if (ws_bytes < L.total) return fail(NABU_EWS, "blstm_bwd: workspace %zu < %zu", ws_bytes, L.total);
  if (d->mode == NABU_LSTM_PERSISTENT && !lstm_persist_supported(d->B, d->T, d->H))
    return fail(NABU_EUNSUP, "blstm_bwd: persistent kernel does not support B=%d H=%d", d->B, d->H);
  hipStream_t s = static_cast<hipStream_t>(stream);
  lstm_persist_ring_cleared(nullptr, s);
  const int B = d->B, T = d->T, D = d->D, H = d->H;
  const int max_len = d->max_len > 0 ? d->max_len : T;
  float *r = static_cast<float *>(reserve);
  float *gates[2] = {r, r + L.gates_elems};
  float *cs[2] = {r + 2 * L.gates_elems, r + 2 * L.gates_elems + L.cs_elems};
  char *w = static_cast<char *>(ws);
  const float *kern[2] = {kernel_fw, kernel_bw};
  float *dkern[2] = {dkernel_fw, dkernel_bw};
  float *dbias[2] = {dbias_fw, dbias_bw};

  float *db_part = nullptr;   // persistent path: bias-gradient partials [db_rows][2][4H]
  int db_rows = 0;
  bool db_done = false;       // the bias gradients were summed by the launch that read the maxima
  // f16x3 with an input gradient: the persistent kernel leaves every workgroup's row maxima of dz in the workspace
  uint32_t *rowmax = (L.pk_planes == 2 && L.pk_in && d_x && L.pk_rowmax_bytes)
                         ? reinterpret_cast<uint32_t *>(w + L.pk_off + L.pk_rowmax) : nullptr;
  bool rowmax_done = false;
  if (parts & 1) {
  // dz rows of frames never visited by the recurrence must be zero
  if (max_len < T)
    for (int dir = 0; dir < 2; ++dir)
      NABU_HIP(hipMemset2DAsync(gates[dir] + (size_t)max_len * 4 * H, (size_t)T * 4 * H * sizeof(float),
                                0, (size_t)(T - max_len) * 4 * H * sizeof(float), B, s));

  NABU_PROFILE_MARK(g_ev_begin, s);
  bool stepwise = !use_persistent(d);
  if (!stepwise) {
    int e = lstm_persist_bwd(B, T, D, H, max_len, len, kern, gates, cs, d_out, reinterpret_cast<int *>(w), w + L.persist_off,
                             L.persist_bytes, &db_part, &db_rows, s, rowmax, &rowmax_done);
    if (e == NABU_EUNSUP && d->mode == NABU_LSTM_AUTO) { stepwise = true; db_part = nullptr; db_rows = 0; rowmax_done = false; }
    else if (e) return e;
  }
  if (stepwise) {
    StepArgs p;
    p.B = B; p.T = T; p.D = D; p.H = H; p.max_len = max_len; p.len = len;
    for (int i = 0; i < 2; ++i) { p.kernel[i] = kern[i]; p.gates[i] = gates[i]; p.cs[i] = cs[i]; }
    p.out = nullptr; p.dout = d_out;
    p.hstate = nullptr;
    p.cstate = reinterpret_cast<float *>(w + L.cstate_off);  // dc carry
    NABU_HIP(hipMemsetAsync(p.cstate, 0, 2 * (size_t)B * H * sizeof(float), s));
    const dim3 grid((H + SU - 1) / SU, (B + SB - 1) / SB, 2);
    for (int t = max_len - 1; t >= 0; --t)
      hipLaunchKernelGGL(lstm_step_bwd_kernel, grid, dim3(256), 0, s, p, t);
    NABU_LAUNCH_CHECK();
  }
  NABU_PROFILE_MARK(g_ev_end, s);
  if (g_phase_hook) g_phase_hook(g_phase_user);
  }

  // weight / input gradients from dz (now stored in gates[])
  const int M = B * T;
  if (L.pk_planes) {
    // packed bf16-plane operands (gemm_pk.hip).  dZ^T of both cells is one operand [8H, BT] (in the reserve): the
    // weight gradients of both cells are column ranges of one product (input part) resp. a batch of two (recurrent part)
    const int P = L.pk_planes, G = 4 * H;
    char *pk = w + L.pk_off;
    char *dzTp = static_cast<char *>(reserve) + L.res_dzT_off;
    const int rpBT = nabu_pk_rows_pad(M), rpG = nabu_pk_rows_pad(2 * G), rpD = nabu_pk_rows_pad(D);
    const int nkbT = nabu_pk_kblocks(M, P);
    const int nkb2 = nabu_pk_kblocks(2 * G, P), kbG = G / 16;
    int e;
    // f16x3 (P = 2): row maxima of every operand — measured (dz, the weights, x) or known (|h| < 1)
    uint32_t *adz = reinterpret_cast<uint32_t *>(pk + L.pk_adz);
    uint32_t *axT = reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_axT_off);    // from the forward pass
    uint32_t *aw2 = reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_aw2_off);
    uint32_t *ahT[2] = {reinterpret_cast<uint32_t *>(pk + L.pk_ahT[0]), reinterpret_cast<uint32_t *>(pk + L.pk_ahT[1])};
    uint32_t *adzT = reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_adzT_off);
    if (parts & 1) {
      const bool both = d_x && L.pk_in;    // dz is also needed row-major (dx): both packs from one read of dz
      if (P == 2) {
        // the maxima of dz: its rows' over BOTH cells (the row scale of dZ as [BT, 8H]) and its columns'
        if (db_part && (!both || rowmax_done)) {
          // the persistent kernel kept them: the gate columns' maxima per unit next to its bias-gradient partials, the
          // frames' per workgroup in the workspace (only asked for where an input gradient follows) — no read of dz
          // (the bias gradients — the sum of the same units' partial rows — come out of the same launch)
          if ((e = pk_amax_from_persist(M, rpBT, T, max_len, 2 * (H / 16), rowmax, both ? adz : nullptr, db_rows, 2 * G,
                                        db_part + lstm_persist_db_floats(B, H), 2 * G, adzT, s, db_part, dbias[0], dbias[1])))
            return e;
          db_done = true;
        } else {
          const FillSeg fill[2] = {{adz, both ? (size_t)rpBT : 0, 0u}, {adzT, (size_t)rpG, 0u}};
          if ((e = multi_fill(fill, 2, s))) return e;
          if ((e = pk_amax_pair(gates[0], gates[1], G, M, G, both ? adz : nullptr, adzT, adzT + G, nullptr, s))) return e;
        }
      }
      for (int dir = 0; dir < 2; ++dir) {
        if (both)
          e = pk_pack_both(P, gates[dir], G, M, G, pk + L.pk_dz, rpBT, dir * kbG, rpBT, dir ? nkb2 - kbG : kbG, dzTp,
                           rpG, dir * G, dir ? rpG - G : G, nkbT, s, adz, adzT);
        else
          e = pk_pack_any(P, 1, gates[dir], G, M, G, dzTp, rpG, dir * G, 0, dir ? rpG - G : G, nkbT, 0, 0, adzT, stream);
        if (e) return e;
      }
      if (d_x && L.pk_in) {
        // dx = [dZ_fw | dZ_bw] · [Wx_fw | Wx_bw]^T: the two cells are two ranges of ONE reduction
        PkPackReq rq[2];
        for (int dir = 0; dir < 2; ++dir)
          rq[dir] = PkPackReq{kern[dir], G, D, G, pk + L.pk_w2, rpD, 0, dir * kbG, rpD, dir ? nkb2 - kbG : kbG, 0, 0,
                              P == 2 ? aw2 : nullptr};
        if ((e = pk_pack_multi(P, 0, rq, 2, s))) return e;
        nabu_pk_gemm_desc g = pk_desc(P, M, D, nkb2, pk + L.pk_dz, rpBT, pk + L.pk_w2, rpD, d_x, D);
        if (P == 2) { g.a_amax[0] = adz; g.b_amax[0] = aw2; g.direct = 2; }
        if ((e = nabu_gemm_pk(&g, w + L.gemm_off, L.gemm_bytes, stream))) return e;
      }
    }
    if ((parts & 2) && L.pk_in) {
      // x^T: the producer layer's forward kernel wrote it (x_pk_cols; its row maxima were set by this layer's forward
      // call) — or one transposing pass over x here
      // (... or, since the forward call packs x anyway, from that call: res_xT_off)
      const void *xTop = L.x_pk ? d->x_pk_cols : L.res_xT_off ? static_cast<const char *>(reserve) + L.res_xT_off : pk + L.pk_xT;
      if (!L.x_pk && !L.res_xT_off)
        if ((e = pk_pack_any(P, 1, x, D, M, D, pk + L.pk_xT, rpD, 0, 0, rpD, nkbT, 0, 0, axT, stream))) return e;
      nabu_pk_gemm_desc g = pk_desc(P, D, 2 * G, nkbT, xTop, rpD, dzTp, rpG, dkern[0], G);
      g.C2[0] = dkern[1]; g.n_split = G;
      // direct = 2: the three plane products chained directly into the accumulators wherever that rounds less often
      // than the exact-fp32 kernel would (gemm_pk.hip; 0.6-0.8 x its error at these shapes, tests/test_hip_gemm_pk.py)
      if (P == 2) { g.a_amax[0] = axT; g.b_amax[0] = adzT; g.direct = 2; }
      if ((e = nabu_gemm_pk(&g, w + L.gemm_off, L.gemm_bytes, stream))) return e;
    }
    if ((parts & 2) && L.pk_rec) {
      // h_{t-1}^T: the forward cell pairs dz[b,t] with out[b,t-1,:H], the backward cell with out[b,t+1,H:].
      // Narrow input (the first layer, D = 40): x^T sits in front of h^T in the same operand and the whole kernel
      // gradient [(D+H), 4H] of a cell is ONE product (its dWx alone cost more on the in-kernel-split kernel)
      const int r0 = L.pk_whole ? D : 0, Mw = r0 + H, rpW = nabu_pk_rows_pad(Mw);
      // h^T: in the caller's hT_pk, written by the forward call (ABI version 3) — or packed here from `out`
      char *hTb[2] = {pk + L.pk_hT[0], pk + L.pk_hT[1]};
      if (L.hT_ext) { hTb[0] = static_cast<char *>(d->hT_pk); hTb[1] = hTb[0] + L.cmp_bytes[2] / 2; }
      if (P == 2 && L.res_ahT_off) {   // |h| <= 1: the bound sits in the reserve since the forward call (both cells share it)
        ahT[0] = ahT[1] = reinterpret_cast<uint32_t *>(static_cast<char *>(reserve) + L.res_ahT_off);
      } else if (P == 2) {   // |h| <= 1 by construction (o · tanh c): one fill for both cells; the input features are measured
        const unsigned hb = L.hT_ext ? CMP_AMAX_BITS : bound_bits(1.0f);
        const FillSeg fill[4] = {{ahT[0], (size_t)r0, 0u}, {ahT[0] + r0, (size_t)(rpW - r0), hb},
                                 {ahT[1], (size_t)r0, 0u}, {ahT[1] + r0, (size_t)(rpW - r0), hb}};
        // (r0 = D is a multiple of 4: every region starts 16-byte aligned)
        if ((e = multi_fill(fill, 4, s))) return e;
        if (L.pk_whole && (e = pk_amax_pair(x, nullptr, D, M, D, nullptr, ahT[0], nullptr, ahT[1], s))) return e;
      }
      {   // [x^T ;] h^T of both cells: one launch
        PkPackReq rq[4];
        int n = 0;
        for (int dir = 0; dir < 2; ++dir) {
          const uint32_t *am = P == 2 ? ahT[dir] : nullptr;
          if (L.pk_whole) rq[n++] = PkPackReq{x, D, M, D, hTb[dir], rpW, 0, 0, D, nkbT, 0, 0, am};
          if (!L.hT_ext)
            rq[n++] = PkPackReq{out + (size_t)dir * H, 2 * H, M, H, hTb[dir], rpW, r0, 0, rpW - r0, nkbT, T, dir ? 1 : -1, am};
        }
        if (n && (e = pk_pack_multi(P, 1, rq, n, s))) return e;
      }
      nabu_pk_gemm_desc g = pk_desc(P, Mw, G, nkbT, hTb[0], rpW, dzTp, rpG, dkern[0] + (size_t)(D - r0) * G, G);
      g.nbatch = 2; g.A[1] = hTb[1]; g.B[1] = dzTp + (size_t)G * 32; g.C[1] = dkern[1] + (size_t)(D - r0) * G;
      if (P == 2) { g.a_amax[0] = ahT[0]; g.a_amax[1] = ahT[1]; g.b_amax[0] = adzT; g.b_amax[1] = adzT + G; g.direct = 2; }
      if ((e = nabu_gemm_pk(&g, w + L.gemm_off, L.gemm_bytes, stream))) return e;
    }
  }
  unsigned short *dzb = nullptr, *dzT = nullptr, *xT = nullptr, *wb = nullptr;
  const bool old_bf16 = L.bf16_pre && !L.pk_in;
  if (old_bf16) {   // bf16 copies of this call's operands: x^T once; dz and dz^T, Wx per direction
    dzb = reinterpret_cast<unsigned short *>(w + L.bf16_off);
    dzT = dzb + (size_t)M * 4 * H;
    xT = dzT + (size_t)4 * H * M;
    wb = xT + (size_t)D * M;
    if (parts & 2)
      if (int e = cvt_bf16_t(M, D, x, D, xT, M, s)) return e;
  }
  for (int dir = 0; dir < 2; ++dir) {
    int e = 0;
    if (L.pk_in || L.pk_whole || !(parts & 2)) {
    } else if (old_bf16) {
      if ((e = cvt_bf16_t(M, 4 * H, gates[dir], 4 * H, dzT, M, s))) return e;
      // dWx = x^T · dz = sum over frames of xT[d, k] * dzT[n, k]
      if ((e = gemm_bf16_pre(D, 4 * H, M, 1.f, xT, M, dzT, M, 0.f, dkern[dir], 4 * H, nullptr, w + L.gemm_off,
                             L.gemm_bytes, s)))
        return e;
    } else {
    // dWx = x^T · dz
    e = nabu_gemm_ex(d->gemm_precision, 1, 0, D, 4 * H, M, 1.f, x, D, gates[dir], 4 * H, 0.f, dkern[dir], 4 * H,
                      nullptr, 0, 0, 0, w + L.gemm_off, L.gemm_bytes, stream);
    if (e) return e;
    }
    // dWh = h_{prev}^T · dz : fw pairs (out[b,t-1,:H], dz[b,t]); bw pairs (out[b,t+1,H:], dz[b,t])
    if (!(L.pk_planes && L.pk_rec) && (parts & 2)) {
    const float *A = dir == 0 ? out : out + H + (size_t)2 * H;
    const float *Bm = dir == 0 ? gates[0] + (size_t)4 * H : gates[1];
    e = nabu_gemm_f32(1, 0, H, 4 * H, B * (T - 1), 1.f, A, 2 * H, Bm, 4 * H, 0.f,
                      dkern[dir] + (size_t)D * 4 * H, 4 * H, nullptr, T > 1 ? T - 1 : 0,
                      (long long)T * 2 * H, (long long)T * 4 * H, w + L.gemm_off, L.gemm_bytes, stream);
    if (e) return e;
    }
    if (!(parts & 1)) continue;
    // db = column sums of dz: the persistent kernel already summed them per unit (one launch adds the few partial
    // rows of both cells)
    if (db_part)
      e = (dir || db_done) ? 0 : colsum_pair(db_rows, 4 * H, db_part, 2 * 4 * H, dbias[0], dbias[1], s);
    else
      e = nabu_colsum_f32(M, 4 * H, gates[dir], 4 * H, 0.f, dbias[dir], w + L.gemm_off, L.gemm_bytes, stream);
    if (e) return e;
    // dx (+)= dz · Wx^T
    if (L.pk_in) {
    } else if (d_x && old_bf16) {
      if ((e = cvt_bf16((size_t)M, 4 * H, gates[dir], 4 * H, dzb, 4 * H, s))) return e;
      if ((e = cvt_bf16((size_t)D, 4 * H, kern[dir], 4 * H, wb, 4 * H, s))) return e;
      if ((e = gemm_bf16_pre(M, D, 4 * H, 1.f, dzb, 4 * H, wb, 4 * H, dir == 0 ? 0.f : 1.f, d_x, D, nullptr,
                             w + L.gemm_off, L.gemm_bytes, s)))
        return e;
    } else if (d_x) {
      e = nabu_gemm_ex(d->gemm_precision, 0, 1, M, D, 4 * H, 1.f, gates[dir], 4 * H, kern[dir], 4 * H,
                        dir == 0 ? 0.f : 1.f, d_x, D, nullptr, 0, 0, 0, w + L.gemm_off, L.gemm_bytes, stream);
      if (e) return e;
    }
  }
  return 0;
}

extern "C" int nabu_blstm_bwd(const nabu_blstm_desc *d_in, const float *x, const int32_t *len,
                              const float *kernel_fw, const float *kernel_bw, const float *out,
                              const float *d_out, void *reserve, float *d_x, float *dkernel_fw,
                              float *dbias_fw, float *dkernel_bw, float *dbias_bw, void *ws,
                              size_t ws_bytes, nabu_stream_t stream) {
  DescScope scope(d_in);
  if (scope.err) return scope.err;
  const nabu_blstm_desc *d = &scope.d;
  return blstm_bwd_parts(3, d, x, len, kernel_fw, kernel_bw, out, d_out, reserve, d_x, dkernel_fw, dbias_fw, dkernel_bw,
                         dbias_bw, ws, ws_bytes, stream);
}
extern "C" int nabu_blstm_bwd_data(const nabu_blstm_desc *d_in, const float *x, const int32_t *len,
                                   const float *kernel_fw, const float *kernel_bw, const float *out,
                                   const float *d_out, void *reserve, float *d_x, float *dbias_fw, float *dbias_bw,
                                   void *ws, size_t ws_bytes, nabu_stream_t stream) {
  DescScope scope(d_in);
  if (scope.err) return scope.err;
  const nabu_blstm_desc *d = &scope.d;
  return blstm_bwd_parts(1, d, x, len, kernel_fw, kernel_bw, out, d_out, reserve, d_x, nullptr, dbias_fw, nullptr, dbias_bw,
                         ws, ws_bytes, stream);
}
extern "C" int nabu_blstm_bwd_weights(const nabu_blstm_desc *d_in, const float *x, const int32_t *len, const float *out,
                                      void *reserve, float *dkernel_fw, float *dkernel_bw, void *ws, size_t ws_bytes,
                                      nabu_stream_t stream) {
  DescScope scope(d_in);
  if (scope.err) return scope.err;
  const nabu_blstm_desc *d = &scope.d;
  return blstm_bwd_parts(2, d, x, len, nullptr, nullptr, out, nullptr, reserve, nullptr, dkernel_fw, nullptr, dkernel_bw,
                         nullptr, ws, ws_bytes, stream);
}
