// gemm_skinny.hip — C[M,N] = A[M,K] · B[K,N] for M <= 64 (the per-step products of the Speller
// decoder: 32..64 batch rows against a [K,N] weight matrix).  A 128x128 tile kernel wastes 3/4 of
// its MFMAs on such a product and occupies 16 of 256 CUs; it is bound by streaming B once.
//
// Grid = (N/32 column slices) x (K/KC k-chunks): every workgroup streams a [KC x 32] block of B
// exactly once with 128-byte coalesced rows, against A^T[KC x M] staged in LDS (k-major, so an
// MFMA operand is 32 consecutive floats).  The 4 waves split the k-chunk, their partial 32x32
// tiles meet in LDS; k-chunks are summed by the deterministic split-K reduce kernel of gemm.hip.
// Exact fp32 (v_mfma_f32_32x32x2_f32).
#include "gemm_args.h"

#include <stdlib.h>

namespace nabu {

// The tile product shared by both kernels: acc (per wave, over its quarter of the k-chunk) of
// A[M, k0:k0+KC] · B[k0:k0+KC, n0:n0+32].  Every global load of the workgroup — the wave's rows of B (one
// float per lane and row, 128-byte row segments) and the A chunk — is issued BEFORE anything waits, so
// the memory latency is paid once per workgroup instead of once per batch of 8 rows (which made a
// 32 KB stream take ~8 us).
template <int MT>
__device__ __forceinline__ void skinny_tile(const float *__restrict__ Ap, int lda, const float *__restrict__ Bp, int ldb,
                                            int k0, int KC, int M, int n0, float *sm, f32x16 (&acc)[MT]) {
  constexpr int MR = 32 * MT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kw = KC / 4;                              // k values of this wave: 16, 32 or 64
  const int li = lane & 31, lk = lane >> 5;
  const float *bp = Bp + (size_t)(k0 + w * kw + lk) * ldb + n0 + li;
  const size_t bstep = 2 * (size_t)ldb;
  float b[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) b[j] = (2 * j < kw) ? bp[(size_t)j * bstep] : 0.f;
  // A^T chunk -> LDS xT[k][m]
  for (int idx = tid; idx < (KC / 4) * MR; idx += 256) {
    const int m = idx % MR, kq = idx / MR;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) v = *reinterpret_cast<const float4 *>(Ap + (size_t)m * lda + k0 + 4 * kq);
    float *d = sm + (size_t)(4 * kq) * MR + m;
    d[0] = v.x; d[MR] = v.y; d[2 * MR] = v.z; d[3 * MR] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float *xp = sm + (size_t)(w * kw + lk) * MR + li;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    if (2 * j < kw) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xp[(size_t)(2 * j) * MR + 32 * t], b[j], acc[t], 0, 0, 0);
    }
  }
  __syncthreads();                                    // xT is dead: the callers reuse LDS for the wave partials
}

template <int MT>   // row tiles of 32
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 32;
  f32x16 acc[MT];
  skinny_tile<MT>(a.A, a.lda, a.B, a.ldb, blockIdx.y * a.ksplit, a.ksplit, a.M, n0, sm, acc);
  float *red = sm;                                    // [4 waves][MT][16][64]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((w * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  for (int e = tid; e < MT * 16 * 64; e += 256) {
    const float s = red[e] + red[MT * 1024 + e] + red[2 * MT * 1024 + e] + red[3 * MT * 1024 + e];
    const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
    const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = n0 + (l & 31);
    if (m >= a.M) continue;
    if (a.nsplit == 1) {
      float *c = a.C + (size_t)m * a.ldc + n;
      float v = a.alpha * s + (a.bias ? a.bias[n] : 0.f);
      if (a.beta != 0.f) v += a.beta * *c;
      *c = v;
    } else {
      a.partial[((size_t)blockIdx.y * a.M + m) * a.N + n] = s;
    }
  }
}

// ---------------------------------------------------------------------------
// Decoder-step variant: the same product with (1) the reduction index running over TWO operand
// pairs, C = [A | A2] · [B ; B2] (the Speller cell's [context, h]·kernel without a concatenated copy),
// and (2) the split-K reduction finished INSIDE the launch: every workgroup writes its partial tile
// through to memory, takes a ticket of its column slice, and the workgroup that draws the last
// ticket sums the slice's partials in chunk order — the result does not depend on which workgroup
// that is (deterministic, no float atomics), and the separate reduce launch (a third of a decoder
// step's kernel time) is gone.  tickets: one counter per column slice, zero before the launch.
struct SkinnyFuse {
  const float *A2, *B2;
  int lda2, ldb2, K1;        // reduction indices [0, K1) come from (A, B), the rest from (A2, B2)
  unsigned *tickets;
  int heavy;                 // 1: full agent-scope fences around the ticket (debugging)
  SkinnyEpilogue ep;         // what the workgroup that holds the finished tile does with it
  float *C2;                 // columns >= split (a multiple of 32) go to C2[m * ldc2 + n - split] with beta2
  int ldc2, split;
  float beta2;
};

template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_fused_kernel(GemmArgs a, SkinnyFuse f) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * 32;
  const int KC = a.ksplit;
  int k0 = blockIdx.y * KC;
  const float *Ap = a.A, *Bp = a.B;
  int lda = a.lda, ldb = a.ldb;
  if (k0 >= f.K1) { k0 -= f.K1; Ap = f.A2; Bp = f.B2; lda = f.lda2; ldb = f.ldb2; }
  f32x16 acc[MT];
  skinny_tile<MT>(Ap, lda, Bp, ldb, k0, KC, a.M, n0, sm, acc);
  float *red = sm;
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((w * MT + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  const int ns = a.nsplit;
  __shared__ float zt[64 * 32];          // the finished tile [row][column of the slice] (cell epilogues)
  const int epi = f.ep.kind;
  for (int e = tid; e < MT * 16 * 64; e += 256) {
    const float s = red[e] + red[MT * 1024 + e] + red[2 * MT * 1024 + e] + red[3 * MT * 1024 + e];
    const int l = e & 63, r = (e >> 6) & 15, t = e >> 10;
    const int m = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = n0 + (l & 31);
    if (m >= a.M) continue;
    if (ns == 1) {
      float v = a.alpha * s + (a.bias ? a.bias[n] : 0.f);
      float *c = n0 >= f.split ? f.C2 + (size_t)m * f.ldc2 + (n - f.split) : a.C + (size_t)m * a.ldc + n;
      const float beta = n0 >= f.split ? f.beta2 : a.beta;
      if (beta != 0.f) v += beta * *c;
      if (epi == 0) *c = v;
      else          zt[m * 32 + (l & 31)] = v;
    } else {   // sc1 (write-through) store: performed at the device coherence point, whatever XCD reads it
      __hip_atomic_store(a.partial + ((size_t)blockIdx.y * a.M + m) * a.N + n, s, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (ns > 1) {
    // Hand-off recipe of MI355X_MICROARCH.md ("sc1 stores AND sc1 loads"): the partial tiles are written
    // through and read back with L1/L2-bypassing loads, the ticket is an agent-scope atomic, and the stores
    // have been acknowledged (vmcnt 0) before the ticket is taken — no assumption about which XCD the
    // partners of a column slice run on.  NABU_SKINNY_HEAVY=1 adds full agent-scope release/acquire fences
    // (L2 write-back + invalidate), for debugging; it doubles the kernel's duration.
    if (f.heavy) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    else         __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(f.tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last_flag = (old == (unsigned)(ns - 1));
    }
    __syncthreads();
    if (!last_flag) return;
    if (f.heavy) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    for (int e = tid; e < a.M * 32; e += 256) {
      const int m = e >> 5, n = n0 + (e & 31);
      float s = 0.f;
      for (int z0 = 0; z0 < ns; z0 += 8) {   // chunk order: independent of the arrival order
        float pv[8];                         // all loads of a batch in flight before the first add
#pragma unroll
        for (int j = 0; j < 8; ++j)
          pv[j] = (z0 + j < ns) ? __hip_atomic_load(a.partial + ((size_t)(z0 + j) * a.M + m) * a.N + n, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT)
                                : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += pv[j];
      }
      float v = a.alpha * s + (a.bias ? a.bias[n] : 0.f);
      float *c = n0 >= f.split ? f.C2 + (size_t)m * f.ldc2 + (n - f.split) : a.C + (size_t)m * a.ldc + n;
      const float beta = n0 >= f.split ? f.beta2 : a.beta;
      if (beta != 0.f) v += beta * *c;
      if (epi == 0) *c = v;
      else          zt[e] = v;
    }
    if (tid == 0) __hip_atomic_store(f.tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for reuse
  }
  if (epi == 0) return;
  __syncthreads();
  const SkinnyEpilogue &ep = f.ep;
  const int U = ep.U;
  if (epi == 1) {
    // tile columns 4j+g: unit u = n0/4 + j, gates i,j,f,o (TF LSTMCell, forget bias 1)
    for (int e = tid; e < a.M * 8; e += 256) {
      const int b = e >> 3, u = (n0 >> 2) + (e & 7);
      const size_t idx = (size_t)b * U + u, zo = (size_t)b * 4 * U + u;
      if (ep.step >= ep.seq_len[b]) {   // finished row: dynamic_decode(impute_finished) freezes the state
        ep.c_new[idx] = ep.c_prev[idx];
        ep.h_new[idx] = ep.h_prev[idx];
        ep.acts[zo] = ep.acts[zo + U] = ep.acts[zo + 2 * U] = ep.acts[zo + 3 * U] = 0.f;
        continue;
      }
      const float *zr = zt + b * 32 + 4 * (e & 7);
      float zi = zr[0] + ep.bias[u], zj = zr[1] + ep.bias[U + u];
      float zf = zr[2] + ep.bias[2 * U + u], zq = zr[3] + ep.bias[3 * U + u];
      if (ep.emb) {   // one-hot input times kernel == one row of the (unpermuted) kernel
        const float *em = ep.emb + (size_t)ep.ids[b] * 4 * U + u;
        zi += em[0]; zj += em[U]; zf += em[2 * U]; zq += em[3 * U];
      }
      const float i = sigmoidf_(zi), g = tanhf_(zj), fg = sigmoidf_(zf + 1.0f), o = sigmoidf_(zq);
      const float c = ep.c_prev[idx] * fg + i * g;
      ep.acts[zo] = i; ep.acts[zo + U] = g; ep.acts[zo + 2 * U] = fg; ep.acts[zo + 3 * U] = o;
      ep.c_new[idx] = c;
      ep.h_new[idx] = tanhf_(c) * o;
    }
  } else {
    // tile = dh of units n0 .. n0+31 (the query projection's gradient added to the direct one)
    for (int e = tid; e < a.M * 32; e += 256) {
      const int b = e >> 5, u = n0 + (e & 31);
      const size_t idx = (size_t)b * U + u, zo = (size_t)b * 4 * U + u;
      if (ep.step >= ep.seq_len[b]) {
        ep.dz[zo] = ep.dz[zo + U] = ep.dz[zo + 2 * U] = ep.dz[zo + 3 * U] = 0.f;
        ep.dc_out[idx] = ep.dc_in[idx];
        continue;
      }
      const float i = ep.acts[zo], g = ep.acts[zo + U], fg = ep.acts[zo + 2 * U], o = ep.acts[zo + 3 * U];
      const float tc = tanhf_(ep.c_new[idx]);
      const float dht = zt[e] + (ep.dh2 ? ep.dh2[(size_t)b * ep.ld_dh2 + u] : 0.f);
      const float dct = ep.dc_in[idx] + dht * o * (1.f - tc * tc);
      ep.dz[zo] = dct * g * i * (1.f - i);
      ep.dz[zo + U] = dct * i * (1.f - g * g);
      ep.dz[zo + 2 * U] = dct * ep.c_prev[idx] * fg * (1.f - fg);
      ep.dz[zo + 3 * U] = dht * tc * o * (1.f - o);
      ep.dc_out[idx] = dct * fg;
    }
  }
}

// C[M,N] = [A | A2]·[B ; B2] (+ bias, beta*C); M <= 64, N % 32 == 0, K1 and K2 multiples of the
// chunk (K2 may be 0).  partial: nchunks*M*N floats; tickets: N/32 zeroed counters (left zero).
int gemm_skinny_fused(int M, int N, int K1, const float *A, int lda, const float *B, int ldb, int K2, const float *A2,
                      int lda2, const float *B2, int ldb2, float beta, float *C, int ldc, const float *bias,
                      float *partial, unsigned *tickets, hipStream_t s, const SkinnyEpilogue *ep,
                      const SkinnySplit *split) {
  int kc = 0;
  for (int c = 256; c >= 64; c >>= 1)
    if (K1 % c == 0 && K2 % c == 0) { kc = c; break; }
  if (M > 64 || N % 32 != 0 || kc == 0 || K1 <= 0 || lda % 4 || (K2 && lda2 % 4))
    return fail(NABU_EUNSUP, "skinny product: unsupported shape M=%d N=%d K=%d+%d", M, N, K1, K2);
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.partial = partial;
  a.M = M; a.N = N; a.K = K1 + K2; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = 1.f; a.beta = beta; a.kseg = 0; a.a_seg = a.b_seg = 0;
  a.ksplit = kc; a.nsplit = (K1 + K2) / kc; a.vecA = a.vecB = 1; a.swz = 0;
  a.nbatch = 1; a.a_bs = a.b_bs = a.c_bs = 0;
  static int heavy_env = -1;
  if (heavy_env < 0) { const char *e = getenv("NABU_SKINNY_HEAVY"); heavy_env = e ? atoi(e) : 0; }
  SkinnyFuse f = {A2, B2, lda2, ldb2, K1, tickets, heavy_env ? 1 : 0, {}, nullptr, 0, N, 0.f};
  if (ep) f.ep = *ep; else f.ep.kind = 0;
  if (split) {
    if (split->split % 32 || split->split <= 0 || split->split >= N || (ep && ep->kind))
      return fail(NABU_EUNSUP, "skinny product: bad output split %d of %d columns", split->split, N);
    f.C2 = split->C2; f.ldc2 = split->ldc2; f.split = split->split; f.beta2 = split->beta2;
  }
  const int MT = M > 32 ? 2 : 1;
  const size_t xt = (size_t)kc * 32 * MT * sizeof(float), red = (size_t)4 * MT * 1024 * sizeof(float);
  const size_t lds = xt > red ? xt : red;
  dim3 grid(N / 32, a.nsplit);
  if (MT == 1) {
    hipLaunchKernelGGL(gemm_skinny_fused_kernel<1>, grid, dim3(256), lds, s, a, f);
  } else {
    // per call (cheap): the attribute is per device and this path is entered from several host threads
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_fused_kernel<2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(gemm_skinny_fused_kernel<2>, grid, dim3(256), lds, s, a, f);
  }
  NABU_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// The decoder chain's products of a sub-batch of <= 16 utterances, C[M <= 16, N] = A[M, K] . W^T with W the weight
// in its ORIGINAL row-major form [N][K] (dq . Wq^T, dz . [Kx | Kh]^T).  gemm_skinny_fused_kernel spends its 13 us on
// memory round trips one behind the other — operand tiles, write-through partial tiles, the stores' acknowledgement,
// the ticket, the partials again, then the epilogue's own operands — not on the 2 x 16 x 2048 x 1536 flops.  Here a
// workgroup owns 16 output columns over the WHOLE reduction (8 waves x K/8), nothing is handed between workgroups,
// and every load of the launch — both operands and what the epilogue reads — is issued before the first wait:
//   * the weights are re-blocked once per backward pass (rows16_swizzle) into the B-operand order of
//     v_mfma_f32_16x16x4_f32: float4 block ((column tile, k group of 16), lane (kq, fl)) = W[16 tile + fl][16 kg + 4 kq ..+3]
//     — a lane's 16-byte load is its B operand of four instructions, a wave's load one contiguous KiB;
//   * A's 16-byte load of lane (row fl, kq) = A[fl][16 kg + 4 kq ..+3] is the A operand of the same four instructions
//     (the k index inside a group is permuted identically on both sides);
//   * the eight waves' accumulators meet in LDS (fixed order), thread (row, column) runs the epilogue.
struct Rows16Args {
  const float *A, *Wsw;
  float *C, *C2;
  int M, N, K, lda, ldc, ldc2, split;
  float beta, beta2;
  const float *A2;          // reduction indices >= K1 (a multiple of 16) come from A2[m][k - K1] (the cell's [context | h])
  int lda2, K1;
  SkinnyEpilogue ep;
};
typedef float r16f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void rows16_swizzle_kernel(int N, int K, const float *__restrict__ W, int ldw,
                                                           float *__restrict__ out) {
  const size_t total = (size_t)N * K / 4;
  const int KG = K / 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), kg = (int)((i >> 6) % KG), nt = (int)((i >> 6) / KG);
    const int fl = lane & 15, kq = lane >> 4;
    reinterpret_cast<r16f4 *>(out)[i] = *reinterpret_cast<const r16f4 *>(W + (size_t)(16 * nt + fl) * ldw + 16 * kg + 4 * kq);
  }
}
// the same blocks from a weight stored [K][N] (forward products h . W): out block ((nt, kg), lane (kq, fl))[r] =
// W[16 kg + 4 kq + r][col(16 nt + fl)]; gate_units > 0: col(n) = (n % 4) * gate_units + n / 4 — output column 4 u + g
// is gate g of unit u, so that a 16-column tile holds the four gates of four units (the cell epilogue)
__global__ __launch_bounds__(256) void rows16_swizzle_kn_kernel(int N, int K, const float *__restrict__ W, int ldw,
                                                              float *__restrict__ out, int gate_units) {
  const size_t total = (size_t)N * K / 4;
  const int KG = K / 16;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int lane = (int)(i & 63), kg = (int)((i >> 6) % KG), nt = (int)((i >> 6) / KG);
    const int fl = lane & 15, kq = lane >> 4, n = 16 * nt + fl;
    const int col = gate_units > 0 ? (n & 3) * gate_units + (n >> 2) : n;
    const float *src = W + (size_t)(16 * kg + 4 * kq) * ldw + col;
    const r16f4 v = {src[0], src[ldw], src[2 * (size_t)ldw], src[3 * (size_t)ldw]};
    reinterpret_cast<r16f4 *>(out)[i] = v;
  }
}
int rows16_swizzle_kn(int N, int K, const float *W, int ldw, float *out, int gate_units, hipStream_t s) {
  if (N % 16 || K % 16 || (gate_units > 0 && N != 4 * gate_units)) return fail(NABU_EUNSUP, "rows16_swizzle_kn: N=%d K=%d", N, K);
  size_t blocks = ((size_t)N * K / 4 + 255) / 256;
  hipLaunchKernelGGL(rows16_swizzle_kn_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, N, K, W, ldw, out,
                     gate_units);
  NABU_LAUNCH_CHECK();
  return 0;
}
int rows16_swizzle(int N, int K, const float *W, int ldw, float *out, hipStream_t s) {
  if (N % 16 || K % 16 || ldw % 4) return fail(NABU_EUNSUP, "rows16_swizzle: N=%d K=%d ld=%d", N, K, ldw);
  size_t blocks = ((size_t)N * K / 4 + 255) / 256;
  hipLaunchKernelGGL(rows16_swizzle_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, N, K, W, ldw, out);
  NABU_LAUNCH_CHECK();
  return 0;
}

// FWD: the forward cell's form (two A segments, epilogue kind 1) — a separate instantiation so that the backward products
// keep one base pointer with immediate offsets and none of the cell's code (they lost ~1 us per launch to it)
template <int KG, bool FWD>      // k groups of 16 per wave: K = 128 KG
__global__ __launch_bounds__(512) void rows16_kernel(Rows16Args a) {
  __shared__ float red[8][16][17];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fl = lane & 15, kq = lane >> 4;
  const int nt = blockIdx.x, n0 = 16 * nt;
  const int mb = 16 * blockIdx.y;                               // row tile (gridDim.y tiles of 16 rows: M <= 64)
  const int eml = tid >> 4, fml = tid >> 2;                     // tile-local rows of the epilogue threads
  const int em = min(mb + eml, a.M - 1), en = tid & 15;         // the epilogue's (row, column) of threads < 256
  const SkinnyEpilogue &ep = a.ep;
  const int U = ep.U;
  // ---- loads: operands first, then what the epilogue reads (clamped addresses, masked later: no branch around a load)
  r16f4 bv[KG], av[KG];
  {
    const r16f4 *Bp = reinterpret_cast<const r16f4 *>(a.Wsw) + ((size_t)nt * (a.K / 16) + (size_t)w * KG) * 64 + lane;
    const int row = min(mb + fl, a.M - 1);
    const float *Ap = a.A + (size_t)row * a.lda + 4 * kq, *Ap2 = a.A2 + (size_t)row * a.lda2 - a.K1 + 4 * kq;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
      const int k = 16 * (w * KG + g);
      bv[g] = Bp[(size_t)g * 64];
      av[g] = *reinterpret_cast<const r16f4 *>(((!FWD || k < a.K1) ? Ap : Ap2) + k);
    }
  }
  const bool second = n0 >= a.split;
  float *cdst = second ? a.C2 + (size_t)em * a.ldc2 + (n0 - a.split) + en : a.C + (size_t)em * a.ldc + n0 + en;
  const float beta = second ? a.beta2 : a.beta;
  float cold = 0.f, e_i = 0.f, e_g = 0.f, e_f = 0.f, e_o = 0.f, e_cn = 0.f, e_cp = 0.f, e_dc = 0.f, e_dh2 = 0.f;
  int e_len = 0;
  const size_t eidx = (size_t)em * U + n0 + en, ezo = (size_t)em * 4 * U + n0 + en;
  if (beta != 0.f) cold = *cdst;
  if (ep.kind == 2) {
    e_len = ep.seq_len[em];
    e_i = ep.acts[ezo]; e_g = ep.acts[ezo + U]; e_f = ep.acts[ezo + 2 * U]; e_o = ep.acts[ezo + 3 * U];
    e_cn = ep.c_new[eidx]; e_cp = ep.c_prev[eidx]; e_dc = ep.dc_in[eidx];
    if (ep.dh2) e_dh2 = ep.dh2[(size_t)em * ep.ld_dh2 + n0 + en];
  }
  // kind 1 (forward cell): thread (row m = tid / 4, unit n0 / 4 + tid % 4) of the first wave
  const int fm = min(mb + fml, a.M - 1), fu = (n0 >> 2) + (tid & 3);
  float f_b[4] = {0.f, 0.f, 0.f, 0.f}, f_e[4] = {0.f, 0.f, 0.f, 0.f}, f_cp = 0.f, f_hp = 0.f;
  int f_len = 0;
  if (FWD && ep.kind == 1 && tid < 64) {
    f_len = ep.seq_len[fm];
    f_cp = ep.c_prev[(size_t)fm * U + fu];
    f_hp = ep.h_prev[(size_t)fm * U + fu];
#pragma unroll
    for (int g = 0; g < 4; ++g) f_b[g] = ep.bias[g * U + fu];
    if (ep.emb) {   // one-hot input times kernel == one row of the (unpermuted) kernel
      const float *em_row = ep.emb + (size_t)ep.ids[fm] * 4 * U + fu;
#pragma unroll
      for (int g = 0; g < 4; ++g) f_e[g] = em_row[g * U];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  r16f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < KG; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][r], bv[g][r], acc, 0, 0, 0);
  // accumulator register c = row 4 kq + c, lane fl = column
#pragma unroll
  for (int c = 0; c < 4; ++c) red[w][4 * kq + c][fl] = acc[c];
  __syncthreads();
  if (FWD && ep.kind == 1) {
    // the finished tile's columns 4 j + g: gates i, j, f, o of unit n0 / 4 + j (TF LSTMCell, forget bias 1)
    __shared__ float zt[16][17];
    if (tid < 256) {
      float zs = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) zs += red[i][tid >> 4][en];
      zt[tid >> 4][en] = zs;
    }
    __syncthreads();
    if (tid >= 64 || mb + fml >= a.M) return;
    const size_t idx = (size_t)fm * U + fu, zo = (size_t)fm * 4 * U + fu;
    float dsc = 1.f;          // output dropout of the cell (the mask of dropout_rows on this step's h)
    if (ep.ho_new && ep.keep > 0.f && ep.keep < 1.f) {
      const size_t e = (size_t)(ep.row0 + fm) * U + fu;
      const float4 sc = dropout_scale4(e >> 2, ep.keep, ep.seed, ep.seed_offset);
      const int j = (int)(e & 3);
      dsc = j == 0 ? sc.x : j == 1 ? sc.y : j == 2 ? sc.z : sc.w;
    }
    if (ep.step >= f_len) {   // finished row: dynamic_decode(impute_finished) freezes the state
      ep.c_new[idx] = f_cp;
      ep.h_new[idx] = f_hp;
      if (ep.ho_new) ep.ho_new[idx] = f_hp * dsc;
      ep.acts[zo] = ep.acts[zo + U] = ep.acts[zo + 2 * U] = ep.acts[zo + 3 * U] = 0.f;
      return;
    }
    const float *zr = &zt[fml][4 * (tid & 3)];
    const float zi = zr[0] + f_b[0] + f_e[0], zj = zr[1] + f_b[1] + f_e[1], zf = zr[2] + f_b[2] + f_e[2], zq = zr[3] + f_b[3] + f_e[3];
    const float i = sigmoidf_(zi), g = tanhf_(zj), fg = sigmoidf_(zf + 1.0f), o = sigmoidf_(zq);
    const float c = f_cp * fg + i * g;
    ep.acts[zo] = i; ep.acts[zo + U] = g; ep.acts[zo + 2 * U] = fg; ep.acts[zo + 3 * U] = o;
    ep.c_new[idx] = c;
    const float hn = tanhf_(c) * o;
    ep.h_new[idx] = hn;
    if (ep.ho_new) ep.ho_new[idx] = hn * dsc;
    return;
  }
  if (tid >= 256 || mb + eml >= a.M) return;
  float v = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) v += red[i][eml][en];
  if (beta != 0.f) v = fmaf(beta, cold, v);
  if (ep.kind == 0) { *cdst = v; return; }
  // kind 2: v = d h of unit n0 + en (the query projection's gradient added to the direct one): LSTM cell backward
  if (ep.step >= e_len) {
    ep.dz[ezo] = ep.dz[ezo + U] = ep.dz[ezo + 2 * U] = ep.dz[ezo + 3 * U] = 0.f;
    ep.dc_out[eidx] = e_dc;
    return;
  }
  if (ep.keep > 0.f && ep.keep < 1.f) {
    const size_t e = (size_t)(ep.row0 + em) * U + n0 + en;
    const float4 sc = dropout_scale4(e >> 2, ep.keep, ep.seed, ep.seed_offset);
    const int j = (int)(e & 3);
    v *= j == 0 ? sc.x : j == 1 ? sc.y : j == 2 ? sc.z : sc.w;
  }
  const float tc = tanhf_(e_cn);
  const float dht = v + e_dh2;
  const float dct = e_dc + dht * e_o * (1.f - tc * tc);
  ep.dz[ezo] = dct * e_g * e_i * (1.f - e_i);
  ep.dz[ezo + U] = dct * e_i * (1.f - e_g * e_g);
  ep.dz[ezo + 2 * U] = dct * e_cp * e_f * (1.f - e_f);
  ep.dz[ezo + 3 * U] = dht * tc * e_o * (1.f - e_o);
  ep.dc_out[eidx] = dct * e_f;
}

bool rows16_ok(int M, int N, int K, int lda) {
  const int kg = K / 128;
  return M >= 1 && M <= 64 && N % 16 == 0 && K % 128 == 0 && lda % 4 == 0 &&
         (kg == 1 || kg == 2 || kg == 3 || kg == 4 || kg == 6 || kg == 8 || kg == 12 || kg == 16);
}
// C[M, N] = A . W^T (+ beta C); ep: kind 0 (plain) or 2 (LSTM cell backward on the finished tile, as gemm_skinny_fused);
// split: columns >= split->split (a multiple of 16) go to split->C2 with split->beta2
int rows16(int M, int N, int K, const float *A, int lda, const float *Wsw, float beta, float *C, int ldc, hipStream_t s,
           const SkinnyEpilogue *ep, const SkinnySplit *split, int K1, const float *A2, int lda2) {
  if (!rows16_ok(M, N, K, lda) || (ep && ep->kind != 0 && ep->kind != 1 && ep->kind != 2) ||
      (split && (split->split % 16 || (ep && ep->kind))) || (A2 && (K1 % 16 || K1 <= 0 || K1 >= K || lda2 % 4)))
    return fail(NABU_EUNSUP, "rows16 product: unsupported shape M=%d N=%d K=%d", M, N, K);
  Rows16Args a = {};
  a.A = A; a.Wsw = Wsw; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldc = ldc; a.beta = beta;
  a.C2 = nullptr; a.ldc2 = 0; a.split = N; a.beta2 = 0.f;
  a.A2 = A2 ? A2 : A; a.lda2 = A2 ? lda2 : lda; a.K1 = A2 ? K1 : K;
  if (split) { a.C2 = split->C2; a.ldc2 = split->ldc2; a.split = split->split; a.beta2 = split->beta2; }
  if (ep) a.ep = *ep; else a.ep.kind = 0;
  if (a.ep.kind == 1) { a.C = nullptr; a.beta = 0.f; }      // the forward cell writes its state, not the product
  const dim3 grid(N / 16, (M + 15) / 16), block(512);
  const bool fwd = a.ep.kind == 1 || A2 != nullptr;
#define ROWS16_CASE(kg)                                                                    \
  case kg:                                                                                 \
    if (fwd) hipLaunchKernelGGL((rows16_kernel<kg, true>), grid, block, 0, s, a);          \
    else     hipLaunchKernelGGL((rows16_kernel<kg, false>), grid, block, 0, s, a);         \
    break;
  switch (K / 128) {
    ROWS16_CASE(1) ROWS16_CASE(2) ROWS16_CASE(3) ROWS16_CASE(4) ROWS16_CASE(6) ROWS16_CASE(8) ROWS16_CASE(12)
    default:
      if (fwd) hipLaunchKernelGGL((rows16_kernel<16, true>), grid, block, 0, s, a);
      else     hipLaunchKernelGGL((rows16_kernel<16, false>), grid, block, 0, s, a);
  }
#undef ROWS16_CASE
  NABU_LAUNCH_CHECK();
  return 0;
}

// k-chunk of the skinny kernel: the largest of 256/128/64 that divides K (0 = not eligible)
int gemm_skinny_chunk(int M, int N, int K) {
  if (M > 64 || N % 32 != 0 || K < 64) return 0;
  for (int kc = 256; kc >= 64; kc >>= 1)
    if (K % kc == 0) return kc;
  return 0;
}

int gemm_skinny_launch(const GemmArgs &a, hipStream_t s) {
  const int MT = a.M > 32 ? 2 : 1;
  const size_t xt = (size_t)a.ksplit * 32 * MT * sizeof(float), red = (size_t)4 * MT * 1024 * sizeof(float);
  const size_t lds = xt > red ? xt : red;
  dim3 grid(a.N / 32, a.nsplit);
  if (MT == 1) {
    hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, dim3(256), lds, s, a);
  } else {
    NABU_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_skinny_kernel<2>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, dim3(256), lds, s, a);
  }
  NABU_LAUNCH_CHECK();
  return 0;
}

}  // namespace nabu
