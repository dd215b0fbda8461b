/* nabu_hip.h — C ABI of libnabu_hip.so: the MI355X (gfx950) kernels behind the
 * Nabu training hot path.
 *
 * The reference (vrenkens/nabu) has no FFI: its numerical backend is the
 * TensorFlow-1.8 op library, called from Python.  Each entry point below names
 * the reference call site (path relative to the reference root, file:line) whose
 * TF ops it replaces.  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *    the caller owns all memory, the library never allocates;
 *  - tensors are contiguous, batch-major, float32; lengths/labels are int32;
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *  - return value: 0 = ok, < 0 = NABU_E* argument error, > 0 = hipError_t;
 *    nabu_last_error() returns a thread-local message;
 *  - no C++ exceptions cross the boundary;
 *  - state the library keeps between calls — all of it listed here: (process-wide, set
 *    before use, not synchronised) the default GEMM arithmetic (nabu_gemm_set_default_precision /
 *    NABU_GEMM_PRECISION) and the persistent kernels' wait bound (nabu_persist_set_timeout_us);
 *    (thread-local) the last error text, the profiling events of
 *    nabu_blstm_set_profile_events and the hook of nabu_blstm_set_phase_hook; plus a lazily built per-kernel attribute cache.  Everything
 *    else — parameters, activations, workspaces, status words — lives in caller-owned memory.
 */
#ifndef NABU_HIP_H
#define NABU_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NABU_ABI_VERSION 3

#define NABU_EINVAL   (-1)  /* bad argument (shape, null pointer, alignment) */
#define NABU_EUNSUP   (-2)  /* shape not supported by the requested kernel   */
#define NABU_EWS      (-3)  /* workspace too small                           */

typedef void *nabu_stream_t; /* hipStream_t */

int nabu_version(void);
const char *nabu_last_error(void);

/* ------------------------------------------------------------------------
 * Dense fp32 GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32):
 *   C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C + bias[N]
 * row-major; op(X) = X or X^T (trans* != 0 means the matrix is STORED
 * transposed, i.e. A is [K,M] / B is [N,K]).  bias may be NULL.
 * Segmented K (kseg > 0): the reduction index k runs over K = nseg*kseg rows
 * that are stored as nseg segments of kseg consecutive rows, segment s of A
 * starting a_seg_stride elements after segment s-1 (same for B); used for the
 * per-utterance shifted h_{t-1}^T·dz products.  Only valid with transA=1,
 * transB=0.  kseg = 0 means one contiguous segment.
 * Replaces: tf MatMul/BiasAdd inside LayerNormBasicLSTMCell._linear
 * (nabu/neuralnetworks/components/layer.py:35-47), tf.contrib.layers.linear
 * (models/ed_decoders/dnn_decoder.py:53-57), Dense layers of the attention
 * (components/attention.py:163-175) and their autodiff (trainers/trainer.py:556).
 * ws: split-K partial sums; query the size with nabu_gemm_ws_bytes. */
size_t nabu_gemm_ws_bytes(int M, int N, int K);
/* Arithmetic of the product (operands and result are fp32 in memory in every mode):
 *   NABU_GEMM_F32    exact fp32 on v_mfma_f32_32x32x2_f32 (the default);
 *   NABU_GEMM_BF16   operands rounded to bf16 inside the kernel, fp32 accumulation, on
 *                    v_mfma_f32_32x32x16_bf16 — the "bf16 MFMA input-to-hidden GEMMs" of
 *                    BASELINE.json configs[4];
 *   NABU_GEMM_BF16X3 / NABU_GEMM_BF16X6  each operand split into 2 / 3 bf16 pieces, 3 / 6 bf16
 *                    MFMA products: ~2^-16 / fp32-level (<= 2^-23) relative accuracy;
 *   NABU_GEMM_F16X3  fp32-level accuracy from three fp16 MFMA products of row-scaled two-plane operands
 *                    (nabu_pk_pack_f16 below): the BLSTM layer's time-batched products take it on packed
 *                    operands; a product on row-major operands (this function) computes bf16x6 instead.
 * NABU_GEMM_DEFAULT = the process default: NABU_GEMM_F32 unless changed by
 * nabu_gemm_set_default_precision() or the environment variable NABU_GEMM_PRECISION
 * (f32 | bf16 | bf16x3 | bf16x6 | f16x3).  Shapes the bf16 kernels do not take (K % 32, M % 4, N % 4,
 * unaligned operands) silently use the exact fp32 kernel — never a lower precision than asked. */
enum { NABU_GEMM_DEFAULT = 0, NABU_GEMM_F32 = 1, NABU_GEMM_BF16 = 2, NABU_GEMM_BF16X3 = 3, NABU_GEMM_BF16X6 = 4,
       NABU_GEMM_F16X3 = 5 };
int nabu_gemm_set_default_precision(int precision);
int nabu_gemm_get_default_precision(void);
int nabu_gemm_ex(int precision, int transA, int transB, int M, int N, int K, float alpha,
                 const float *A, int lda, const float *B, int ldb, float beta, float *C,
                 int ldc, const float *bias, int kseg, long long a_seg_stride,
                 long long b_seg_stride, void *ws, size_t ws_bytes, nabu_stream_t stream);
int nabu_gemm_f32(int transA, int transB, int M, int N, int K, float alpha,
                  const float *A, int lda, const float *B, int ldb, float beta,
                  float *C, int ldc, const float *bias, int kseg,
                  long long a_seg_stride, long long b_seg_stride, void *ws,
                  size_t ws_bytes, nabu_stream_t stream);

/* Decoder-step product: C[M,N] = [A | A2]·[B ; B2] + beta*C + bias for M <= 64 batch rows — the
 * reduction index runs over two operand pairs (the Speller cell's concat([context, h])·kernel,
 * nabu/neuralnetworks/models/ed_decoders/speller.py:33-45 -> tf LSTMCell's single matmul on the
 * concatenated input) without a concatenated copy, in ONE launch: workgroups own (column slice,
 * k-chunk) pairs, write their partial tiles through to memory, and the last arriver of a column
 * slice (a ticket per slice) sums the partials in chunk order — deterministic, no float atomics,
 * no separate reduce launch.  Requirements: N % 32 == 0, K1 and K2 multiples of 64 (K2 may be 0),
 * lda/lda2 % 4 == 0, 16-byte aligned operands; NABU_EUNSUP otherwise.  Exact fp32. */
size_t nabu_gemm2_ws_bytes(int M, int N, int K1, int K2);
int nabu_gemm2_f32(int M, int N, int K1, const float *A, int lda, const float *B, int ldb, int K2,
                   const float *A2, int lda2, const float *B2, int ldb2, float beta, float *C, int ldc,
                   const float *bias, void *ws, size_t ws_bytes, nabu_stream_t stream);

/* bf16-RESIDENT operands — the "bf16 MFMA input-to-hidden GEMMs" of BASELINE.json configs[4] with the
 * operands converted once instead of inside every product (each is used several times: the layer input
 * in the forward product and the weight gradient, dz in the input gradient and the weight gradient):
 *   nabu_cvt_bf16      dst (bf16, [R,ldd], or [C,ldd] when transpose != 0) = RNE(src fp32 [R,ld]);
 *                      plain copy: C % 8 == 0, ld % 4 == 0, ldd % 8 == 0; transposed: ld % 4, ldd % 2;
 *   nabu_gemm_bf16_nt  C[M,N] (fp32) = alpha * sum_k A[m,k]·B[n,k] + beta*C + bias[N], A [M,lda] and B [N,ldb]
 *                      bf16 with the reduction index contiguous; K % 64 == 0, lda % 8 == ldb % 8 == 0
 *                      (NABU_EUNSUP otherwise); fp32 accumulation (v_mfma_f32_32x32x16_bf16), deterministic
 *                      split-K through ws (nabu_gemm_bf16_nt_ws_bytes).
 * nabu_blstm_fwd/_bwd use them for the products X·Wx, dZ·Wx^T, X^T·dZ when desc.gemm_precision is
 * NABU_GEMM_BF16 and the shapes allow; same rounding as NABU_GEMM_BF16 of nabu_gemm_ex.
 * Replaces: the same tf MatMul ops as nabu_gemm_ex (components/layer.py:35-47 and their autodiff). */
int nabu_cvt_bf16(size_t R, int C, const float *src, int ld, void *dst_bf16, int ldd, int transpose,
                  nabu_stream_t stream);
size_t nabu_gemm_bf16_nt_ws_bytes(int M, int N, int K);
int nabu_gemm_bf16_nt(int M, int N, int K, float alpha, const void *A_bf16, int lda, const void *B_bf16, int ldb,
                      float beta, float *C, int ldc, const float *bias, void *ws, size_t ws_bytes,
                      nabu_stream_t stream);

/* PACKED bf16-plane operands (gemm_pk.hip) — the time-batched input-to-hidden products and their
 * gradients on the bf16 matrix pipe at fp32-equivalent accuracy ("bf16x6", planes = 3) or in plain bf16
 * (planes = 1, BASELINE.json configs[4]).  An fp32 operand is converted ONCE per use-site into
 *     packed[kb][plane][row][16 k]   bf16; row = the operand's M (A) or N (B) index, padded to a multiple
 *                                    of 256 (nabu_pk_rows_pad); kb = k / 16, padded with zero blocks to
 *                                    nabu_pk_kblocks(K, planes); within a row's 32 bytes the two 16-byte
 *                                    halves are swapped when bit 3 of the row index is set;
 *     planes = 3: x = h + m + l with h = rne_bf16(x), m = rne_bf16(x - h), l = x - h - m (exact: the three
 *                 8-bit significands hold the 24 bits of x);  planes = 1: h only.
 * nabu_pk_pack writes one source matrix into a packed operand: rows [row_off, row_off + fill_rows) and
 * k-blocks [kb_off, kb_off + fill_kb), zeros where the source (R x C valid elements, row stride ld) ends.
 *   transposed = 0: packed row = source row, k = source column (fill_rows over R, fill_kb over ceil(C/16));
 *   transposed = 1: packed row = source column, k = source row (fill_rows over C, fill_kb over ceil(R/16));
 *                   with period > 0 the value at reduction index r is source row r + shift when
 *                   0 <= r % period + shift < period and 0 otherwise (the h_{t-1}^T·dz_t pairs of the
 *                   recurrent weight gradient: period = T, shift = -1 / +1 for the forward / backward cell).
 * nabu_gemm_pk: C_b[M,N] = alpha * sum_k A_b[m,k]·B_b[n,k] + beta*C_b + bias[n] for b < nbatch (<= 2), fp32
 * result; planes = 3 adds the six plane products h·h, h·m, m·h, m·m, h·l, l·h (v_mfma_f32_32x32x16_bf16, fp32
 * accumulation).  Columns n >= n_split (a multiple of 256, 0 = off) are written to C2_b[m, n - n_split] with
 * bias2 — one product fills the gate buffers of both directions.  Deterministic split-K through ws.
 * An operand packed with 3 planes can be used by a planes = 1 product (a_planes / b_planes = planes stored).
 * Replaces: the tf MatMul of LayerNormBasicLSTMCell._linear over all frames and its autodiff
 * (nabu/neuralnetworks/components/layer.py:35-47, nabu/neuralnetworks/trainers/trainer.py:556-558). */
typedef struct nabu_pk_gemm_desc {
  uint32_t size;          /* = sizeof(nabu_pk_gemm_desc) */
  int32_t planes;         /* 3 = bf16x6 (fp32-equivalent), 2 = f16x3 (fp32-equivalent, scaled fp16 planes), 1 = bf16 */
  int32_t M, N, nkb;      /* output size per batch entry; k-blocks of 16 to reduce over */
  int32_t nbatch;         /* 1 or 2 independent products in one launch */
  const void *A[2], *B[2];/* packed operands (first k-block of the reduction) */
  int32_t a_rows_pad, b_rows_pad, a_planes, b_planes;
  float *C[2], *C2[2];
  int32_t ldc, n_split;
  const float *bias, *bias2;
  float alpha, beta;
  const uint32_t *a_amax[2], *b_amax[2];  /* planes = 2: the row maxima the operands were packed with */
  int32_t direct;         /* planes = 2: 1 = chain the three plane products into the accumulators directly (three
                           * roundings of the matrix pipe per 16 k where the exact-fp32 instruction has eight; 10 %
                           * faster) instead of promoting 16-k partial sums (0, the default: one rounding per 16 k);
                           * 2 = direct when a workgroup's reduction is at most twice as long as the exact-fp32
                           * kernel's would be for the same product (fewer roundings than that kernel), else promoted */
} nabu_pk_gemm_desc;
int nabu_pk_rows_pad(int rows);
int nabu_pk_kblocks(int K, int planes);
size_t nabu_pk_bytes(int rows, int K, int planes);
int nabu_pk_pack(int planes, int transposed, const float *src, long long ld, int R, int C, void *dst,
                 int dst_rows_pad, int row_off, int kb_off, int fill_rows, int fill_kb, int period,
                 int shift, nabu_stream_t stream);
/* planes = 2, "f16x3": two scaled fp16 planes per operand and three plane products (l·h, h·l, h·h on
 * v_mfma_f32_32x32x16_f16) — half the matrix instructions of bf16x6 at the same error level.  Every packed ROW
 * (an M resp. N index, all its k) has a power-of-two scale 2^(14 - floor(log2 amax)) derived from `amax`, the
 * bit pattern of the row's largest magnitude: x·scale = h + l with h = rne_f16(x·scale), l = rne_f16(x·scale - h);
 * the product's epilogue divides the scales out.  amax[dst_rows_pad] (uint32, device) is indexed by packed row;
 * it comes from nabu_pk_amax (atomic maxima of the rows and / or columns of a source matrix into ZEROED arrays:
 * `rows[r]` for transposed = 0 packs, `cols[c]` for transposed = 1 packs) or from nabu_pk_amax_fill when a
 * bound is known a priori (LSTM outputs: 1).  Any upper bound of the row's magnitudes is valid; a bound 2^j too
 * large costs j bits of the l plane.  The same array goes into nabu_pk_gemm_desc.a_amax / b_amax. */
int nabu_pk_amax(const float *src, long long ld, int R, int C, uint32_t *rows, uint32_t *cols, nabu_stream_t stream);
int nabu_pk_amax_fill(uint32_t *dst, int n, float value, nabu_stream_t stream);
int nabu_pk_pack_f16(int transposed, const float *src, long long ld, int R, int C, void *dst, int dst_rows_pad,
                     int row_off, int kb_off, int fill_rows, int fill_kb, int period, int shift,
                     const uint32_t *amax, nabu_stream_t stream);
size_t nabu_gemm_pk_ws_bytes(const nabu_pk_gemm_desc *d);
int nabu_gemm_pk(const nabu_pk_gemm_desc *d, void *ws, size_t ws_bytes, nabu_stream_t stream);

/* out[n] = beta*out[n] + sum_m A[m*lda + n]  (bias gradients; deterministic
 * two-stage tree).  ws >= nabu_colsum_ws_bytes(M,N). */
size_t nabu_colsum_ws_bytes(int M, int N);
int nabu_colsum_f32(int M, int N, const float *A, int lda, float beta, float *out,
                    void *ws, size_t ws_bytes, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * One bidirectional LSTM layer — layer.blstm
 * (nabu/neuralnetworks/components/layer.py:8-51): two
 * tf.contrib.rnn.LayerNormBasicLSTMCell(layer_norm=False) driven by
 * bidirectional_dynamic_rnn(sequence_length=len) and concatenated on the
 * feature axis.  Gate order i,j,f,o; forget bias +1 at run time; rows t >= len
 * produce 0 and freeze (c,h); the backward direction starts at each sequence's
 * own last frame.
 *   x [B,T,D], len [B], kernel_* [(D+H),4H], bias_* [4H], out [B,T,2H].
 * reserve (nabu_blstm_reserve_bytes) keeps the gate activations and cell states
 * for the backward pass; ws (nabu_blstm_ws_bytes) is scratch.
 * mode: NABU_LSTM_AUTO picks the persistent whole-sequence kernel when the
 * shape is supported, else one launch per timestep.
 * Persistent kernels, by shape (nabu_amd/csrc/), three families: H in {128, 256, 512} on a whole MI355X (256 CUs): the
 * recurrent product on the 16-bit matrix pipe as three fp16 plane products of row-scaled operands — lstm_persist_mxh.hip:
 * launches of up to 32 batch rows, 8 per unit, 16 hidden units per workgroup (a narrow first-layer input, D <= 64, is
 * projected inside the forward kernel); lstm_persist_mxf.hip: 33..64 rows at H = 512 as sixteen units of 8 rows, 32 hidden
 * units per workgroup; fp32-equivalent: outputs and gradients agree with the exact-fp32 kernels to rounding and are as
 * close to a float64 layer (tests/test_hip_fullsize.py, tests/test_hip_real_operands.py); larger batches as consecutive
 * launches; otherwise (H = 64, fewer CUs, NABU_PERSIST_MX=0, recurrent_precision = NABU_REC_F32) the exact-fp32
 * v_mfma_f32_4x4x1 kernels of lstm_persist.hip.  All need their whole grid co-resident: every launch is validated against
 * the occupancy query first (all chunks of a call before the first is enqueued) and NABU_LSTM_AUTO steps instead where it
 * does not fit. */
#define NABU_LSTM_AUTO       0
#define NABU_LSTM_STEPWISE   1
#define NABU_LSTM_PERSISTENT 2

/* nabu_blstm_desc.flags */
#define NABU_BLSTM_FWD_ONLY   1   /* no backward pass will follow (validation, decoding): the reserve holds the
                                     activations only — no room for the packed dZ^T / row maxima of the weight-gradient
                                     products (0.2-0.8 GB per cfg2 layer); nabu_blstm_bwd* reject such a descriptor */
/* nabu_blstm_desc.recurrent_precision */
#define NABU_REC_DEFAULT      0   /* the recurrent product h.W_h on the 16-bit matrix pipe: three fp16 plane products of
                                     row-scaled operands, fp32-equivalent (lstm_persist_mxh.hip / _mxf.hip) */
#define NABU_REC_F32          1   /* exact fp32 (v_mfma_f32_4x4x1, lstm_persist.hip): with gemm_precision = NABU_GEMM_F32
                                     the whole layer is float32 end to end, like layer.py:35-47 on TF's fp32 MatMul */

typedef struct {
  uint32_t size;      /* sizeof(nabu_blstm_desc), ABI versioning (the 32-byte layout of ABI version 1 — everything
                         up to gemm_precision — is still accepted: the fields behind it read as 0) */
  int32_t B, T, D, H;
  int32_t max_len;    /* max(len) if known on the host, else T */
  int32_t mode;       /* NABU_LSTM_* */
  int32_t gemm_precision; /* NABU_GEMM_* of the input-to-hidden products X·Wx, dZ·Wx^T, X^T·dZ
                             (BASELINE.json configs[4]: "bf16 MFMA input-to-hidden GEMMs");
                             the recurrent weight gradient H^T·dZ follows the process default */
  float x_bound;      /* > 0: the caller guarantees |x| <= x_bound for every element of the layer input — the previous
                         layer's LSTM outputs (|o tanh c| <= 1), after dropout 1 / keep_prob: the f16x3 packs of x take
                         their row scales from the bound and skip the measuring pass over x.  0: nothing is known, x is
                         measured (the first layer's features) */
  int32_t flags;      /* NABU_BLSTM_* */
  int32_t recurrent_precision;   /* NABU_REC_* */
  /* ---- ABI version 3 (the 44-byte layout of version 2 — everything up to recurrent_precision — is still accepted).
   * PACKED COMPANIONS of the layer's output and input.  With gemm_precision = NABU_GEMM_F16X3 every dense product of a
   * layer reads its operands as two fp16 planes (nabu_pk_pack_f16's layout); up to ABI version 2 each call converted
   * x, x^T and h^T itself, re-reading tensors the previous recurrent kernel had written microseconds earlier (0.44 ms of
   * a 12 ms cfg2 step).  |h| = |o tanh c| <= 1 is known a priori, so the forward recurrent kernel can write the planes
   * itself, next to `out`, at the scale 2^14 it already uses for its own exchange (row maxima = the bit pattern of
   * 1.0f for every packed row):
   *   out_pk_rows  `out` as the NEXT layer's input operand: rows = frames after stacking `out_stack` consecutive
   *                frames (ops.pyramid_stack, components/ops.py:6-60, as a view: T %% out_stack == 0), k = out_stack 2H;
   *                nabu_pk_bytes(B T / out_stack, out_stack 2H, 2) bytes
   *   out_pk_cols  the transposed operand of the same matrix (the next layer's x^T . dZ): nabu_pk_bytes(out_stack 2H,
   *                B T / out_stack, 2)
   *   hT_pk        this layer's own h_(t-1)^T operands of dWh, forward cell then backward cell, each
   *                nabu_pk_bytes(Mw, B T, 2), Mw = H (D + H where the first layer's x^T shares the operand: D < 256);
   *                written by nabu_blstm_fwd, read by nabu_blstm_bwd / _bwd_weights of the SAME descriptor
   *   x_pk_rows / x_pk_cols   this layer's input as the producer layer wrote it (its out_pk_rows / out_pk_cols): the
   *                call reads no fp32 x for its products
   * All optional (NULL = the call packs for itself, as before).  Buffers are caller-owned and must be ZERO-FILLED ONCE
   * before their first use (the kernels write only positions that exist: k-blocks beyond the reduction length and rows
   * beyond the operand stay zero) and may then be reused call after call.  nabu_blstm_fwd GUARANTEES the buffers it is
   * given are complete on return, whatever path it takes: written by the recurrent kernel where that is possible
   * (fp16-plane kernels, launches of <= 32 rows, max_len == T: nabu_blstm_emits_packed), by the pack kernels otherwise.
   * nabu_blstm_pk_bytes reports the sizes and whether the layer described would use the companions at all. */
  int32_t out_stack;            /* frames of `out` per packed row of out_pk_rows / out_pk_cols: 1 or 2 (0 reads as 1) */
  void *out_pk_rows;
  void *out_pk_cols;
  void *hT_pk;
  const void *x_pk_rows;
  const void *x_pk_cols;
} nabu_blstm_desc;

/* sizes of the packed companions of d (bytes; 0 where the layer described does not use that companion: another
 * gemm_precision, fewer than 2048 frames, an input of fewer than 256 features ...):
 *   bytes[0], bytes[1]  x_pk_rows, x_pk_cols as THIS layer would read them (its input [B, T, D])
 *   bytes[2]            hT_pk of this layer
 *   bytes[3], bytes[4]  out_pk_rows, out_pk_cols for d->out_stack (0 when T is not a multiple of it)
 * returns 0, or NABU_EINVAL for a bad descriptor */
int nabu_blstm_pk_bytes(const nabu_blstm_desc *d, size_t bytes[5]);
/* which of the companions d names nabu_blstm_fwd's recurrent kernel writes ITSELF: bit 0 out_pk_rows, bit 1 out_pk_cols,
 * bit 2 hT_pk (the others are made by pack kernels behind the recurrence — a caller that can pack lazily may prefer to
 * leave those out of the descriptor) */
int nabu_blstm_emits_packed(const nabu_blstm_desc *d);

/* RESERVE CONTRACT (since ABI version 2): nabu_blstm_fwd records, in host memory of THIS process, a fingerprint of the
 * layout it wrote (shape, planes, flags, recurrent_precision, sizes), keyed by the reserve's address; nabu_blstm_bwd /
 * _bwd_data / _bwd_weights compare it with the layout they derive and return NABU_EINVAL for a reserve that no forward
 * call of this process wrote, that was written under another layout (descriptor or process default precision changed
 * between the passes), or whose fingerprint has been displaced (the table keeps the 4096 most recently written
 * reserves).  A backward call therefore needs the forward call of the SAME process on the SAME address first; a
 * reserve copied elsewhere or produced by another process is rejected. */
size_t nabu_blstm_reserve_bytes(const nabu_blstm_desc *d);
size_t nabu_blstm_ws_bytes(const nabu_blstm_desc *d);
int nabu_blstm_fwd(const nabu_blstm_desc *d, const float *x, const int32_t *len,
                   const float *kernel_fw, const float *bias_fw,
                   const float *kernel_bw, const float *bias_bw, float *out,
                   void *reserve, void *ws, size_t ws_bytes, nabu_stream_t stream);
/* Gradient of nabu_blstm_fwd (autodiff of layer.blstm, trainers/trainer.py:556-558).
 * d_out [B,T,2H]; d_x [B,T,D] is overwritten (may be NULL for the first layer);
 * dkernel_*, dbias_* are overwritten.  reserve is consumed (overwritten). */
int nabu_blstm_bwd(const nabu_blstm_desc *d, const float *x, const int32_t *len,
                   const float *kernel_fw, const float *kernel_bw, const float *out,
                   const float *d_out, void *reserve, float *d_x, float *dkernel_fw,
                   float *dbias_fw, float *dkernel_bw, float *dbias_bw, void *ws,
                   size_t ws_bytes, nabu_stream_t stream);
/* The same gradient in two calls: nabu_blstm_bwd_data runs the recurrence backwards and produces what the layer
 * BELOW waits for (d_x) plus the bias gradients, leaving dz in `reserve`; nabu_blstm_bwd_weights turns that dz into
 * dkernel_fw / dkernel_bw ([(D+H),4H], overwritten) and may run any time later on the same stream with the same
 * x / out / reserve (a different ws is fine).  data + weights == nabu_blstm_bwd, bit for bit.  Why: nothing waits
 * for the weight gradients, and a long burst of bf16 matrix work in front of a latency-bound persistent recurrent
 * kernel slows that kernel down (the chip's clock recovers slowly: +0.5 ms on the 1000-frame layer of cfg2), so a
 * trainer runs the weight-gradient products of all layers after the last recurrence (TF's scheduler was free to
 * do the same with the MatMul gradients of trainers/trainer.py:556-558). */
int nabu_blstm_bwd_data(const nabu_blstm_desc *d, const float *x, const int32_t *len,
                        const float *kernel_fw, const float *kernel_bw, const float *out,
                        const float *d_out, void *reserve, float *d_x, float *dbias_fw, float *dbias_bw,
                        void *ws, size_t ws_bytes, nabu_stream_t stream);
int nabu_blstm_bwd_weights(const nabu_blstm_desc *d, const float *x, const int32_t *len, const float *out,
                           void *reserve, float *dkernel_fw, float *dkernel_bw, void *ws, size_t ws_bytes,
                           nabu_stream_t stream);

/* 1 if nabu_blstm_fwd/bwd will run the persistent whole-sequence kernel for d. */
int nabu_blstm_uses_persistent(const nabu_blstm_desc *d);
/* Profiling hook (thread-local): when non-NULL, nabu_blstm_fwd/bwd record the
 * caller-owned hipEvent_t ev_begin right before and ev_end right after the
 * recurrent kernel(s) on the call's stream, so that bench.py can time the
 * dominant kernel live inside the timed region.  Pass NULLs to switch it off. */
int nabu_blstm_set_profile_events(void *ev_begin, void *ev_end);
/* Hook (thread-local) that nabu_blstm_bwd calls on the host right after it has enqueued the
 * recurrent kernel(s) and before it enqueues the dense products dWx, dWh, dx.  The persistent
 * recurrent kernels need every one of their workgroups resident at the same time, so nothing else
 * may run beside them; the products have no such constraint.  A data-parallel caller uses the hook
 * to start the all-reduce of an already finished gradient bucket on its communication stream at
 * exactly this point (ordered after the recurrence by an event) and joins that stream again before
 * the next recurrent launch — the exchange then overlaps the products only.  NULL switches it off.
 * Replaces: the asynchronous parameter-server pushes of the reference (trainers/trainer.py:479-510). */
typedef void (*nabu_phase_hook_t)(void *user);
int nabu_blstm_set_phase_hook(nabu_phase_hook_t fn, void *user);
/* Bound of every in-kernel wait of the persistent recurrent kernels (process-wide setting;
 * default 200 000 us, us <= 0 restores it).  A wait that runs out sets the status word (first
 * int32 of ws: 4*block + {1 forward, 2 backward, 3 start-up handshake}), every workgroup leaves at
 * its next barrier, later launches on that ws return at once until the caller has read and
 * cleared the word: the results of the step are invalid and the host must raise. */
int nabu_persist_set_timeout_us(long long us);

/* ops.pyramid_stack (nabu/neuralnetworks/components/ops.py:6-60) when T is not
 * a multiple of numsteps: y [B,Tp,F] = x [B,T,F] zero-padded in time (for T a
 * multiple the stack is a free view on the batch-major buffer).  The inverse
 * (gradient) drops the padded frames. */
int nabu_pad_time_f32(int B, int T, int Tp, int F, const float *x, float *y,
                      nabu_stream_t stream);
int nabu_unpad_time_f32(int B, int T, int Tp, int F, const float *y, float *x,
                        nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * CTC loss and gradient — loss_functions.CTC
 * (nabu/neuralnetworks/trainers/loss_functions.py:180-214): tf.nn.ctc_loss on
 * batch-major logits with blank = C-1, softmax inside, frames >= logit_len
 * ignored.  labels [B,Lmax] zero padded, label_len [B].
 *   nll [B]                     per-utterance -log p(labels | logits)
 *   dlogits [B,T,C]             grad_scale * d nll[b] / d logits  (0 past len)
 *   status [1] int32 (device)   set to 1+b if utterance b has no valid
 *                               alignment (TF raises in that case)
 * ws >= nabu_ctc_ws_bytes(B,T,Lmax). */
size_t nabu_ctc_ws_bytes(int B, int T, int Lmax);
int nabu_ctc_loss_grad(int B, int T, int C, int Lmax, const float *logits,
                       const int32_t *logit_len, const int32_t *labels,
                       const int32_t *label_len, float grad_scale, float *nll,
                       float *dlogits, int32_t *status, void *ws, size_t ws_bytes,
                       nabu_stream_t stream);

/* Masked sparse softmax cross-entropy averaged over the target length —
 * loss_functions.average_cross_entropy / cross_entropy
 * (nabu/neuralnetworks/trainers/loss_functions.py:78-109,155-165):
 *   loss[b]    = sum_{t<logit_len[b]} xent(logits[b,t], targets[b,t]) / target_len[b]
 *   dlogits    = grad_scale * d loss[b] / d logits   (0 for t >= logit_len[b])
 * logits [B,L,C]; targets [B,ldt] int32 with ldt >= L. */
int nabu_xent_loss_grad(int B, int L, int C, int ldt, const float *logits,
                        const int32_t *targets, const int32_t *logit_len,
                        const int32_t *target_len, float grad_scale, float *loss,
                        float *dlogits, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * Speller decoder step kernels — RNNDecoder._decode / Speller.create_cell
 * (nabu/neuralnetworks/models/ed_decoders/rnn_decoder.py:13-82, speller.py:13-69):
 * tf.contrib.rnn.LSTMCell inside tf.contrib.seq2seq.AttentionWrapper driven by
 * dynamic_decode(impute_finished=True).  The dense products of a step run on
 * nabu_gemm_f32; these are the fused non-GEMM parts.
 *
 * nabu_lstm_cell_fwd: z [B,4U] = (dense part of) [inputs, h]·kernel; adds bias [4U]
 *   and, when emb_rows != NULL, row ids[b] of emb_rows [C,4U] (the one-hot input
 *   times the kernel is a row gather); gate order i,j,f,o, forget bias +1;
 *   acts [B,4U] keeps (i,g,f,o) for the gradient; rows with step >= seq_len[b] are
 *   finished: state copied through, acts = 0.
 * nabu_lstm_cell_bwd: dz [B,4U] from dh (+ dh2 if not NULL), dc_in and the saved
 *   acts / cell states; dc_out is the cell gradient handed to step-1. */
int nabu_lstm_cell_fwd(int B, int U, int step, const int32_t *seq_len, const float *z,
                       const float *bias, const float *emb_rows, const int32_t *ids,
                       const float *c_prev, const float *h_prev, float *acts,
                       float *c_new, float *h_new, nabu_stream_t stream);
int nabu_lstm_cell_bwd(int B, int U, int step, const int32_t *seq_len, const float *acts,
                       const float *c_new, const float *c_prev, const float *dh,
                       const float *dh2, const float *dc_in, float *dz, float *dc_out,
                       nabu_stream_t stream);

/* Fused attention step — attention.factory 'vanilla' (tf BahdanauAttention) and
 * 'location_aware' (nabu/neuralnetworks/components/attention.py:6-39,90-240):
 *   score[b,t] = sum_u v[u] * tanh(keys[b,t,u] + q[b,u] (+ f[b,t,u]))
 *   f = conv1d(align_prev, conv_kernel [K,F], 'same') · conv_proj [F,U]   (location)
 *   align = softmax over t < enc_len[b] (score mask -inf), ctx = align^T · values
 * keys [B,Te,U] (= values·memory_kernel, one GEMM per batch), values [B,Te,E]
 * (rows >= enc_len are zero), q [B,U] (= h·query_kernel).  Rows with
 * step >= dec_len[b] are finished: align/ctx copy align_prev/ctx_prev.
 * The backward kernel recomputes tanh, ACCUMULATES into dkeys [B,Te,U] and into the
 * partials dv_part [B*S,U], dconv_proj_part [B*S,F,U], dconv_kernel_part
 * [B,K,F] (the caller zeroes them before the first step and column-sums them
 * afterwards; S below), writes dq [B,U] and dalign_out [B,Te] (gradient w.r.t. align_prev;
 * location only).  dalign_in (may be NULL) is the gradient that reaches this
 * step's alignments through the next step's location features.
 * probability_fn: alignments = softmax(score) | sigmoid(score) | sigmoid(score) / sum_t sigmoid(score)
 * over the unmasked frames (masked frames get 0).  normalized_sigmoid keeps its normaliser per
 * utterance in znorm [B] (written by the forward kernel, read by the backward kernel; NULL otherwise).
 * 'windowed' (WindowedAttention, attention.py:294-396) is the vanilla score restricted to the
 * frames [m - left - 1, m + right), m = the first frame at which the cumulated PREVIOUS alignment
 * exceeds 0.5 (Te if none does); the initial alignment is one-hot at frame 0 (the decoder drivers
 * set it).  The window is a constant of the step: the backward kernel is the vanilla one. */
typedef struct {
  uint32_t size;       /* sizeof(nabu_attn_desc) */
  int32_t B, Te, E, U;
  int32_t kind;        /* 0 = vanilla (Bahdanau), 1 = location_aware, 2 = windowed */
  int32_t K, F;        /* location_aware: filtersize, numfilt; windowed: left_window_width, right_window_width */
  int32_t prob_fn;     /* probability_fn (attention.py:9-13): 0 softmax, 1 sigmoid, 2 normalized_sigmoid */
} nabu_attn_desc;
int nabu_attn_fwd(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                  const int32_t *enc_len, const float *keys, const float *values,
                  const float *q, const float *v, const float *conv_kernel,
                  const float *conv_proj, const float *align_prev, const float *ctx_prev,
                  float *align, float *ctx, float *znorm, void *ws, size_t ws_bytes,
                  nabu_stream_t stream);
/* bytes of ws the forward pass needs (0 when the batch alone fills the chip): small batches are cut
 * into slices of encoder frames like the backward pass, partial softmax sums / contexts meet in ws */
size_t nabu_attn_fwd_ws_bytes(const nabu_attn_desc *d);
/* The backward pass cuts every utterance into S = nabu_attn_bwd_slices(d) slices of encoder frames
 * (one workgroup each, so that a step fills the chip at small batch sizes): dv_part and
 * dconv_proj_part have B*S rows; ctx is THIS step's context (the forward output: it turns the
 * softmax's sum over all frames into a dot product, which is what makes the slices independent);
 * ws (nabu_attn_bwd_ws_bytes) holds the per-slice dq and the d location features. */
int nabu_attn_bwd_slices(const nabu_attn_desc *d);
size_t nabu_attn_bwd_ws_bytes(const nabu_attn_desc *d);
int nabu_attn_bwd(const nabu_attn_desc *d, int step, const int32_t *dec_len,
                  const int32_t *enc_len, const float *keys, const float *values,
                  const float *q, const float *v, const float *conv_kernel,
                  const float *conv_proj, const float *align_prev, const float *align,
                  const float *ctx, const float *dctx, const float *dalign_in, float *dq,
                  float *dkeys, float *dv_part, float *dconv_proj_part,
                  float *dconv_kernel_part, float *dalign_out, const float *znorm, void *ws,
                  size_t ws_bytes, nabu_stream_t stream);

/* Whole-sequence decoder driver: RNNDecoder._decode over all L = max(dec_len)
 * steps in one call (rnn_decoder.py:59-82: ScheduledEmbeddingTrainingHelper,
 * BasicDecoder, dynamic_decode(impute_finished=True)),
 * i.e. per step: recurrent GEMMs + nabu_lstm_cell_fwd per layer (optional output
 * dropout, speller.py:38-43), query GEMM, nabu_attn_fwd (with sample_prob > 0 also the step's
 * projection + nabu_sample_ids for the next input); then ONE projection GEMM for
 * all steps (rnn_cell.py:145-155).  nabu_speller_bwd is its gradient; weight
 * gradients that are sums over steps are single GEMMs over all steps.
 *   values [B,Te,E] encoder output (rows >= enc_len zero), ids [L,B] decoder input
 *   labels (row 0 = C-1, then the targets shifted by one), logits [B,L,C].
 * With one layer, softmax attention (vanilla; forward also location-aware), no dropout / sampling and B = 32 (forward:
 * or 64) the L steps of the forward and of the backward pass are ONE persistent launch each (speller_persist.hip; NABU_SPELLER_PERSIST=0 /
 * NABU_SPELLER_PERSIST_BWD=0 keep the per-step kernels).
 * Its in-kernel waits are bounded like those of the recurrent layers: ws[0] (int32) is its status word,
 * 0 = ok, sticky — provide the workspace zero-initialised once; a non-zero word means a launch gave up
 * (results invalid) and makes later launches on that workspace return at once.
 * Kernel layouts are TF's: lstm_kernel[0] [(C+E+U),4U] (rows: one-hot, context, h),
 * lstm_kernel[n>0] [(2U),4U], memory_kernel [E,U], query_kernel [U,U], attention_v [U],
 * conv_kernel [K,F], conv_proj [F,U], out_kernel [(U+E),C] (rows: h, context), out_bias [C]. */
#define NABU_SPELLER_MAX_LAYERS 4
typedef struct {
  uint32_t size;
  int32_t B, Te, E, U, C, L, num_layers;
  int32_t kind, K, F, prob_fn;        /* attention: see nabu_attn_desc */
  float keep_prob;                    /* output dropout of every LSTM layer; 1 = off */
  unsigned long long seed, seed_offset;
  float sample_prob;                  /* scheduled sampling probability (speller.cfg sample_prob); 0 = off */
  unsigned long long sample_seed, sample_offset;
} nabu_speller_desc;
typedef struct {
  const float *memory_kernel, *query_kernel, *attention_v, *conv_kernel, *conv_proj, *out_kernel, *out_bias;
  const float *lstm_kernel[NABU_SPELLER_MAX_LAYERS], *lstm_bias[NABU_SPELLER_MAX_LAYERS];
} nabu_speller_params;
typedef struct {
  float *memory_kernel, *query_kernel, *attention_v, *conv_kernel, *conv_proj, *out_kernel, *out_bias;
  float *lstm_kernel[NABU_SPELLER_MAX_LAYERS], *lstm_bias[NABU_SPELLER_MAX_LAYERS];
} nabu_speller_grads;
/* Scheduled sampling — tf.contrib.seq2seq.ScheduledEmbeddingTrainingHelper as used by
 * rnn_decoder.py:59-66: after step t, row b draws select ~ Bernoulli(sample_prob); if selected
 * the decoder input of step t+1 is a sample from Categorical(softmax(logits_t[b])) instead of
 * the target y_t (no gradient flows through the sample).
 *   out_ids[b] = select ? sample : teacher_ids[b];  logits [B,C];
 * Philox4x32-10, counter (b, offset), key seed: a pure function of its arguments. */
int nabu_sample_ids(int B, int C, const float *logits, float prob, unsigned long long seed,
                    unsigned long long offset, const int32_t *teacher_ids, int32_t *out_ids,
                    nabu_stream_t stream);
/* The decoder inputs actually used by the last nabu_speller_fwd on `reserve` ([L,B] int32,
 * equal to `ids` when sample_prob == 0), copied to out_ids (device). */
int nabu_speller_decoder_inputs(const nabu_speller_desc *d, const void *reserve, int32_t *out_ids,
                                nabu_stream_t stream);
size_t nabu_speller_reserve_bytes(const nabu_speller_desc *d);
size_t nabu_speller_ws_bytes(const nabu_speller_desc *d);
/* 1 when the decoder steps of this descriptor run as ONE persistent launch (speller_persist.hip) in the forward
 * (backward = 0) resp. backward pass on the current device, 0 when they take the step chain.  Diagnostic: the two
 * paths compute the same function (tests/test_hip_speller.py compares them). */
int nabu_speller_uses_persistent(const nabu_speller_desc *d, int backward);
int nabu_speller_fwd(const nabu_speller_desc *d, const float *values, const int32_t *enc_len,
                     const int32_t *ids, const int32_t *dec_len, const nabu_speller_params *p,
                     float *logits, void *reserve, void *ws, size_t ws_bytes, nabu_stream_t stream);
/* every gradient buffer is overwritten; dvalues [B,Te,E] */
int nabu_speller_bwd(const nabu_speller_desc *d, const float *values, const int32_t *enc_len,
                     const int32_t *ids, const int32_t *dec_len, const nabu_speller_params *p,
                     const float *dlogits, void *reserve, const nabu_speller_grads *g,
                     float *dvalues, void *ws, size_t ws_bytes, nabu_stream_t stream);

/* x[b,t,:] = 0 for t >= len[b] (dynamic_decode zeroes the outputs of finished rows). */
int nabu_mask_time_f32(int B, int L, int F, float *x, const int32_t *len, nabu_stream_t stream);
/* y[b,l,:] = x[l,b,:] (time-major per-step buffers <-> the batch-major API). */
int nabu_swap01_f32(int L, int B, int F, const float *x, float *y, nabu_stream_t stream);
/* dK[c,:] = sum_{i: ids[i]==c} dz[i,:]  (gradient of the kernel rows gathered by the
 * one-hot decoder inputs), ids [N], dz [N,W], dK [C,W]; deterministic. */
int nabu_scatter_rows_f32(int C, int N, int W, const int32_t *ids, const float *dz, float *dK,
                          nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * Fused per-element gradient clip + TF-style Adam on flat buffers —
 * Trainer._update (nabu/neuralnetworks/trainers/trainer.py:525,560-569):
 *   g = clamp(grad_scale*grad, -clip, clip); m = b1 m + (1-b1) g;
 *   v = b2 v + (1-b2) g^2; param -= lr_t * m / (sqrt(v) + eps)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller. */
int nabu_adam_clip_step(size_t n, float *param, const float *grad, float *m,
                        float *v, float lr_t, float b1, float b2, float eps,
                        float clip, float grad_scale, nabu_stream_t stream);
/* g = clamp(g, -clip, clip) in place (data-parallel mode clips per replica
 * BEFORE the all-reduce, trainer.py:556-569). */
int nabu_clip_f32(size_t n, float *g, float clip, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * Regularisation noise (Philox4x32-10, counter = element index, key = seed):
 *   dropout: y = x * mask / keep_prob, mask ~ Bernoulli(keep_prob) —
 *     tf.nn.dropout(x, keep_prob) in Listener/DBLSTM/Speller
 *     (models/ed_encoders/listener.py:57-59,67-69; dblstm.py:52-54).  The
 *     same (seed, offset) regenerates the mask, so the gradient is the same
 *     call applied to dy.
 *   gaussian noise: y = x + stddev * N(0,1) — input_noise
 *     (listener.py:40-45, dblstm.py:37-42). */
int nabu_dropout_f32(size_t n, const float *x, float *y, float keep_prob,
                     unsigned long long seed, unsigned long long offset,
                     nabu_stream_t stream);
int nabu_gaussian_noise_f32(size_t n, const float *x, float *y, float stddev,
                            unsigned long long seed, unsigned long long offset,
                            nabu_stream_t stream);

/* out[0] = scale * sum(x) (deterministic single-block tree): tf.reduce_mean of
 * the per-utterance losses (trainers/loss_functions.py:161,206-212). */
int nabu_sum_f32(size_t n, const float *x, float scale, float *out, nabu_stream_t stream);
/* y += a*x : gradient accumulation where a tensor has several consumers. */
int nabu_axpy_f32(size_t n, float a, const float *x, float *y, nabu_stream_t stream);

/* out[i] = ceil(in[i] / d): the sequence lengths after ops.pyramid_stack
 * (nabu/neuralnetworks/components/ops.py:56-58), computed where the next layer reads them
 * (a host-side recomputation + upload would synchronise the stream once per layer). */
int nabu_ceil_div_i32(int n, const int32_t *in, int d, int32_t *out, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * DNNDecoder hidden layers (models/ed_decoders/dnn_decoder.py:40-51):
 * tf.contrib.layers.fully_connected = linear + ReLU, optional
 * tf.contrib.layers.layer_norm, dropout.  The linear part is nabu_gemm_f32.
 *   relu:      y = max(x, 0);  relu_bwd: dx = y > 0 ? dy : 0   (in place allowed)
 *   layer_norm (TF-1.8 contrib defaults begin_norm_axis=1, begin_params_axis=-1,
 *   variance_epsilon 1e-12): for a [B,T,F] input the moments are taken over ALL of
 *   (T,F) per batch row — padded frames included — and gamma/beta are [F]:
 *     y[b,n] = (x[b,n] - mean_b) * rstd_b * gamma[n % F] + beta[n % F],  n < N = T*F
 *   fwd saves mean[B], rstd[B]; bwd writes dx and the per-row partial sums
 *   dgamma_part/dbeta_part [B,F] (reduce them with nabu_colsum_f32; deterministic). */
int nabu_relu_f32(size_t n, const float *x, float *y, nabu_stream_t stream);
int nabu_relu_bwd_f32(size_t n, const float *y, const float *dy, float *dx, nabu_stream_t stream);
int nabu_layer_norm_fwd(int B, int N, int F, const float *x, const float *gamma, const float *beta,
                        float eps, float *y, float *mean, float *rstd, nabu_stream_t stream);
int nabu_layer_norm_bwd(int B, int N, int F, const float *x, const float *gamma, const float *dy,
                        const float *mean, const float *rstd, float *dx, float *dgamma_part,
                        float *dbeta_part, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * Inference decoders — nabu/neuralnetworks/decoders.
 *
 * nabu_ctc_beam_search: CTCDecoder.__call__ (decoders/ctc_decoder.py:44-68) =
 * tf.nn.ctc_beam_search_decoder(logits, seq_len) with its defaults top_paths = 1,
 * beam_width = 100, merge_repeated = 1: CTC prefix beam search (prefix tree with per-node
 * (total, blank, label) log-probabilities, bounded beam of the beam_width best prefixes per
 * frame, blank = C-1), one workgroup per utterance over all its frames; merge_repeated also
 * collapses equal neighbours of the OUTPUT labelling, as TF's BeamEntry::LabelSeq does.
 *   logits [B,T,C] batch-major (softmax is taken inside), logit_len [B];
 *   out_ids [B,T] (entries >= out_len are -1), out_len [B], out_logprob [B] (may be NULL):
 *   log P(best prefix) under the per-frame normalised posteriors.
 * The workspace holds the prefix trees; it is initialised by the call. */
size_t nabu_ctc_beam_ws_bytes(int B, int T, int C, int beam_width);
int nabu_ctc_beam_search(int B, int T, int C, int beam_width, int merge_repeated, const float *logits,
                         const int32_t *logit_len, int32_t *out_ids, int32_t *out_len,
                         float *out_logprob, void *ws, size_t ws_bytes, nabu_stream_t stream);

/* tf.edit_distance(hyp, truth, normalize=False) per utterance (ctc_decoder.py:112-118,
 * decoders/beam_search_decoder.py:176-179): Levenshtein distance of hyp[b,:hyp_len[b]] and
 * truth[b,:truth_len[b]] (negative lengths count as 0); dist [B] int32. */
int nabu_edit_distance(int B, const int32_t *hyp, int ldh, const int32_t *hyp_len, const int32_t *truth,
                       int ldt, const int32_t *truth_len, int32_t *dist, nabu_stream_t stream);

/* Attention beam search — BeamSearchDecoder.__call__ (decoders/beam_search_decoder.py:31-114)
 * over components/beam_search_decoder.py:141-451 driven by dynamic_decode(maximum_iterations =
 * max_steps): the encoder output is tiled beam_width times, every step runs the Speller cell
 * (same kernels as nabu_speller_fwd) on B*beam_width rows, then
 *   nabu_beam_prune: candidates = every (beam, label) expansion plus one "stay" hypothesis per
 *     finished beam; log-softmax(logits / temperature); score = logprob / ((5+len)/6)^length_penalty;
 *     the beam_width best by score (ties: lower candidate index, as tf.nn.top_k) survive;
 *   nabu_beam_gather: survivors take the new cell state of their parent, "stay" hypotheses keep
 *     the state they had.
 * The search ends when every beam slot has been finished at some step (dynamic_decode ORs
 * `finished` over steps: the reference's decoder does not track it itself) or after max_steps.
 * Then the backwards search through the parent pointers (finalize, :341-451).
 *   values [B,Te,E] (rows >= enc_len zero), params as nabu_speller_fwd;
 *   sequences [B,W,max_steps] int32 (valid [:, :, :*num_steps]), lengths [B,W] (labels before
 *   the end token C-1), scores [B,W], alignments [B,W,max_steps,Te] or NULL.
 * SYNCHRONISES the stream once per step (the stop test) — an inference-time API.
 * Where the reference derives the parent slot of a "stay" hypothesis as idx % C (valid while
 * beam_width <= C), the slot idx - beam_width*C is used, so wider beams also work. */
typedef struct {
  uint32_t size;
  int32_t B, Te, E, U, C, num_layers;
  int32_t kind, K, F, prob_fn;        /* attention: see nabu_attn_desc */
  int32_t beam_width, max_steps;
  float length_penalty, temperature;
} nabu_beam_desc;
size_t nabu_speller_beam_ws_bytes(const nabu_beam_desc *d);
int nabu_speller_beam_search(const nabu_beam_desc *d, const float *values, const int32_t *enc_len,
                             const nabu_speller_params *p, int32_t *sequences, int32_t *lengths,
                             float *scores, float *alignments, int32_t *num_steps, void *ws,
                             size_t ws_bytes, nabu_stream_t stream);
/* One pruning step on its own (testable without a model).  logits [B*W,C]; logprobs, lengths,
 * finished, seen [B,W] are updated in place; pred_ids, parent, stay [B,W] and all_seen [B] are
 * written; scratch [B, W*C+W] floats. */
int nabu_beam_prune(int B, int W, int C, const float *logits, float temperature, float length_penalty,
                    float *logprobs, int32_t *lengths, int32_t *finished, int32_t *seen,
                    int32_t *pred_ids, int32_t *parent, int32_t *stay, int32_t *all_seen,
                    float *scratch, nabu_stream_t stream);
/* dst[b,w,:] = (stay[b,w] ? old : fresh)[b, parent[b,w], :]  for [B,W,F] float rows */
int nabu_beam_gather(int B, int W, int F, const float *fresh, const float *old, const int32_t *parent,
                     const int32_t *stay, float *dst, nabu_stream_t stream);

/* ------------------------------------------------------------------------
 * Host-side helper of the on-disk data path (the only entry point that takes HOST memory):
 * CRC-32C (Castagnoli) of `n` bytes, continuing from `crc` (0 to start) — the checksum of the
 * TFRecord framing of the reference's per-utterance files.
 * Replaces: tf.python_io.TFRecordWriter / tf.TFRecordReader record checks
 * (nabu/processing/tfwriters/tfwriter.py:30-45, nabu/processing/tfreaders/tfreader.py:71-92). */
uint32_t nabu_crc32c_host(const void *data_host, size_t n, uint32_t crc);

#ifdef __cplusplus
}
#endif
#endif /* NABU_HIP_H */
