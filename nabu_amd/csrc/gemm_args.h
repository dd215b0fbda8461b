// gemm_args.h — argument block shared by the fp32-MFMA and the bf16-split-MFMA GEMM kernels.
#pragma once
#include "common.h"

namespace nabu {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 16, LDT = 132;
constexpr int FBK = 32;   // k-tile of the fast kernels

struct GemmArgs {
  const float *A, *B, *bias;
  float *C;
  float *partial;
  int M, N, K, lda, ldb, ldc;
  float alpha, beta;
  int kseg;
  long long a_seg, b_seg;
  int ksplit;  // k-range per z-slice (multiple of FBK)
  int nsplit;
  int vecA, vecB;  // 16-byte vector loads allowed
  int swz;         // tile order: 0 = blockIdx as is, g > 0 = XCD-aware with groups of g row tiles
  // batched products (generic kernel only): blockIdx.z = batch index, operands advance by these strides
  int nbatch;
  long long a_bs, b_bs, c_bs;
};

// Workgroup -> output tile.  The dispatcher deals workgroups round-robin over the 8 XCDs (linear id
// % 8), each with its own 4 MiB L2.  With swz > 0 every XCD walks a CONTIGUOUS range of the tile
// sequence, and the sequence itself goes column-major through bands of `swz` row tiles, so that the
// ~96 workgroups resident on an XCD share few operand panels (each panel is then fetched into that
// L2 once instead of once per XCD).
__device__ __forceinline__ void tile_of_block(int swz, int *tm, int *tn) {
  const int nbx = gridDim.x, nby = gridDim.y;
  if (swz <= 0) { *tm = blockIdx.y; *tn = blockIdx.x; return; }
  const int total = nbx * nby, pid = blockIdx.y * nbx + blockIdx.x;
  const int per = (total + 7) / 8, tall = total % 8 ? total % 8 : 8;
  const int xcd = pid % 8, local = pid / 8;
  const int t = xcd < tall ? xcd * per + local : tall * per + (xcd - tall) * (per - 1) + local;
  const int band = swz * nbx, g = t / band, first = g * swz;
  const int rows = min(nby - first, swz), r = t - g * band;
  *tm = first + r % rows;
  *tn = r / rows;
}

// skinny products (gemm_skinny.hip): M <= 64, no transposes
int gemm_skinny_chunk(int M, int N, int K);
int gemm_skinny_launch(const GemmArgs &a, hipStream_t stream);
// decoder-step variant: two operand pairs along the reduction, split-K reduce finished in the launch,
// optionally followed by an LSTM-cell epilogue on the finished tile (see gemm_skinny.hip)
// Epilogues of the decoder-step product (kind): 0 = write C; 1 = LSTM cell forward on a tile of z whose
// columns are GATE-INTERLEAVED (column 4u+g = gate g of unit u: the product runs against a column-permuted
// copy of the TF kernel, so that a 32-column slice holds all four gates of 8 units); 2 = LSTM cell backward
// on a tile of dh (the product dq·Wq^T + beta*dH, natural unit order).  Folding the cell into the product's
// last workgroup removes one dependent kernel (launch + memory round trip) from every decoder step.
// Same arithmetic as lstm_cell_fwd/_bwd_kernel (speller.hip).
struct SkinnyEpilogue {
  int kind, U, step;
  const int32_t *seq_len;
  // forward
  const float *bias, *emb;
  const int32_t *ids;
  const float *c_prev, *h_prev;
  float *acts, *c_new, *h_new;
  float *ho_new;   // forward, rows16_kernel only (may be null): the cell output behind its dropout (keep, seed, seed_offset,
                   // row0 below: element (row0 + m) U + u of the stream, the mask dropout_rows would apply to h_new)
  // backward (acts, c_new = c of this step, c_prev as above are inputs)
  const float *dh2;
  int ld_dh2;
  const float *dc_in;
  float *dz, *dc_out;
  // backward, rows16_kernel only: output dropout of the cell (0 or 1 = off): the gradient of the cell OUTPUT (projection +
  // query) goes through mask / keep, the recurrent carry dh2 does not; element (row0 + m) U + u of the Philox stream
  // (seed, seed_offset) — dropout_scale4 of common.h, the masks of the forward pass
  float keep;
  unsigned long long seed, seed_offset;
  int row0;
};

// optional second destination of the fused skinny product: columns >= split (a multiple of 32) are written to
// C2[m * ldc2 + n - split] with their own beta (no epilogue kinds with it)
struct SkinnySplit {
  float *C2;
  int ldc2, split;
  float beta2;
};

// the decoder chain's products of a sub-batch of <= 16 rows (gemm_skinny.hip): weights re-blocked once per pass
bool rows16_ok(int M, int N, int K, int lda);
int rows16_swizzle(int N, int K, const float *W, int ldw, float *out, hipStream_t s);
int rows16_swizzle_kn(int N, int K, const float *W, int ldw, float *out, int gate_units, hipStream_t s);
// K1 / A2 / lda2: reduction indices >= K1 come from a second operand (the cell's [context | h]); ep kinds 0, 1 (forward
// cell: the weights' columns gate-interleaved, rows16_swizzle_kn(gate_units = U)), 2 (cell backward)
int rows16(int M, int N, int K, const float *A, int lda, const float *Wsw, float beta, float *C, int ldc, hipStream_t s,
           const SkinnyEpilogue *ep = nullptr, const SkinnySplit *split = nullptr, int K1 = 0, const float *A2 = nullptr,
           int lda2 = 0);
int gemm_skinny_fused(int M, int N, int K1, const float *A, int lda, const float *B, int ldb, int K2, const float *A2,
                      int lda2, const float *B2, int ldb2, float beta, float *C, int ldc, const float *bias,
                      float *partial, unsigned *tickets, hipStream_t s, const SkinnyEpilogue *ep = nullptr,
                      const SkinnySplit *split = nullptr);

// batch of nbatch independent products C_i (+)= op(A_i)·op(B_i) in ONE launch of the generic fp32 kernel
// (operand i starts i*stride elements after operand 0); no split-K
int gemm_batched_f32(bool transA, bool transB, int M, int N, int K, const float *A, int lda, long long a_bs,
                     const float *B, int ldb, long long b_bs, float beta, float *C, int ldc, long long c_bs, int nbatch,
                     hipStream_t s);

// split-K policy shared by the tile kernels: >= target workgroups, k-range a multiple of kmult
int gemm_split_for(int M, int N, int K, int kmult, int target, int *ksplit);
int gemm_splitk_reduce(const GemmArgs &a, hipStream_t s);

// bf16 operands resident in memory (gemm_bf16_pre.hip): C = alpha * A·B^T + beta*C + bias, A [M,lda] and
// B [N,ldb] bf16 with k contiguous; conversions fp32 -> bf16 (plain / transposed copies)
bool gemm_bf16_pre_ok(int M, int N, int K, int lda, int ldb);
size_t gemm_bf16_pre_ws_bytes(int M, int N, int K);
int gemm_bf16_pre(int M, int N, int K, float alpha, const unsigned short *A, int lda, const unsigned short *B, int ldb,
                  float beta, float *C, int ldc, const float *bias, void *ws, size_t ws_bytes, hipStream_t s);
int cvt_bf16(size_t R, int C, const float *src, int ld, unsigned short *dst, int ldd, hipStream_t s);
int cvt_bf16_t(int R, int C, const float *src, int ld, unsigned short *dst, int ldd, hipStream_t s);

// packed bf16-plane operands (gemm_pk.hip): does the current device offer the tile loop's 144 KiB of LDS?
bool gemm_pk_device_ok();
// f16x3 row maxima from per-unit partial maxima [rows][ld] (lstm_persist.hip): amax[n] = bits of max_r |part[r][n]|
int pk_amax_from_partials(int rows, int N, const float *part, int ld, uint32_t *amax, hipStream_t s);
// the same maxima AND the row maxima of dz as [BT, ...] (adz, rows_pad entries) out of the persistent backward kernel's
// per-workgroup row maxima rowpart[nparts][BT] (bit patterns): one launch, no read of dz (either output may be null)
int pk_amax_from_persist(int BT, int rows_pad, int T, int max_len, int nparts, const uint32_t *rowpart, uint32_t *adz,
                         int crow, int N, const float *colpart, int ld, uint32_t *adzT, hipStream_t s,
                         const float *sumpart = nullptr, float *db0 = nullptr, float *db1 = nullptr);
// (sumpart: [crow, ld] partial column sums -> db0 | db1 [N / 2] each, summed in row order: colsum_pair's result from the
//  launch that reads the maxima)
// row / column maxima (nabu_pk_amax) of TWO sources of one shape in one launch: rows shared, columns per source;
// cols_b (optional): a second array that receives the column maxima as well.  src1 = nullptr: one source.
int pk_amax_pair(const float *src0, const float *src1, long long ld, int R, int C, uint32_t *rows, uint32_t *cols0,
                 uint32_t *cols1, uint32_t *cols_b, hipStream_t s);
// up to 4 packs of one kind (natural / transposed, the arguments of nabu_pk_pack / nabu_pk_pack_f16) in ONE launch
struct PkPackReq {
  const float *src; long long ld; int R, C;
  void *dst; int rows_pad, row_off, kb_off, fill_rows, fill_kb, period, shift;
  const uint32_t *amax;      // planes = 2 only
};
int pk_pack_multi(int planes, int transposed, const PkPackReq *req, int n, hipStream_t s);
// natural AND transposed pack of one source in one pass
int pk_pack_both(int planes, const float *src, long long ld, int R, int C, void *dst_n, int rows_pad_n, int kb_off_n,
                 int fill_rows_n, int fill_kb_n, void *dst_t, int rows_pad_t, int row_off_t, int fill_rows_t,
                 int fill_kb_t, hipStream_t s, const unsigned *amax_n = nullptr, const unsigned *amax_t = nullptr);
// (planes = 2, f16x3: amax_n / amax_t = the row maxima of the two destinations, indexed by packed row)

// bf16-split kernels (gemm_bf16.hip); planes = 1 (bf16), 2 (bf16x3) or 3 (bf16x6)
int gemm_bf16_launch(const GemmArgs &a, bool transA, bool transB, int planes, dim3 grid, hipStream_t stream);

}  // namespace nabu
