// lstm_persist.h — whole-sequence persistent recurrent kernels (internal API).
#pragma once
#include "common.h"

namespace nabu {
// PACKED COMPANIONS written by the forward fp16-plane kernel next to `out` (include/nabu_hip.h, nabu_blstm_desc ABI version
// 3): the layer's output as f16x3 operands in gemm_pk.hip's layout [k / 16][plane][rows_pad][16 k] (halves of a 32-byte
// record swapped when bit 3 of the row is set), scaled by 2^14 (row maxima = the bit pattern of 1.0f).  Null pointer =
// that companion is not written.  Offsets inside one companion stay below 2^31 (checked by the caller).
struct EmitArgs {
  char *x_rows;            // out, `stack` frames per row: row = b (T / stack) + t / stack, k = (t % stack) 2H + dir H + unit
  char *x_cols;            // the transposed operand: row = that k, reduction index = that row
  char *hT[2];             // h_(t-1)^T of the forward / backward cell: row = hT_row0 + unit, reduction index = b T + t, shifted
                           // by one frame (fw: out[b, t - 1], bw: out[b, t + 1]; 0 at the ends)
  unsigned x_rows_pad, x_cols_pad, hT_rows_pad;
  int hT_row0;
  int stack_shift;         // log2(stack): 0 or 1
  int b0;                  // first batch row of this launch within the whole batch
};
// does a forward launch of this shape write companions itself? (fp16-plane kernels of lstm_persist_mxh.hip, launches of
// <= 32 rows, every frame visited)
bool lstm_persist_emits(int B, int T, int H, int max_len);
bool lstm_persist_supported(int B, int T, int H);
void lstm_persist_set_timeout_us(long long us);
unsigned long long lstm_persist_timeout_ticks();   // bound of every in-kernel wait (wall_clock64 ticks)
size_t lstm_persist_ws_bytes(int B, int T, int H);
int lstm_persist_fwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     float *out, int *status, void *ws, size_t ws_bytes, hipStream_t stream, const float *x = nullptr,
                     const float *const bias[2] = nullptr, void *xws = nullptr, const EmitArgs *emit = nullptr);
// xws: lstm_persist_xws_bytes(B, T, D) bytes of workspace for the fp16-plane kernels' copy of x (two fp16 planes per frame)
size_t lstm_persist_xws_bytes(int B, int T, int D, int H);
// x, bias given (only when lstm_persist_fuses_input says so: the fp16-plane kernels with D <= 64, one launch of <= 32 rows;
// the exact-fp32 kernels with D = 40 on the 4-row geometry): gates[] need NOT hold
// the input projection, the kernel computes x_t . Wx + b itself; it still leaves the activations there
bool lstm_persist_fuses_input(int B, int T, int D, int H);
int lstm_persist_bwd(int B, int T, int D, int H, int max_len, const int32_t *len,
                     const float *const kernel[2], float *const gates[2], float *const cs[2],
                     const float *dout, int *status, void *ws, size_t ws_bytes, float **db_part, int *db_rows,
                     hipStream_t stream, uint32_t *rowmax = nullptr, bool *rowmax_done = nullptr);
// rowmax (optional, [2 directions x H / 16][B T] uint32): every workgroup's largest |dz| (bit pattern) of every frame
// row over its 64 gate columns, written step by step next to dz; *rowmax_done says whether the kernels that ran keep
// it (the fp16-plane kernels of lstm_persist_mxh.hip do) — the row scales of dZ as an f16x3 operand without a pass over dz
// Per-thread switch (set by the blstm entry points from nabu_blstm_desc.recurrent_precision for the duration of a
// call): true = only the exact-fp32 kernels of lstm_persist.hip, whatever the shape
void lstm_persist_set_exact(bool exact);
bool lstm_persist_exact();
// db_part: [db_rows][2 directions][4H] bias-gradient partial sums written by the backward kernels
// (inside ws); bias gradient of direction d = column sums of db_part[:, d, :].  In the same layout, at
// db_part + lstm_persist_db_floats(B, H): the largest |dz| of every gate column per unit (column maxima of dz = the
// maximum over the rows; the row scales of the f16x3 weight-gradient products' dZ^T operand)
size_t lstm_persist_db_floats(int B, int H);
// One launch less in front of a forward recurrent launch: a caller that fills something anyway (lstm.hip: the maxima
// of the input projection) adds the region lstm_persist_ring_seg names to that fill and says so with
// lstm_persist_ring_cleared on the same host thread; the next launch on that workspace and stream then skips its own
// reset.  false: not offered for this shape (several launches share the ring).
struct FillSeg;
bool lstm_persist_ring_seg(bool fwd, int B, int T, int H, void *ws, FillSeg *seg);
void lstm_persist_ring_cleared(const FillSeg *seg, hipStream_t stream);
}  // namespace nabu
