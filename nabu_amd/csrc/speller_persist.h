// speller_persist.h — the Speller's decoder steps as ONE persistent launch (internal API, forward pass).
#pragma once
#include "common.h"

namespace nabu {

struct SpPersistDesc {
  int B, L, U, E, Te, C;
  int kind = 0, K = 0, F = 0;      // attention: 0 vanilla, 1 location-aware (filter taps, filters); softmax
};

// shapes the persistent forward kernel takes (single LSTM layer, vanilla softmax attention, no dropout, no
// scheduled sampling are checked by the caller): B = 32 or 64 (two launches of 32 rows), U and E multiples of 32, the
// keys slice of a workgroup (and the location-aware filters) fit the LDS; the values slice too, or it is streamed
bool speller_persist_ok(const SpPersistDesc &d);
size_t speller_persist_ws_bytes(const SpPersistDesc &d);

// kperm: [(E+U), 4U] gate-interleaved dense rows of the cell kernel (column 4u+g); emb: the kernel's first C rows
// (gate-major columns g*U+u); bias [4U] gate-major; wq [U,U]; v [U]; keys [B,Te,U]; values [B,Te,E]; ids [L,B].
// Writes the time-major reserve arrays of nabu_speller_fwd: H, Cs [(L+1),B,U] (index 0 = zero state, set by the
// caller), acts [L,B,4U], q [L,B,U], ctx [(L+1),B,E], align [(L+1),B,Te].
int speller_persist_fwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const int32_t *ids,
                        const float *kperm, const float *bias, const float *emb, const float *wq, const float *v,
                        const float *keys, const float *values, const float *conv_kernel, const float *conv_proj, float *H,
                        float *Cs, float *acts, float *q, float *ctx, float *align, int *status, void *ws, size_t ws_bytes,
                        hipStream_t stream);

// Backward pass of the step loop (same shapes; the caller has run the output projection's gradient into dH / dCtx).
// kxhT [4U, E+U]: transposed dense rows of the cell kernel (k = gate-major column).  Writes dq [L,B,U], dz [L,B,4U]
// (gate-major), dkeys [B,Te,U] (overwritten), dv_part [B*8,U] (one row per utterance and frame slice), and adds the
// carried d context into dCtx [L,B,E].
bool speller_persist_bwd_ok(const SpPersistDesc &d);
size_t speller_persist_bwd_ws_bytes(const SpPersistDesc &d);
int speller_persist_bwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const float *kxhT,
                        const float *wq, const float *v, const float *keys, const float *values, const float *acts,
                        const float *Cs, const float *q, const float *ctx, const float *align, const float *dH, float *dCtx,
                        float *dq, float *dz, float *dkeys, float *dv_part, int *status, void *ws, size_t ws_bytes,
                        hipStream_t stream);

}  // namespace nabu
