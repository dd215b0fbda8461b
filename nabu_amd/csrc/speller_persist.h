// speller_persist.h — the Speller's decoder steps as ONE persistent launch (internal API, forward pass).
#pragma once
#include "common.h"

namespace nabu {

struct SpPersistDesc {
  int B, L, U, E, Te, C;
  int kind = 0, K = 0, F = 0;      // attention: 0 vanilla, 1 location-aware (filter taps, filters); softmax
  // output dropout of the cell (speller.py:36-40: DropoutWrapper(output_keep_prob)): the mask of step t is the Philox
  // stream dropout_rows draws for (seed, seed_offset + t) over [B, U]; 1 = off
  float keep_prob = 1.f;
  unsigned long long seed = 0, seed_offset = 0;
  float *drop_scale = nullptr;     // [L, B, U] scratch (keep_prob < 1): the scale factors 0 | 1 / keep of every step, WRITTEN
                                   // by either pass in one small launch in front of its decoder kernel
  // scheduled sampling (rnn_decoder.py:59-66, ScheduledEmbeddingTrainingHelper): with probability sample_prob the
  // next input of a row is drawn from softmax(logits of this step) — the draws of nabu_sample_ids for
  // (sample_seed, sample_offset + t, batch row); 0 = teacher forcing
  float sample_prob = 0.f;
  unsigned long long sample_seed = 0, sample_offset = 0;
  unsigned *sample_draws = nullptr;   // [L, B, 2] (sample_prob > 0): the two Philox words of every (step, row), written likewise
};

// shapes the persistent forward kernel takes (single LSTM layer, vanilla softmax attention, no dropout, no
// scheduled sampling are checked by the caller): B = 32 or 64 (two launches of 32 rows), U and E multiples of 32, the
// keys slice of a workgroup (and the location-aware filters) fit the LDS; the values slice too, or it is streamed
bool speller_persist_ok(const SpPersistDesc &d);
bool speller_persist_streams_values(const SpPersistDesc &d);
size_t speller_persist_ws_bytes(const SpPersistDesc &d);

// kperm: [(E+U), 4U] gate-interleaved dense rows of the cell kernel (column 4u+g); emb: the kernel's first C rows
// (gate-major columns g*U+u); bias [4U] gate-major; wq [U,U]; v [U]; keys [B,Te,U]; values [B,Te,E]; ids [L,B].
// Writes the time-major reserve arrays of nabu_speller_fwd: H, Cs [(L+1),B,U] (index 0 = zero state, set by the
// caller), acts [L,B,4U], q [L,B,U], ctx [(L+1),B,E], align [(L+1),B,Te]; with d.keep_prob < 1 also Ho [(L+1),B,U],
// the dropped cell outputs (what the query and the output projection see; the recurrence keeps h).
int speller_persist_fwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const int32_t *ids,
                        const float *kperm, const float *bias, const float *emb, const float *wq, const float *v,
                        const float *keys, const float *values, const float *conv_kernel, const float *conv_proj, float *H,
                        float *Ho, float *Cs, float *acts, float *q, float *ctx, float *align, int *status, void *ws,
                        size_t ws_bytes, hipStream_t stream, const float *out_kernel = nullptr,
                        const float *out_bias = nullptr, int32_t *ids_used = nullptr);
// (scheduled sampling: out_kernel [(U+E), C], out_bias [C] = the output projection; ids_used [L,B] = `ids`, whose
// rows 1.. the kernel overwrites with the inputs it actually used)

// Backward pass of the step loop (same shapes, B = 32 or 64; the caller has run the output projection's gradient into
// dH / dCtx).
// kxhT [4U, E+U]: transposed dense rows of the cell kernel (k = gate-major column).  Writes dq [L,B,U], dz [L,B,4U]
// (gate-major), dkeys [B,Te,U] (overwritten), dv_part [B*8,U] (one row per utterance and frame slice), and adds the
// carried d context into dCtx [L,B,E].
bool speller_persist_bwd_ok(const SpPersistDesc &d);
size_t speller_persist_bwd_ws_bytes(const SpPersistDesc &d);
int speller_persist_bwd(const SpPersistDesc &d, const int32_t *dec_len, const int32_t *enc_len, const float *kxhT,
                        const float *wq, const float *v, const float *keys, const float *values, const float *acts,
                        const float *Cs, const float *q, const float *ctx, const float *align, const float *dH, float *dCtx,
                        float *dq, float *dz, float *dkeys, float *dv_part, int *status, void *ws, size_t ws_bytes,
                        hipStream_t stream, const float *conv_kernel = nullptr, const float *conv_proj = nullptr,
                        float *ds_all = nullptr, float *cf_all = nullptr, float *dck_part = nullptr);
// location-aware attention (d.kind = 1): conv_kernel [K, F], conv_proj [F, U]; the kernel then writes the steps'
// d scores ds_all [L, B, Te] and location features cf_all [L, B, Te, F] — the inputs of attn_param_grads_kernel
// (speller.hip), which produces d keys / d attention_v / d conv_proj; dkeys and dv_part are NOT written — and the
// conv kernel's gradient as one partial row per (utterance, frame slice): dck_part [B * 8, K * F].

}  // namespace nabu
