// decode.hip — inference decoders: CTC prefix beam search, the attention
// beam search of the reference's BeamSearchDecoder, and the edit distance both evaluate with.
//
// These are latency-bound integer/control kernels, not bandwidth kernels: a decoding step handles
// a few thousand candidates per utterance.  The design goal is that nothing leaves the device
// between the model's forward pass and the decoded label sequences:
//   * CTC: ONE launch for the whole batch; a workgroup owns an utterance, walks its frames and
//     keeps the beam (<= beam_width prefixes with their blank/label/total log-probabilities) in
//     LDS; the prefix tree (parent, label, beam slot, children table) lives in an HBM workspace
//     that stays L2-resident.
//   * attention: the per-step cell kernels of speller.hip on B*beam_width rows, then one
//     pruning workgroup per utterance and row gathers of the cell state.
// Selecting the k best of n candidates, exact and deterministic with ties to the lower candidate
// index: CTC (k = 100 of ~4000 per frame) uses a radix select of the k-th largest key + compaction +
// rank sort (select_best); the attention search (k = 16) uses k rounds of a workgroup-wide arg-max
// in which every thread caches the best of its own strided share and only the owner of the removed
// candidate rescans.
#include <float.h>
#include <limits.h>

#include <vector>

#include "common.h"

namespace nabu {

constexpr int DT = 256;  // threads per workgroup of the decoding kernels

__device__ __forceinline__ float lse2d(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b), n = fminf(a, b);
  return m + log1pf(expf(n - m));
}

struct Best {
  float v;
  int i;
};
// larger value wins, equal values go to the smaller index; NaN never wins
__device__ __forceinline__ Best better(Best a, Best b) { return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a; }

// `red` holds two sets of per-wave results used alternately (parity = round & 1), so that one
// barrier per round suffices: round k+1 writes the other set while stragglers still read set k.
__device__ __forceinline__ Best block_best(Best x, Best *red, int parity) {
  red += parity * (DT / 64);
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    x = better(x, y);
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  Best r = red[0];
#pragma unroll
  for (int w = 1; w < DT / 64; ++w) r = better(r, red[w]);
  return r;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  return v;
}


// order-preserving map float -> uint32 (larger float = larger integer); -inf and NaN map to 0 = "dead"
__device__ __forceinline__ unsigned key_image(float x) {
  if (!(x > -INFINITY)) return 0u;
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

// The `want` best of keys[0..n) by (value desc, index asc), dead keys (-inf/NaN) never: writes their
// indices best-first to sel[] and returns how many there are (uniform).  One barrier-synchronised
// routine for the whole workgroup; scratch: hist[256], list_u/list_i[want], ties[n], sc[8].
__device__ int select_best(const float *keys, int n, int want, int *hist, unsigned *list_u, int *list_i,
                           int *ties, int *sel, int *sc) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned prefix = 0;
  int need = want;             // rank of the threshold key among the keys matching the prefix so far
  bool all_live = false;       // fewer live keys than `want`: take them all
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned himask = pass ? (0xFFFFFFFFu << (shift + 8)) : 0u;
    for (int i = tid; i < 256; i += DT) hist[i] = 0;
    __syncthreads();
    for (int idx = tid; idx < n; idx += DT) {
      const unsigned u = key_image(keys[idx]);
      if (u && (u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
      const int s4 = h0 + h1 + h2 + h3;
      int inc = s4;                                  // inclusive suffix sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_down(inc, o);
        if (lane + o < 64) inc += t;
      }
      if (lane == 0) sc[2] = inc;                    // keys matching the prefix
      int a = inc - s4;                              // keys in higher bins
      const int hh[4] = {h0, h1, h2, h3};
#pragma unroll
      for (int b = 3; b >= 0; --b) {
        if (a < need && need <= a + hh[b]) { sc[0] = 4 * lane + b; sc[1] = need - a; }
        a += hh[b];
      }
    }
    __syncthreads();
    if (sc[2] < need) { all_live = true; break; }    // only possible in pass 0 (uniform)
    prefix |= (unsigned)sc[0] << shift;
    need = sc[1];
    __syncthreads();
  }
  // compaction
  if (tid == 0) { sc[3] = 0; sc[4] = 0; }
  __syncthreads();
  const unsigned thr = all_live ? 0u : prefix;
  for (int idx = tid; idx < n; idx += DT) {
    const unsigned u = key_image(keys[idx]);
    if (!u) continue;
    if (u > thr) {
      const int pos = atomicAdd(&sc[3], 1);
      list_u[pos] = u; list_i[pos] = idx;
    } else if (u == thr) {
      ties[atomicAdd(&sc[4], 1)] = idx;
    }
  }
  __syncthreads();
  const int above = sc[3], nties = all_live ? 0 : sc[4];
  // of the keys equal to the threshold the `need` lowest indices survive
  for (int t = tid; t < nties; t += DT) {
    const int me = ties[t];
    int rank = 0;
    if (nties > need)
      for (int j = 0; j < nties; ++j) rank += ties[j] < me;
    else
      rank = 0;
    if (nties <= need || rank < need) {
      const int pos = atomicAdd(&sc[3], 1);
      list_u[pos] = thr; list_i[pos] = me;
    }
  }
  __syncthreads();
  const int m = sc[3];
  (void)above;
  // rank sort (keys descending, equal keys by ascending index)
  for (int t = tid; t < m; t += DT) {
    const unsigned u = list_u[t];
    const int i = list_i[t];
    int rank = 0;
    for (int j = 0; j < m; ++j) {
      const unsigned uj = list_u[j];
      rank += (uj > u) || (uj == u && list_i[j] < i);
    }
    sel[rank] = i;
  }
  __syncthreads();
  return m;
}

// ===========================================================================
// CTC prefix beam search
struct CtcBeamArgs {
  int B, T, C, W, merge, NN;
  const float *logits;
  const int32_t *len;
  int32_t *out_ids, *out_len;
  float *out_lp;
  int32_t *nodes;   // per utterance: parent[NN], label[NN], slot[NN], children[NN*(C-1)]
  size_t per_utt;   // int32 words per utterance
};

__global__ __launch_bounds__(DT) void ctc_beam_kernel(CtcBeamArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  __shared__ int s_nl, s_nnodes;
  const int b = blockIdx.x, tid = threadIdx.x, W = p.W, C = p.C, C1 = p.C - 1, blank = p.C - 1;
  float *inp = dsm;                                 // [C]   log-softmax of the frame
  float *s_tot = inp + C;                           // [W]   newp of the leaves, best first
  float *s_blk = s_tot + W, *s_lab = s_blk + W;
  float *o_tot = s_lab + W, *o_blk = o_tot + W;     // [W]   oldp
  float *n_tot = o_blk + W, *n_blk = n_tot + W, *n_lab = n_blk + W;
  int *s_node = reinterpret_cast<int *>(n_lab + W); // [W]   tree node of a beam slot
  int *s_lbl = s_node + W, *n_node = s_lbl + W, *n_lbl = n_node + W, *sel = n_lbl + W;
  float *keys = reinterpret_cast<float *>(sel + W); // [W*C] selection keys: leaves, then expansions
  int *ties = reinterpret_cast<int *>(keys + W * C);   // [W*C] candidates equal to the W-th key
  unsigned *list_u = reinterpret_cast<unsigned *>(ties + W * C);   // [W] survivors: key image, index
  int *list_i = reinterpret_cast<int *>(list_u + W);
  int *hist = list_i + W;                              // [256]
  __shared__ int s_sc[8];
  int32_t *parent = p.nodes + (size_t)b * p.per_utt, *label = parent + p.NN, *slot = label + p.NN,
          *child = slot + p.NN;
  int Tb = p.len[b];
  Tb = Tb < 0 ? 0 : (Tb > p.T ? p.T : Tb);
  if (tid == 0) {
    // the root: P(empty prefix) = 1, all of it "ending in blank"
    parent[0] = -1; label[0] = -1; slot[0] = 0;
    s_node[0] = 0; s_lbl[0] = -1;
    s_tot[0] = 0.f; s_blk[0] = 0.f; s_lab[0] = -INFINITY;
    s_nl = 1; s_nnodes = 1;
  }
  __syncthreads();
  const float *lg = p.logits + (size_t)b * p.T * C;
  for (int t = 0; t < Tb; ++t) {
    const int nl = s_nl;
    if (nl == 0) break;
    if (tid < 64) {
      const float *x = lg + (size_t)t * C;
      float m = -INFINITY;
      for (int c = tid; c < C; c += 64) m = fmaxf(m, x[c]);
      m = wave_max(m);
      float s = 0.f;
      for (int c = tid; c < C; c += 64) s += expf(x[c] - m);
      const float lz = m + logf(wave_sum(s));
      for (int c = tid; c < C; c += 64) inp[c] = x[c] - lz;
    }
    __syncthreads();
    // (1) the leaves: P(prefix @t) from P(prefix @t-1) and from the parent prefix
    float ot = 0.f, ob = 0.f, nt = 0.f, nb = 0.f, nlb = 0.f;
    if (tid < nl) {
      ot = s_tot[tid]; ob = s_blk[tid];
      nlb = s_lab[tid];
      const int node = s_node[tid], par = parent[node], lab = s_lbl[tid];
      if (par >= 0) {
        const int ps = slot[par];
        if (ps >= 0) nlb = lse2d(nlb, lab == s_lbl[ps] ? s_blk[ps] : s_tot[ps]);   // parent still in the beam
        nlb += inp[lab];
      }
      nb = ot + inp[blank];
      nt = lse2d(nb, nlb);
    }
    __syncthreads();
    if (tid < nl) {
      o_tot[tid] = ot; o_blk[tid] = ob;
      s_tot[tid] = nt; s_blk[tid] = nb; s_lab[tid] = nlb;
      keys[tid] = nt;
    }
    __syncthreads();
    // (2) expansions by one label of every leaf whose child is not itself a leaf
    const int ncand = nl * C1, n = nl + ncand;
    for (int idx = tid; idx < ncand; idx += DT) {
      const int i = idx / C1, c = idx - i * C1;
      const int ch = child[(size_t)s_node[i] * C1 + c];
      float sc = -INFINITY;
      if (!(ch >= 0 && slot[ch] >= 0) && o_tot[i] > -INFINITY) sc = inp[c] + (c == s_lbl[i] ? o_blk[i] : o_tot[i]);
      keys[nl + idx] = sc;
    }
    __syncthreads();
    // (3) the beam_width best: radix select of the W-th largest key (4 passes over 8-bit digits of
    // the order-preserving integer image of the floats, 256-bin histograms in LDS), compaction of
    // everything above it plus the lowest-index ties, then a rank sort of the <= W survivors
    const int nsel = select_best(keys, n, W, hist, list_u, list_i, ties, sel, s_sc);
    // (4) the new beam, best first; prefixes entering the tree get a node
    if (tid < nl) slot[s_node[tid]] = -1;
    __syncthreads();
    if (tid < nsel) {
      const int g = sel[tid];
      if (g < nl) {
        n_node[tid] = s_node[g]; n_lbl[tid] = s_lbl[g];
        n_tot[tid] = s_tot[g]; n_blk[tid] = s_blk[g]; n_lab[tid] = s_lab[g];
      } else {
        const int idx = g - nl, i = idx / C1, c = idx - i * C1;
        int32_t *cp = child + (size_t)s_node[i] * C1 + c;
        int ch = *cp;
        if (ch < 0) {
          ch = atomicAdd(&s_nnodes, 1);
          *cp = ch;
          parent[ch] = s_node[i];
          label[ch] = c;
        }
        const float sc = inp[c] + (c == s_lbl[i] ? o_blk[i] : o_tot[i]);
        n_node[tid] = ch; n_lbl[tid] = c;
        n_tot[tid] = sc; n_lab[tid] = sc; n_blk[tid] = -INFINITY;
      }
    }
    __syncthreads();
    if (tid < nsel) {
      s_node[tid] = n_node[tid]; s_lbl[tid] = n_lbl[tid];
      s_tot[tid] = n_tot[tid]; s_blk[tid] = n_blk[tid]; s_lab[tid] = n_lab[tid];
      slot[n_node[tid]] = tid;
    }
    if (tid == 0) s_nl = nsel;
    __syncthreads();
  }
  // the labelling of the best leaf (BeamEntry::LabelSeq): walk to the root
  __shared__ int s_len;
  if (tid == 0) {
    int n = 0;
    if (s_nl > 0) {
      int prev = -1;
      for (int x = s_node[0]; x > 0; x = parent[x]) {
        const int l = label[x];
        if (!p.merge || l != prev) ++n;
        prev = l;
      }
      int k = n;
      prev = -1;
      for (int x = s_node[0]; x > 0; x = parent[x]) {
        const int l = label[x];
        if (!p.merge || l != prev) p.out_ids[(size_t)b * p.T + --k] = l;
        prev = l;
      }
    }
    p.out_len[b] = n;
    if (p.out_lp) p.out_lp[b] = s_nl > 0 ? s_tot[0] : -INFINITY;
    s_len = n;
  }
  __syncthreads();
  for (int k = s_len + tid; k < p.T; k += DT) p.out_ids[(size_t)b * p.T + k] = -1;
}

static size_t ctc_beam_lds(int C, int W) { return ((size_t)C + 15 * (size_t)W + 2 * (size_t)W * C + 256) * 4; }

// ===========================================================================
// Levenshtein distance, one workgroup per pair, anti-diagonal wavefront in LDS
__global__ __launch_bounds__(DT) void edit_distance_kernel(int B, const int32_t *__restrict__ hyp, int ldh,
                                                           const int32_t *__restrict__ hyp_len,
                                                           const int32_t *__restrict__ truth, int ldt,
                                                           const int32_t *__restrict__ truth_len,
                                                           int32_t *__restrict__ dist) {
  extern __shared__ int ism[];
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = hyp_len[b], m = truth_len[b];
  n = n < 0 ? 0 : (n > ldh ? ldh : n);
  m = m < 0 ? 0 : (m > ldt ? ldt : m);
  int *d2 = ism, *d1 = d2 + ldh + 1, *d0 = d1 + ldh + 1;
  const int32_t *h = hyp + (size_t)b * ldh, *r = truth + (size_t)b * ldt;
  // diagonal k holds D[i][k-i] at index i
  for (int k = 0; k <= n + m; ++k) {
    const int lo = k - m > 0 ? k - m : 0, hi = k < n ? k : n;
    for (int i = lo + tid; i <= hi; i += DT) {
      const int j = k - i;
      int v;
      if (i == 0) v = j;
      else if (j == 0) v = i;
      else {
        v = min(d1[i - 1], d1[i]) + 1;
        v = min(v, d2[i - 1] + (h[i - 1] != r[j - 1] ? 1 : 0));
      }
      d0[i] = v;
    }
    __syncthreads();
    int *tmp = d2; d2 = d1; d1 = d0; d0 = tmp;
  }
  if (tid == 0) dist[b] = d1[n];
}

// ===========================================================================
// attention beam search: pruning, state gather, backwards search
struct PruneArgs {
  int B, W, C;
  const float *logits;
  float inv_temp, lpw;
  float *logprobs;
  int32_t *lengths, *finished, *seen, *pred, *parent, *stay, *all_seen;
  float *scratch;
};

__device__ __forceinline__ float length_penalty(int len, float w, float pen6) {
  return w == 0.f ? 1.f : powf(5.f + (float)len, w) / pen6;
}

__global__ __launch_bounds__(DT) void beam_prune_kernel(PruneArgs p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  __shared__ Best red[2 * (DT / 64)];
  __shared__ int s_all;
  const int b = blockIdx.x, tid = threadIdx.x, W = p.W, C = p.C, end = p.C - 1;
  const int n = W * C + W;
  float *lz = dsm, *plp = lz + W;
  int *plen = reinterpret_cast<int *>(plp + W), *pfin = plen + W, *sel = pfin + W;
  float *sc = p.scratch + (size_t)b * n;
  const float *lg = p.logits + (size_t)b * W * C;
  for (int w = tid >> 6; w < W; w += DT / 64) {
    const float *x = lg + (size_t)w * C;
    const int l = tid & 63;
    float m = -INFINITY;
    for (int c = l; c < C; c += 64) m = fmaxf(m, x[c] * p.inv_temp);
    m = wave_max(m);
    float s = 0.f;
    for (int c = l; c < C; c += 64) s += expf(x[c] * p.inv_temp - m);
    s = wave_sum(s);
    if (l == 0) lz[w] = m + logf(s);
  }
  for (int w = tid; w < W; w += DT) {
    plp[w] = p.logprobs[(size_t)b * W + w];
    plen[w] = p.lengths[(size_t)b * W + w];
    pfin[w] = p.finished[(size_t)b * W + w];
  }
  if (tid == 0) s_all = 1;
  __syncthreads();
  const float pen6 = powf(6.f, p.lpw);
  auto candidate = [&](int idx, float &lp, int &len, int &id) {
    if (idx < W * C) {
      const int w = idx / C, c = idx - w * C;
      const float nl = pfin[w] ? -FLT_MAX : lg[idx] * p.inv_temp - lz[w];
      lp = plp[w] + nl;
      len = plen[w] + (c != end ? 1 : 0);
      id = c;
    } else {
      const int j = idx - W * C;
      lp = pfin[j] ? plp[j] : -FLT_MAX;
      len = plen[j];
      id = end;
    }
  };
  for (int idx = tid; idx < n; idx += DT) {
    float lp; int len, id;
    candidate(idx, lp, len, id);
    sc[idx] = lp / length_penalty(len, p.lpw, pen6);
  }
  __syncthreads();
  Best mine = {-INFINITY, INT_MAX};
  for (int idx = tid; idx < n; idx += DT) mine = better(mine, Best{sc[idx], idx});
  for (int k = 0; k < W; ++k) {
    Best g = block_best(mine, red, k & 1);
    if (g.i == INT_MAX) g.i = k;                         // only NaNs left (NaN logits): any slot
    if (tid == 0) sel[k] = g.i;
    if ((g.i % DT) == tid) {
      sc[g.i] = NAN;                                      // taken; -inf scores stay selectable (tf.nn.top_k)
      mine = Best{-INFINITY, INT_MAX};
      for (int idx = tid; idx < n; idx += DT) mine = better(mine, Best{sc[idx], idx});
    }
  }
  __syncthreads();
  for (int k = tid; k < W; k += DT) {
    const int g = sel[k];
    float lp; int len, id;
    candidate(g, lp, len, id);
    const size_t o = (size_t)b * W + k;
    const int st = g >= W * C;
    p.parent[o] = st ? g - W * C : g / C;
    p.stay[o] = st;
    p.pred[o] = id;
    p.logprobs[o] = lp;
    p.lengths[o] = len;
    const int fin = id == end;
    p.finished[o] = fin;
    const int sn = p.seen[o] | fin;
    p.seen[o] = sn;
    if (!sn) s_all = 0;
  }
  __syncthreads();
  if (tid == 0) p.all_seen[b] = s_all;
}

__global__ __launch_bounds__(DT) void beam_gather_kernel(int W, int F, const float *__restrict__ fresh,
                                                         const float *__restrict__ old,
                                                         const int32_t *__restrict__ parent,
                                                         const int32_t *__restrict__ stay,
                                                         float *__restrict__ dst) {
  const int row = blockIdx.x, b = row / W;
  const float *src = (stay[row] ? old : fresh) + ((size_t)b * W + parent[row]) * F;
  float *d = dst + (size_t)row * F;
  for (int f = blockIdx.y * DT + threadIdx.x; f < F; f += gridDim.y * DT) d[f] = src[f];
}

// dst[(b*W+w)*F + f] = src[b*F + f]   (tf.contrib.seq2seq.tile_batch), 32-bit words
__global__ __launch_bounds__(DT) void tile_rows_kernel(int W, size_t F, const uint32_t *__restrict__ src,
                                                       uint32_t *__restrict__ dst) {
  const int row = blockIdx.x, b = row / W;
  for (size_t f = (size_t)blockIdx.y * DT + threadIdx.x; f < F; f += (size_t)gridDim.y * DT)
    dst[(size_t)row * F + f] = src[(size_t)b * F + f];
}

__global__ __launch_bounds__(DT) void fill_i32_kernel(size_t n, int32_t v, int32_t *x) {
  const size_t i = (size_t)blockIdx.x * DT + threadIdx.x;
  if (i < n) x[i] = v;
}

// logprobs = [0, -inf, ...] per utterance (beam_search_decoder.py:155-157)
__global__ __launch_bounds__(DT) void beam_init_kernel(int N, int W, float *logprobs) {
  const int i = blockIdx.x * DT + threadIdx.x;
  if (i < N) logprobs[i] = (i % W) == 0 ? 0.f : -INFINITY;
}

// finalize: follow the parent pointers from the last step (one thread per final beam slot)
__global__ __launch_bounds__(DT) void beam_backtrace_kernel(int N, int W, int Tn, int Tmax,
                                                            const int32_t *__restrict__ hist_pred,
                                                            const int32_t *__restrict__ hist_parent,
                                                            int32_t *__restrict__ seq, int32_t *__restrict__ src) {
  const int row = blockIdx.x * DT + threadIdx.x;
  if (row >= N) return;
  const int b = row / W;
  int beam = row - b * W;
  for (int t = Tn - 1; t >= 0; --t) {
    const size_t o = (size_t)t * N + (size_t)b * W + beam;
    seq[(size_t)row * Tmax + t] = hist_pred[o];
    src[(size_t)row * Tmax + t] = beam;
    beam = hist_parent[o];
  }
  for (int t = Tn; t < Tmax; ++t) seq[(size_t)row * Tmax + t] = 0;
}

__global__ __launch_bounds__(DT) void beam_align_kernel(int N, int W, int Tn, int Tmax, int Te,
                                                        const float *__restrict__ hist_align,
                                                        const int32_t *__restrict__ src,
                                                        float *__restrict__ out) {
  const int row = blockIdx.x, t = blockIdx.y, b = row / W;
  float *o = out + ((size_t)row * Tmax + t) * Te;
  if (t >= Tn) {
    for (int e = threadIdx.x; e < Te; e += DT) o[e] = 0.f;
    return;
  }
  const float *a = hist_align + ((size_t)t * N + (size_t)b * W + src[(size_t)row * Tmax + t]) * Te;
  for (int e = threadIdx.x; e < Te; e += DT) o[e] = a[e];
}

__global__ __launch_bounds__(DT) void beam_scores_kernel(int N, float lpw, const float *__restrict__ logprobs,
                                                         const int32_t *__restrict__ lengths,
                                                         float *__restrict__ scores, int32_t *__restrict__ out_len) {
  const int i = blockIdx.x * DT + threadIdx.x;
  if (i >= N) return;
  scores[i] = logprobs[i] / length_penalty(lengths[i], lpw, powf(6.f, lpw));
  out_len[i] = lengths[i];
}

static int set_lds(const void *fn, size_t bytes, const char *what) {
  if (bytes > 160 * 1024) return fail(NABU_EUNSUP, "%s: needs %zu bytes of LDS (160 KiB per workgroup)", what, bytes);
  if (bytes > 48 * 1024) NABU_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return 0;
}

// workspace of the attention beam search, offsets in 32-bit words
struct BeamWs {
  size_t valuesT, keysB, keysT, lenT, big, z, q, logits, acts, ids, logprobs, lengths, finished, seen, parent,
      stay, all_seen, scratch, hist_pred, hist_parent, hist_align, src, gemm, gemm_bytes, total;
  size_t h[3][NABU_SPELLER_MAX_LAYERS], c[3][NABU_SPELLER_MAX_LAYERS], ctx[3], align[3];
};

static BeamWs beam_ws(const nabu_beam_desc *d) {
  BeamWs s;
  const size_t B = d->B, W = d->beam_width, N = B * W, U = d->U, E = d->E, Te = d->Te, C = d->C, S = d->max_steps;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  s.valuesT = take(N * Te * E);
  s.keysB = take(B * Te * U);
  s.keysT = take(N * Te * U);
  s.lenT = take(N); s.big = take(N);
  s.z = take(N * 4 * U); s.q = take(N * U); s.logits = take(N * C); s.acts = take(N * 4 * U);
  s.ids = take(N); s.logprobs = take(N); s.lengths = take(N); s.finished = take(N); s.seen = take(N);
  s.parent = take(N); s.stay = take(N); s.all_seen = take(B);
  {
    const nabu_attn_desc ad = {sizeof(nabu_attn_desc), (int32_t)N, d->Te, d->E, d->U, d->kind, d->K, d->F, d->prob_fn};
    const size_t need = nabu_attn_fwd_ws_bytes(&ad) / 4, prune = B * (W * C + W);
    s.scratch = take(need > prune ? need : prune);
  }
  s.hist_pred = take(S * N); s.hist_parent = take(S * N);
  s.hist_align = take(S * N * Te);
  s.src = take(N * S);
  for (int k = 0; k < 3; ++k) {
    for (int n = 0; n < d->num_layers; ++n) { s.h[k][n] = take(N * U); s.c[k][n] = take(N * U); }
    s.ctx[k] = take(N * E);
    s.align[k] = take(N * Te);
  }
  size_t g = 0;
  auto mx = [&](size_t v) { if (v > g) g = v; };
  mx(nabu_gemm_ws_bytes((int)N, (int)(4 * U), (int)E)); mx(nabu_gemm_ws_bytes((int)N, (int)(4 * U), (int)U));
  mx(nabu_gemm_ws_bytes((int)N, (int)U, (int)U)); mx(nabu_gemm_ws_bytes((int)N, (int)C, (int)U));
  mx(nabu_gemm_ws_bytes((int)N, (int)C, (int)E)); mx(nabu_gemm_ws_bytes((int)(B * Te), (int)U, (int)E));
  s.gemm_bytes = (g + 255) / 256 * 256;
  s.gemm = take(s.gemm_bytes / 4 + 4);
  s.total = o;
  return s;
}

static int check_beam(const nabu_beam_desc *d) {
  if (!d || d->size != sizeof(nabu_beam_desc)) return fail(NABU_EINVAL, "beam search: bad descriptor size");
  if (d->B <= 0 || d->Te <= 0 || d->E <= 0 || d->U <= 0 || d->C <= 1) return fail(NABU_EINVAL, "beam search: bad dimensions");
  if (d->beam_width <= 0 || d->max_steps <= 0) return fail(NABU_EINVAL, "beam search: beam_width and max_steps must be positive");
  if (!(d->temperature > 0.f)) return fail(NABU_EINVAL, "beam search: temperature must be positive");
  if (d->kind < 0 || d->kind > 2 || d->prob_fn < 0 || d->prob_fn > 2)
    return fail(NABU_EINVAL, "beam search: unknown attention kind or probability_fn");
  if (d->num_layers < 1 || d->num_layers > NABU_SPELLER_MAX_LAYERS) return fail(NABU_EUNSUP, "beam search: 1..%d layers", NABU_SPELLER_MAX_LAYERS);
  if (d->U % 4 || d->E % 4) return fail(NABU_EUNSUP, "beam search: num_units and encoder dim must be multiples of 4");
  if ((long long)d->B * d->beam_width > (1 << 20)) return fail(NABU_EUNSUP, "beam search: B*beam_width too large");
  return 0;
}

static int bmm(int M, int N, int K, const float *A, int lda, const float *Bm, int ldb, float beta, float *C, int ldc,
               const float *bias, float *ws, size_t wsb, nabu_stream_t st) {
  return nabu_gemm_f32(0, 0, M, N, K, 1.f, A, lda, Bm, ldb, beta, C, ldc, bias, 0, 0, 0, ws, wsb, st);
}
#define DEC_TRY(call) do { int e_ = (call); if (e_) return e_; } while (0)

}  // namespace nabu

using namespace nabu;

extern "C" size_t nabu_ctc_beam_ws_bytes(int B, int T, int C, int beam_width) {
  if (B <= 0 || T <= 0 || C <= 1 || beam_width <= 0) return 0;
  const size_t NN = 1 + (size_t)T * beam_width;
  return (size_t)B * NN * (3 + (size_t)C - 1) * 4;
}

extern "C" int nabu_ctc_beam_search(int B, int T, int C, int beam_width, int merge_repeated, const float *logits,
                                    const int32_t *logit_len, int32_t *out_ids, int32_t *out_len,
                                    float *out_logprob, void *ws, size_t ws_bytes, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && T > 0 && C > 1 && beam_width > 0, "ctc_beam_search: bad dimensions");
  NABU_CHECK_ARG(logits && logit_len && out_ids && out_len && ws, "ctc_beam_search: null pointer");
  if (beam_width > DT) return fail(NABU_EUNSUP, "ctc_beam_search: beam_width <= %d", DT);
  const size_t need = nabu_ctc_beam_ws_bytes(B, T, C, beam_width);
  if (ws_bytes < need) return fail(NABU_EWS, "ctc_beam_search: workspace too small");
  const size_t lds = ctc_beam_lds(C, beam_width);
  DEC_TRY(set_lds(reinterpret_cast<const void *>(ctc_beam_kernel), lds, "ctc_beam_search"));
  hipStream_t s = static_cast<hipStream_t>(stream);
  NABU_HIP(hipMemsetAsync(ws, 0xFF, need, s));   // every tree pointer = -1
  CtcBeamArgs a;
  a.B = B; a.T = T; a.C = C; a.W = beam_width; a.merge = merge_repeated ? 1 : 0;
  a.NN = 1 + T * beam_width;
  a.logits = logits; a.len = logit_len; a.out_ids = out_ids; a.out_len = out_len; a.out_lp = out_logprob;
  a.nodes = static_cast<int32_t *>(ws);
  a.per_utt = (size_t)a.NN * (3 + (size_t)C - 1);
  hipLaunchKernelGGL(ctc_beam_kernel, dim3(B), dim3(DT), lds, s, a);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_edit_distance(int B, const int32_t *hyp, int ldh, const int32_t *hyp_len, const int32_t *truth,
                                  int ldt, const int32_t *truth_len, int32_t *dist, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && ldh >= 0 && ldt >= 0, "edit_distance: bad dimensions");
  NABU_CHECK_ARG(hyp_len && truth_len && dist && (hyp || ldh == 0) && (truth || ldt == 0), "edit_distance: null pointer");
  const size_t lds = 3 * ((size_t)ldh + 1) * 4;
  DEC_TRY(set_lds(reinterpret_cast<const void *>(edit_distance_kernel), lds, "edit_distance"));
  hipLaunchKernelGGL(edit_distance_kernel, dim3(B), dim3(DT), lds, static_cast<hipStream_t>(stream), B, hyp, ldh,
                     hyp_len, truth, ldt, truth_len, dist);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_beam_prune(int B, int W, int C, const float *logits, float temperature, float length_penalty_w,
                               float *logprobs, int32_t *lengths, int32_t *finished, int32_t *seen,
                               int32_t *pred_ids, int32_t *parent, int32_t *stay, int32_t *all_seen,
                               float *scratch, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && W > 0 && C > 1 && temperature > 0.f, "beam_prune: bad arguments");
  NABU_CHECK_ARG(logits && logprobs && lengths && finished && seen && pred_ids && parent && stay && all_seen && scratch,
                 "beam_prune: null pointer");
  const size_t lds = 5 * (size_t)W * 4;
  DEC_TRY(set_lds(reinterpret_cast<const void *>(beam_prune_kernel), lds, "beam_prune"));
  PruneArgs a = {B, W, C, logits, 1.f / temperature, length_penalty_w, logprobs, lengths, finished, seen,
                 pred_ids, parent, stay, all_seen, scratch};
  hipLaunchKernelGGL(beam_prune_kernel, dim3(B), dim3(DT), lds, static_cast<hipStream_t>(stream), a);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" int nabu_beam_gather(int B, int W, int F, const float *fresh, const float *old, const int32_t *parent,
                                const int32_t *stay, float *dst, nabu_stream_t stream) {
  NABU_CHECK_ARG(B > 0 && W > 0 && F > 0, "beam_gather: bad dimensions");
  NABU_CHECK_ARG(fresh && old && parent && stay && dst, "beam_gather: null pointer");
  const int gy = (F + DT * 4 - 1) / (DT * 4);
  hipLaunchKernelGGL(beam_gather_kernel, dim3(B * W, gy > 64 ? 64 : gy), dim3(DT), 0,
                     static_cast<hipStream_t>(stream), W, F, fresh, old, parent, stay, dst);
  NABU_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t nabu_speller_beam_ws_bytes(const nabu_beam_desc *d) {
  if (check_beam(d)) return 0;
  return beam_ws(d).total * 4;
}

extern "C" int nabu_speller_beam_search(const nabu_beam_desc *d, const float *values, const int32_t *enc_len,
                                        const nabu_speller_params *p, int32_t *sequences, int32_t *lengths,
                                        float *scores, float *alignments, int32_t *num_steps, void *ws,
                                        size_t ws_bytes, nabu_stream_t stream) {
  if (int e = check_beam(d)) return e;
  NABU_CHECK_ARG(values && enc_len && p && sequences && lengths && scores && num_steps && ws, "speller_beam_search: null pointer");
  const BeamWs L = beam_ws(d);
  if (ws_bytes < L.total * 4) return fail(NABU_EWS, "speller_beam_search: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float *w = static_cast<float *>(ws);
  int32_t *wi = static_cast<int32_t *>(ws);
  const int B = d->B, W = d->beam_width, N = B * W, U = d->U, E = d->E, Te = d->Te, C = d->C, nl = d->num_layers,
            S = d->max_steps;
  float *gw = w + L.gemm;
  const size_t gwb = L.gemm_bytes;
  const nabu_attn_desc ad = {sizeof(nabu_attn_desc), N, Te, E, U, d->kind, d->K, d->F, d->prob_fn};
  const size_t attn_wsb = nabu_attn_fwd_ws_bytes(&ad);   // sliced forward (small B*W): partials in the pruning scratch
  auto tile = [&](const void *src, void *dst, size_t F) {
    size_t gy = (F + DT * 4 - 1) / (DT * 4);
    hipLaunchKernelGGL(tile_rows_kernel, dim3(N, gy > 64 ? 64 : (unsigned)gy), dim3(DT), 0, s, W, F,
                       static_cast<const uint32_t *>(src), static_cast<uint32_t *>(dst));
  };
  // tile_batch of the encoder output / its lengths; keys = memory_layer(values) once per utterance
  DEC_TRY(bmm(B * Te, U, E, values, E, p->memory_kernel, U, 0.f, w + L.keysB, U, nullptr, gw, gwb, stream));
  tile(values, w + L.valuesT, (size_t)Te * E);
  tile(w + L.keysB, w + L.keysT, (size_t)Te * U);
  tile(enc_len, wi + L.lenT, 1);
  NABU_LAUNCH_CHECK();
  const int gN = (N + DT - 1) / DT;
  hipLaunchKernelGGL(fill_i32_kernel, dim3(gN), dim3(DT), 0, s, (size_t)N, INT_MAX, wi + L.big);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(gN), dim3(DT), 0, s, (size_t)N, C - 1, wi + L.ids);       // start tokens
  hipLaunchKernelGGL(beam_init_kernel, dim3(gN), dim3(DT), 0, s, N, W, w + L.logprobs);
  NABU_LAUNCH_CHECK();
  NABU_HIP(hipMemsetAsync(wi + L.lengths, 0, (size_t)N * 4, s));
  NABU_HIP(hipMemsetAsync(wi + L.finished, 0, (size_t)N * 4, s));
  NABU_HIP(hipMemsetAsync(wi + L.seen, 0, (size_t)N * 4, s));
  int cur = 0, fresh = 1, nxt = 2;                 // state sets: before the step, after the cell, after pruning
  for (int n = 0; n < nl; ++n) {
    NABU_HIP(hipMemsetAsync(w + L.h[cur][n], 0, (size_t)N * U * 4, s));
    NABU_HIP(hipMemsetAsync(w + L.c[cur][n], 0, (size_t)N * U * 4, s));
  }
  NABU_HIP(hipMemsetAsync(w + L.ctx[cur], 0, (size_t)N * E * 4, s));
  NABU_HIP(hipMemsetAsync(w + L.align[cur], 0, (size_t)N * Te * 4, s));
  if (d->kind == 2) DEC_TRY(first_col_one(N, Te, w + L.align[cur], s));
  float *z = w + L.z, *lg = w + L.logits;
  const int32_t *big = wi + L.big, *lenT = wi + L.lenT;
  int32_t *par = wi + L.parent, *stay = wi + L.stay;
  std::vector<int32_t> done(B);
  int Tn = 0;
  for (int t = 0; t < S; ++t) {
    // the cell on all B*W rows (the step of nabu_speller_fwd; no dropout at inference)
    for (int n = 0; n < nl; ++n) {
      const float *Kn = p->lstm_kernel[n];
      if (n == 0) {
        DEC_TRY(bmm(N, 4 * U, E, w + L.ctx[cur], E, Kn + (size_t)C * 4 * U, 4 * U, 0.f, z, 4 * U, nullptr, gw, gwb, stream));
        DEC_TRY(bmm(N, 4 * U, U, w + L.h[cur][0], U, Kn + (size_t)(C + E) * 4 * U, 4 * U, 1.f, z, 4 * U, nullptr, gw, gwb, stream));
        DEC_TRY(nabu_lstm_cell_fwd(N, U, 0, big, z, p->lstm_bias[0], Kn, wi + L.ids, w + L.c[cur][0], w + L.h[cur][0],
                                   w + L.acts, w + L.c[fresh][0], w + L.h[fresh][0], stream));
      } else {
        DEC_TRY(bmm(N, 4 * U, U, w + L.h[fresh][n - 1], U, Kn, 4 * U, 0.f, z, 4 * U, nullptr, gw, gwb, stream));
        DEC_TRY(bmm(N, 4 * U, U, w + L.h[cur][n], U, Kn + (size_t)U * 4 * U, 4 * U, 1.f, z, 4 * U, nullptr, gw, gwb, stream));
        DEC_TRY(nabu_lstm_cell_fwd(N, U, 0, big, z, p->lstm_bias[n], nullptr, nullptr, w + L.c[cur][n], w + L.h[cur][n],
                                   w + L.acts, w + L.c[fresh][n], w + L.h[fresh][n], stream));
      }
    }
    const float *htop = w + L.h[fresh][nl - 1];
    DEC_TRY(bmm(N, U, U, htop, U, p->query_kernel, U, 0.f, w + L.q, U, nullptr, gw, gwb, stream));
    DEC_TRY(nabu_attn_fwd(&ad, 0, big, lenT, w + L.keysT, w + L.valuesT, w + L.q, p->attention_v, p->conv_kernel,
                          p->conv_proj, w + L.align[cur], w + L.ctx[cur], w + L.align[fresh], w + L.ctx[fresh],
                          w + L.acts /* normaliser scratch: the saved gate activations are not used at inference */,
                          w + L.scratch, attn_wsb, stream));
    // AttentionProjectionWrapper: [h, context of this step]·W + b (rnn_cell.py:145-155)
    DEC_TRY(bmm(N, C, U, htop, U, p->out_kernel, C, 0.f, lg, C, p->out_bias, gw, gwb, stream));
    DEC_TRY(bmm(N, C, E, w + L.ctx[fresh], E, p->out_kernel + (size_t)U * C, C, 1.f, lg, C, nullptr, gw, gwb, stream));
    // expand + prune; the predicted ids are the next step's inputs
    DEC_TRY(nabu_beam_prune(B, W, C, lg, d->temperature, d->length_penalty, w + L.logprobs, wi + L.lengths,
                            wi + L.finished, wi + L.seen, wi + L.ids, par, stay, wi + L.all_seen, w + L.scratch, stream));
    for (int n = 0; n < nl; ++n) {
      DEC_TRY(nabu_beam_gather(B, W, U, w + L.h[fresh][n], w + L.h[cur][n], par, stay, w + L.h[nxt][n], stream));
      DEC_TRY(nabu_beam_gather(B, W, U, w + L.c[fresh][n], w + L.c[cur][n], par, stay, w + L.c[nxt][n], stream));
    }
    DEC_TRY(nabu_beam_gather(B, W, E, w + L.ctx[fresh], w + L.ctx[cur], par, stay, w + L.ctx[nxt], stream));
    DEC_TRY(nabu_beam_gather(B, W, Te, w + L.align[fresh], w + L.align[cur], par, stay, w + L.align[nxt], stream));
    NABU_HIP(hipMemcpyAsync(wi + L.hist_pred + (size_t)t * N, wi + L.ids, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    NABU_HIP(hipMemcpyAsync(wi + L.hist_parent + (size_t)t * N, par, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
    NABU_HIP(hipMemcpyAsync(w + L.hist_align + (size_t)t * N * Te, w + L.align[nxt], (size_t)N * Te * 4,
                            hipMemcpyDeviceToDevice, s));
    const int tmp = cur; cur = nxt; nxt = tmp;
    Tn = t + 1;
    // dynamic_decode's stop test: every slot has been finished at some step
    NABU_HIP(hipMemcpyAsync(done.data(), wi + L.all_seen, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    NABU_HIP(hipStreamSynchronize(s));
    bool all = true;
    for (int b = 0; b < B; ++b) all = all && done[b] != 0;
    if (all) break;
  }
  // finalize (beam_search_decoder.py:341-451)
  hipLaunchKernelGGL(beam_backtrace_kernel, dim3(gN), dim3(DT), 0, s, N, W, Tn, S, wi + L.hist_pred, wi + L.hist_parent,
                     sequences, wi + L.src);
  NABU_LAUNCH_CHECK();
  if (alignments) {
    hipLaunchKernelGGL(beam_align_kernel, dim3(N, S), dim3(DT), 0, s, N, W, Tn, S, Te, w + L.hist_align, wi + L.src, alignments);
    NABU_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(beam_scores_kernel, dim3(gN), dim3(DT), 0, s, N, d->length_penalty, w + L.logprobs, wi + L.lengths,
                     scores, lengths);
  NABU_LAUNCH_CHECK();
  *num_steps = Tn;
  return 0;
}
