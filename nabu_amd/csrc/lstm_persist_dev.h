// lstm_persist_dev.h — device-side helpers shared by the persistent recurrent kernels
// (lstm_persist.hip: exact-fp32 4x4x1 product, 4 / 8 rows per unit; lstm_persist_mxh.hip / _mxf.hip: fp16-plane product on
// v_mfma_f32_16x16x32_bf16, 8 rows per unit).  Protocol description: header of lstm_persist.hip.
#pragma once
#include "lstm_persist.h"

#include <stdlib.h>

namespace nabu {

constexpr unsigned SENT = 0xFFFFFFFFu;
constexpr unsigned OOB = 0xFFFFFFF0u;   // buffer offset beyond every exchange ring / tensor: access dropped
constexpr int UC = 16;    // hidden units per workgroup
#ifndef NABU_RING_BWD
#define NABU_RING_BWD 2
#endif
#ifndef NABU_FWD_NACC
#define NABU_FWD_NACC 2   // 4 chains measured: product phase 720 -> 680 ns, step time unchanged (2.35 -> 2.40 us)
#endif
constexpr int RING = 4;               // exchange ring depth, forward (all-gather of h)
constexpr int RINGB = NABU_RING_BWD;  // ... backward (reduce-scatter of dh)
constexpr int NCU = 256;  // MI355X
constexpr size_t TABLE_BYTES = 4096;   // XCC-id table in front of the ring

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PersistArgs {
  int B, T, D, H, max_len, nshard;
  const int32_t *len;
  const float *kernel[2];
  float *gates[2];
  float *cs[2];
  float *out;         // forward
  const float *x;     // forward, narrow input (XK > 0): the layer input [B, T, D], projected inside the kernel
  const void *xplanes;    // ... fp16-plane kernels (XIN): x as two fp16 planes [B][T][2][64 k], scaled by 2^14 / xscale[0]
  const float *xscale;    // ... xscale[0] = g, a power of two with |x| <= g (lstm_mxh_prepare_x)
  const float *bias[2];   // ... and the cells' biases [4H]
  const float *dout;  // backward
  float *db_part;     // backward: [shards][2 directions][4H] bias-gradient partial sums (one row per unit)
  float *amax_part;   // backward: same shape, the largest |dz| of every gate column over the unit's rows and steps
  unsigned *rowmax_part;   // backward (fp16-plane kernels; may be null): [2 directions x P workgroups][rowmax_stride] bit
  unsigned rowmax_stride;  // patterns of every frame row's largest |dz| over the workgroup's 64 gate columns
  int shard_base;     // first shard of this launch (batches split over several launches)
  unsigned *table;    // [grid] XCC ids, pre-set to SENT
  char *xbuf;         // exchange ring
  int *status;
  unsigned long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
  int dbg;  // NABU_PERSIST_DEBUG: 1 no exchange wait, 2 no matrix product, 4 phase stamps,
            // 8 force write-through publishing, 16 force BS = 8 (timing experiments only)
  EmitArgs emit;      // forward, fp16-plane kernels <.., EMIT = true>: packed companions of `out` (lstm_persist.h)
};

__device__ __forceinline__ float dpp_f(float v, const int ctrl_sel) {
  // quad permutes only (well defined on every wave64 target)
  int r;
  const int x = __builtin_bit_cast(int, v);
  switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;   // [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;   // [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, true); break;   // bcast lane 0
    case 3: r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, true); break;   // bcast lane 1
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, true); break;   // bcast lane 2
    default: r = __builtin_amdgcn_update_dpp(0, x, 0xFF, 0xF, 0xF, true); break;  // bcast lane 3
  }
  return __builtin_bit_cast(float, r);
}
#define QUAD_XOR1(v) dpp_f(v, 0)
#define QUAD_XOR2(v) dpp_f(v, 1)
#define QUAD_BCAST(v, i) dpp_f(v, 2 + (i))

// Gate functions.  v_rcp_f32 is good to one unit in the last place but not centred (mean -0.06 .. -0.08 ulp on [1, 2),
// tools/experiments/ub/rcp_bias.hip), and 2 r - 1 turns that into a RELATIVE bias of tanh that grows as tanh shrinks: one
// cell update came out 3e-8 low in h on average (tools/experiments/ub/cell_bias.hip; the step kernels' true divisions:
// 4e-9) — invisible in any single value, but a sum over 32 000 frames (a bias gradient) or a thousand dependent steps
// collects it: at T = 1000 the persistent kernels' gradients stood at 1.4-5 x the step kernels' error against float64.
// So: one Newton step behind the reciprocal (two fused multiply-adds: centred again, mean as the division's), and tanh
// as (1 - e) / (1 + e), whose numerator cancels nothing behind the reciprocal (half the rms error of 2 r - 1 besides);
// the clamp keeps e = exp(-2x) finite (|tanh| = 1 to the last bit beyond |x| = 9 already).
__device__ __forceinline__ float rcp_newton(float d) {
  const float r = __builtin_amdgcn_rcpf(d);
  return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return rcp_newton(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * __builtin_amdgcn_fmed3f(x, -30.0f, 30.0f));
  return (1.0f - e) * rcp_newton(1.0f + e);
}

__device__ __forceinline__ bool has_sentinel(const u32x4 v) {
  return v.x == SENT || v.y == SENT || v.z == SENT || v.w == SENT;
}
__device__ __forceinline__ bool all_sentinel(const u32x4 v) {
  return v.x == SENT && v.y == SENT && v.z == SENT && v.w == SENT;
}
__device__ __forceinline__ float sel4(int q, float a, float b, float c, float d) {
  return q == 0 ? a : q == 1 ? b : q == 2 ? c : d;
}

// PER-STEP PREFETCH.  hipcc's wait-count insertion drains the WHOLE vector memory queue
// (s_waitcnt vmcnt(0)) at most control-flow joins; a compiler-visible prefetch of the next step's
// saved tensors would put its HBM latency, or the acknowledgement of the exchange stores, on the
// critical path.  The prefetch is therefore an LDS-DMA load (buffer_load ... lds: no destination
// register, so no stale register copies are possible) issued from inline assembly, invisible to
// the compiler, and claimed with an explicit counted wait before an ordinary LDS read:
// wait_vm<N>, N = vector memory instructions certainly issued after the prefetch.  Vector memory
// operations complete in issue order, so waits the compiler inserts for its own loads can only
// become stricter by the extra operation, never weaker.  Lane l of wave w lands at
// stage[64 w + l]; out-of-range offsets deliver 0.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 raw_rsrc(const void *base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return (i32x4){(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void prefetch_lds_b32(i32x4 rsrc, unsigned off, const float *smem, const float *stage_wave) {
  const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)((const char *)stage_wave - (const char *)smem));
  asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m0), "v"(off), "s"(rsrc) : "memory");
}
// the same with 16 bytes per lane (lane l of the wave lands at stage_wave + 16 l bytes: 1 KiB per wave)
__device__ __forceinline__ void prefetch_lds_b128(i32x4 rsrc, unsigned off, const float *smem, const float *stage_wave) {
  const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)((const char *)stage_wave - (const char *)smem));
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0), "v"(off), "s"(rsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// publish 16 bytes: plain store when the whole unit shares one L2, else write-through
__device__ __forceinline__ void xstore(const u32x4 v, __amdgpu_buffer_rsrc_t rs, unsigned off, bool coloc) {
  if (coloc) __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 0);
  else       __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, 16);
}

// Bounded spin bookkeeping: returns true when the caller must give up.
struct SpinGuard {
  unsigned long long t0;
  unsigned spins;
  __device__ __forceinline__ void start() { t0 = wall_clock64(); spins = 0; }
  __device__ __forceinline__ bool expired(const PersistArgs &p) {
    // back off between polls: 512 workgroups re-reading 8 KiB each as fast as the L2 answers
    // (one way latency is ~50 ns, tools/experiments/ub/pingpong.hip) would saturate the L2 they wait on
    switch ((p.dbg >> 8) & 7) {
      case 1: __builtin_amdgcn_s_sleep(1); break;
      case 2: __builtin_amdgcn_s_sleep(2); break;
      case 3: __builtin_amdgcn_s_sleep(4); break;
      case 4: __builtin_amdgcn_s_sleep(8); break;
      case 5: __builtin_amdgcn_s_sleep(16); break;
      default: break;
    }
    if ((++spins & 31u) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return true;
    return wall_clock64() - t0 > p.timeout_ticks;
  }
};

// A poll round that did not find its data (fp16-plane kernels: lstm_persist_mxh.hip, lstm_persist_mxf.hip): bounded-spin
// bookkeeping in ONE place — the clock is first read here, every 8th failed round the wave naps, looks at the status word
// (another workgroup gave up, or an earlier launch on this workspace did) and at the clock; on expiry it raises the
// workgroup's flag and the status word (code: 1 forward, 2 backward, + 4 x block) and tells the caller to leave its loop.
__device__ __forceinline__ bool poll_round_failed(const PersistArgs &p, int *flag, int lane, int &fails,
                                                  unsigned long long &t_fail, int code) {
  if (fails == 0) t_fail = wall_clock64();
  if ((++fails & 7) != 0) return false;
  __builtin_amdgcn_s_sleep(1);
  if (__hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t_fail <= p.timeout_ticks)
    return false;
  if (lane == 0) {
    flag[0] = 1;
    __hip_atomic_store(p.status, code + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}

// NABU_PERSIST_DEBUG & 4: block 0 / thread 0 records the wall clock (10 ns units) at phase
// boundaries of the middle timestep into status[320 + 32*pass + i] (pass 0 fwd, 1 bwd).
#define NABU_STAMP(pass, i)                                                       \
  do {                                                                            \
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0 && s == p.max_len / 2)         \
      p.status[320 + 32 * (pass) + (i)] = (int)(wall_clock64());                  \
  } while (0)

// CLOCK STAMPS (always on, fp16-plane kernels): block 0 / thread 0 leaves the shader-clock counter (s_memtime: one tick per
// shader cycle) and the constant 100 MHz wall clock at the start and at the end of the launch in
// status[280 + 4 pass + {0: clock, 1: wall at start; 2: clock, 3: wall at end}] (pass 0 forward, 1 backward; low 32 bits:
// differences are exact for launches shorter than 1.8 s).  (clock_end - clock_start) / (wall_end - wall_start) x 100 MHz
// is the clock the chip actually sustained under THIS kernel — bench.py prints it (`effective_clock`), so that a slow
// box of the pool is visible as such.  Two stores per launch, outside the step loop and outside every counted wait.
constexpr int CLOCK_STAMP_BASE = 280;
__device__ __forceinline__ void clock_stamp(const PersistArgs &p, int pass, int end) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    p.status[CLOCK_STAMP_BASE + 4 * pass + 2 * end] = (int)__builtin_readcyclecounter();
    p.status[CLOCK_STAMP_BASE + 4 * pass + 2 * end + 1] = (int)wall_clock64();
  }
}

// Kernel start: publish my XCC id, wait for the ids of my unit, decide whether the unit
// is co-located on one XCD.  Returns false on timeout.  flag[0] = failure, flag[1] = coloc.
// Logical identity of a block.  Blocks b and b+256 share a CU (measured: the dispatcher fills
// every CU once before it places a second workgroup), so the second wave of blocks is rotated by
// NU/2 units: the two workgroups of a CU then belong to DIFFERENT units of the same XCD and can
// interleave (same-unit workgroups are in lockstep and would always collide on the VALU).
__device__ __forceinline__ void block_identity(int NU, int *unit, int *slot, bool rotate) {
  const int b = blockIdx.x;
  int u = b % NU;
  if (rotate && b >= NCU && NCU % NU == 0) u = (u + NU / 2) % NU;
  *unit = u;
  *slot = b / NU;
}

__device__ __forceinline__ bool unit_handshake(const PersistArgs &p, int unit, int slot, int NU, int P, int *flag) {
  const int tid = threadIdx.x;
  const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));  // HW_REG_XCC_ID
  if (tid == 0) {
    flag[0] = __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag[1] = 0;
    if (blockIdx.x < 256) p.status[16 + blockIdx.x] = (int)xcc;   // diagnostic
    __hip_atomic_store(p.table + unit + NU * slot, xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (flag[0]) return false;   // an earlier kernel of this workspace timed out
  if (tid < 64) {
    SpinGuard guard;
    guard.start();
    unsigned v = xcc;
    bool failed = false;
    for (;;) {
      if (tid < P) v = __hip_atomic_load(p.table + unit + NU * tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(v != SENT)) break;
      if (guard.expired(p)) { failed = true; break; }
    }
    const bool same = __all(v == xcc) && !(p.dbg & 8);
    if (tid == 0) {
      if (failed) {
        flag[0] = 1;
        __hip_atomic_store(p.status, 3 + 4 * (int)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      flag[1] = same ? 1 : 0;
    }
  }
  __syncthreads();
  return flag[0] == 0;
}

}  // namespace nabu
