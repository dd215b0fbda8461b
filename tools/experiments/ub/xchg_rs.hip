// xchg_rs: the backward recurrence's exchange skeleton (reduce-scatter) without arithmetic.  32 workgroups per XCD; per
// step every wave publishes 4 x 1 KiB (its 8 destination tiles, 16-byte pieces), polls the 4 KiB addressed to its rows
// until no word holds the sentinel, and hands the pieces back (sentinel stores).  ring of 3 slots.
// build: hipcc --offload-arch=gfx950 -O3 xchg_rs.hip -o xchg_rs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(char *buf, int steps, int delay_ticks, int do_reset, long long *out) {
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int P = 32;
  constexpr unsigned PIECE = 512, BLOCK = P * PIECE, SLOT = P * BLOCK;   // 512 KiB per ring slot
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf + (size_t)xcd * (3 * SLOT), 0, 3 * SLOT, 0x00020000);
  const int n = lane & 15, q = lane >> 4;
  const int s8 = lane & 7, kq = (lane >> 3) & 3, r2 = lane >> 5, row = 2 * w + r2;
  const unsigned in_off = slot * BLOCK + (s8 * 8 + row) * 64 + kq * 16;
  long long t0 = wall_clock64();
  unsigned sink = 0;
  for (int s = 1; s <= steps; ++s) {
    const unsigned sb = (s % 3) * SLOT;
    const u32x4 pv = {(unsigned)s, (unsigned)s, (unsigned)s, (unsigned)s};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int dest = 8 * w + t + (n < 8 ? 0 : 4);
      __builtin_amdgcn_raw_buffer_store_b128(pv, rs, sb + dest * BLOCK + slot * PIECE + (n & 7) * 64 + q * 16, 0, 0);
    }
    u32x4 v[4];
    for (;;) {
      unsigned mx = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, sb + in_off + i * 4096u, 0, 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = max(mx, max(max(v[i].x, v[i].y), max(v[i].z, v[i].w)));
      if (__all(mx != 0xFFFFFFFFu)) { sink += v[0].x; break; }
    }
    if (do_reset) {
      const u32x4 sent = {~0u, ~0u, ~0u, ~0u};
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_buffer_store_b128(sent, rs, sb + in_off + i * 4096u, 0, 0);
    }
    if (delay_ticks) {
      const long long d0 = wall_clock64();
      while (wall_clock64() - d0 < delay_ticks) __builtin_amdgcn_s_sleep(1);
    }
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (sink == 0x12345 ? 1 : 0);
}

int main() {
  char *buf; long long *out;
  const size_t bytes = (size_t)8 * 3 * 32 * 32 * 512;
  (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 256 * 8);
  const int steps = 1500;   // (without resets a slot still holds step s - 3: the poll then passes at once — lower bound)
  for (int do_reset = 1; do_reset >= 0; --do_reset)
  for (int delay : {0, 60, 120}) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(buf, 0xFF, bytes);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, buf, steps, delay, do_reset, out);
      (void)hipDeviceSynchronize();
    }
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    printf("reset %d delay %4d ns: %.0f ns per step (hand-off = %.0f)\n", do_reset, delay * 10, mx * 10.0 / steps, mx * 10.0 / steps - delay * 10);
  }
  return 0;
}
