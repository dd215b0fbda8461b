// xchg: the forward recurrence's exchange skeleton without any arithmetic.  32 workgroups per XCD (4 waves each) form a
// unit; per step every wave publishes 12 cells of 16 bytes (value = step) and then polls the 6 KiB of the unit's slot
// that "its k range" covers until every word carries the step.  Reports ns per step by store flavour and by an
// artificial delay between "poll complete" and "publish" (the product + gate phase of the real kernel).
// build: hipcc --offload-arch=gfx950 -O3 xchg.hip -o xchg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void k(char *buf, int steps, int delay_ticks, long long *out) {
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr unsigned SLOT = 24 * 512 * 2;   // 24 KiB per ring slot
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf + xcd * (4 * SLOT), 0, 4 * SLOT, 0x00020000);
  // my 12 cells: k groups 2 slot + g (g = lane >> 3 & 1), cell plane * 8 + row, row = 2 w + r2
  const int u16 = lane & 15, r2 = (lane >> 4) & 1, ppl = u16 & 7;
  const bool pub = lane < 32 && ppl < 3;
  const unsigned pub_off = (unsigned)(((2 * slot + (u16 >> 3)) * 24 + ppl * 8 + 2 * w + r2) * 16);
  const int n = lane & 15, q = lane >> 4;
  const unsigned off1 = (unsigned)(((w * 16 + q) * 24 + n) * 16);
  const unsigned off2 = (unsigned)(((w * 16 + q) * 24 + 16 + (n & 7)) * 16) + (unsigned)(n >> 3) * 1536u;
  long long t0 = wall_clock64();
  unsigned sink = 0;
  for (int s = 1; s <= steps; ++s) {
    const u32x4 pv = {(unsigned)s, (unsigned)s, (unsigned)s, (unsigned)s};
    __builtin_amdgcn_raw_buffer_store_b128(pv, rs, pub ? (s & 3) * SLOT + pub_off : 0xFFFFFFF0u, 0, AUX);
    for (;;) {
      u32x4 b[6];
      unsigned mn = 0xFFFFFFFFu;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (s & 3) * SLOT + off1 + j * 1536u, 0, 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[4 + j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (s & 3) * SLOT + off2 + 2 * j * 1536u, 0, 16);
#pragma unroll
      for (int j = 0; j < 6; ++j) mn = min(mn, min(min(b[j].x, b[j].y), min(b[j].z, b[j].w)));
      // cells published so far carry s, older ones s - 4 (or 0)
      if (__all(mn == (unsigned)s)) { sink += b[0].x; break; }
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
  (void)hipMalloc(&buf, 8 * 4 * 24576); (void)hipMalloc(&out, 256 * 8);
  const int steps = 2000;
  for (int delay : {0, 30, 60, 100})
  for (int aux = 0; aux < 4; ++aux) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(buf, 0, 8 * 4 * 24576);
      if (aux == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, buf, steps, delay, out);
      if (aux == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, buf, steps, delay, out);    // sc0
      if (aux == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, buf, steps, delay, out);    // nt
      if (aux == 3) hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, buf, steps, delay, out);   // sc1
      (void)hipDeviceSynchronize();
    }
    std::vector<long long> h(256);
    (void)hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (auto v : h) mx = v > mx ? v : mx;
    const char *names[] = {"plain", "sc0  ", "nt   ", "sc1  "};
    printf("delay %4d ns  store %s: %.0f ns per step (hand-off = %.0f)\n", delay * 10, names[aux], mx * 10.0 / steps, mx * 10.0 / steps - delay * 10);
  }
  return 0;
}
