// l2hot: every workgroup of an XCD (32 of them, 4 waves each) reads the SAME 24 KiB from its L2 with sc1 loads, as
// the forward recurrence does with the published h planes.  How long does one round take, by line stride?
// build: hipcc --offload-arch=gfx950 -O3 l2hot.hip -o l2hot ; run: ./l2hot
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(char *buf, size_t per_xcd, int stride, int iters, int nload, int priv, long long *out) {
  const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf + xcd * per_xcd, 0, (int)per_xcd, 0x00020000);
  unsigned acc = 0;
  // line index of (wave, load i, lane): 8 lines per load instruction
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < nload) {
        const unsigned line = (unsigned)((w * nload + i) * 8 + (lane >> 3)) + (priv ? slot * 192u : 0u);
        v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, line * (unsigned)stride + (lane & 7) * 16u + (it & 1) * 0u, 0, 16);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < nload) acc += v[i].x ^ v[i].w;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0 + (acc == 0x12345 ? 1 : 0);
}

int main() {
  const size_t per_xcd = 64u << 20;
  char *buf; long long *out;
  hipMalloc(&buf, 8 * per_xcd); hipMalloc(&out, 256 * 8);
  hipMemset(buf, 1, 8 * per_xcd);
  const int iters = 2000;
  for (int priv = 0; priv < 2; ++priv)
  for (int nload = 6; nload >= 2; nload -= 4)
  for (int stride : {128, 256, 384, 512, 1024, 2048, 4096, 4224, 8192, 16384, 16512}) {
    if (priv && (size_t)stride * 192 * 32 > per_xcd) continue;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, buf, per_xcd, stride, iters, nload, priv, out);
      hipDeviceSynchronize();
    }
    std::vector<long long> h(256);
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    long long mx = 0, mn = 1ll << 60;
    for (auto v : h) { mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
    printf("%s nload %d stride %5d: %.0f ns per round (min %.0f)  -> %.2f TB/s per XCD\n", priv ? "private" : "shared ", nload, stride,
           mx * 10.0 / iters, mn * 10.0 / iters, 32.0 * 4 * nload * 1024 / (mx * 10.0 / iters) / 1000.0);
  }
  return 0;
}
