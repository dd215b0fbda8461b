#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// blocks 0 and `partner` ping-pong a 16-byte message through global memory (plain store or sc1 store, sc1 load poll)
__global__ __launch_bounds__(256, 2) void k(unsigned* buf, int* out, int iters, int partner, int wt, int delay) {
  extern __shared__ float sm[];
  if (blockIdx.x != 0 && blockIdx.x != partner) return;
  const bool me0 = blockIdx.x == 0;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 1 << 20, 0x00020000);
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    out[me0 ? 2 : 3] = xcc;
  }
  if (threadIdx.x >= 64) return;
  unsigned long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (me0) {
      u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
      if (wt) __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16, 0, 0);
      for (;;) { u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, 4096 + threadIdx.x * 16, 0, 16); if (__all(r.x == (unsigned)i && r.w == (unsigned)i)) break; }
    } else {
      for (;;) { u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, 0, 16); if (__all(r.x == (unsigned)i && r.w == (unsigned)i)) break; }
      if (delay) { unsigned long long d0 = wall_clock64(); while (wall_clock64() - d0 < (unsigned long long)delay) ; }
      u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i};
      if (wt) __builtin_amdgcn_raw_buffer_store_b128(v, rs, 4096 + threadIdx.x * 16, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(v, rs, 4096 + threadIdx.x * 16, 0, 0);
    }
  }
  unsigned long long t1 = wall_clock64();
  if (me0 && threadIdx.x == 0) out[0] = (int)((t1 - t0) * 10 / iters);   // ns per round trip
}
int main() {
  unsigned* buf; hipMalloc(&buf, 1 << 20); int* out; hipMalloc(&out, 64);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int partner : {8, 1}) for (int wt : {0}) for (int delay : {0, 100}) {
    hipMemset(buf, 0, 1 << 20); hipMemset(out, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(512), dim3(256), 65536, 0, buf, out, 20000, partner, wt, delay);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("[kernel %.3f ms for 20000 round trips = %.0f ns each] ", ms, ms * 1e6 / 20000);
    int h[4]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("delay %d ticks; partner block %3d (xcc %d vs %d) %s store: round trip %d ns -> one way %d ns\n", delay, partner, h[2], h[3], wt ? "sc1  " : "plain", h[0], h[0] / 2);
  }
  return 0;
}
