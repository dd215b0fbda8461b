#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
  extern __shared__ float sm[];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11));
    unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
    out[2*blockIdx.x] = xcc; out[2*blockIdx.x+1] = hw;
  }
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
  if (sm[(threadIdx.x+1)%256] < 0) out[0] = 0;
}
int main() {
  unsigned *out; hipMalloc(&out, 4096*8);
  for (int lds : {65536, 98304}) for (int nb : {256, 512}) {
    if (lds == 98304 && nb == 512) continue;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), lds, 0, out, 100000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2*nb); hipMemcpy(h.data(), out, 8*nb, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    for (int b = 0; b < nb; ++b) { unsigned key = (h[2*b] << 16) | ((h[2*b+1] >> 8) & 0xFF); cu[key].push_back(b); }
    printf("lds %d blocks %d: distinct CUs %zu\n", lds, nb, cu.size());
    int n = 0; for (auto& kv : cu) { if (n++ < 12) { printf("  xcc %u cu/sh/se %02x:", kv.first >> 16, kv.first & 0xFF); for (int b : kv.second) printf(" %d", b); printf("\n"); } }
    std::map<int,int> hist; for (auto& kv : cu) hist[kv.second.size()]++;
    for (auto& kv : hist) printf("  %d CUs hold %d blocks\n", kv.second, kv.first);
    std::map<int,int> dif; for (auto& kv : cu) if (kv.second.size()==2) dif[kv.second[1]-kv.second[0]]++;
    for (auto& kv : dif) printf("  pair distance %d: %d\n", kv.first, kv.second);
    for (int b = 0; b < 4; ++b) printf("  b%d xcc %u hw %08x\n", b, h[2*b], h[2*b+1]);
  }
  return 0;
}
