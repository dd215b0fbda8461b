// Does the tile shape of the operand packs matter to HBM?  A pack reads [R x C] fp32 and writes two fp16 planes as
// [C/16][plane][R][16] (rows form) or [R/16][plane][C][16] (transposed form); gemm_pk.hip's kernels move 64 x 64 tiles
// through LDS (256-byte row segments in, 2 KiB runs out).  This times the same data movement (plain hi/lo split, no
// scales) for tiles of 64 rows x TW columns.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/pack_tile tools/experiments/ub/pack_tile.hip && /tmp/pack_tile
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_store(const float *x, char *dst, size_t plane_stride) {
  unsigned h[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 a = (_Float16)x[2 * i], b = (_Float16)x[2 * i + 1];
    const _Float16 c = (_Float16)(x[2 * i] - (float)a), d = (_Float16)(x[2 * i + 1] - (float)b);
    h[i] = (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
    l[i] = (unsigned)__builtin_bit_cast(unsigned short, c) | ((unsigned)__builtin_bit_cast(unsigned short, d) << 16);
  }
  u32x4 *d0 = reinterpret_cast<u32x4 *>(dst), *d1 = reinterpret_cast<u32x4 *>(dst + plane_stride);
  d0[0] = (u32x4){h[0], h[1], h[2], h[3]}; d0[1] = (u32x4){h[4], h[5], h[6], h[7]};
  d1[0] = (u32x4){l[0], l[1], l[2], l[3]}; d1[1] = (u32x4){l[4], l[5], l[6], l[7]};
}
// rows form: packed row = source row
template <int TW>
__global__ __launch_bounds__(256) void rows_k(const float *src, int R, int C, char *dst) {
  __shared__ __attribute__((aligned(16))) float tile[64][TW + 4];
  const int tid = threadIdx.x, r0 = blockIdx.y * 64, c0 = blockIdx.x * TW;
#pragma unroll
  for (int j = 0; j < TW / 16; ++j) {
    const int i = tid + 256 * j, r = i / (TW / 4), c4 = (i % (TW / 4)) * 4;
    *reinterpret_cast<float4 *>(&tile[r][c4]) = *reinterpret_cast<const float4 *>(src + (size_t)(r0 + r) * C + c0 + c4);
  }
  __syncthreads();
  const int r = tid & 63;
  const size_t plane = (size_t)R * 32;
#pragma unroll
  for (int kbl = tid >> 6; kbl < TW / 16; kbl += 4) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4 *>(&tile[r][kbl * 16 + 4 * i]);
      x[4 * i] = v.x; x[4 * i + 1] = v.y; x[4 * i + 2] = v.z; x[4 * i + 3] = v.w;
    }
    split_store(x, dst + ((size_t)(c0 / 16 + kbl) * 2) * plane + (size_t)(r0 + r) * 32, plane);
  }
}
// transposed form: packed row = source column, k = source row
template <int TW>
__global__ __launch_bounds__(256) void cols_k(const float *src, int R, int C, char *dst) {
  __shared__ __attribute__((aligned(16))) float tile[64][TW + 4];
  const int tid = threadIdx.x, k0 = blockIdx.y * 64, c0 = blockIdx.x * TW;
#pragma unroll
  for (int j = 0; j < TW / 16; ++j) {
    const int i = tid + 256 * j, r = i / (TW / 4), c4 = (i % (TW / 4)) * 4;
    *reinterpret_cast<float4 *>(&tile[r][c4]) = *reinterpret_cast<const float4 *>(src + (size_t)(k0 + r) * C + c0 + c4);
  }
  __syncthreads();
  const int kbl = tid >> 6;
  const size_t plane = (size_t)C * 32;
#pragma unroll
  for (int c = tid & 63; c < TW; c += 64) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = tile[kbl * 16 + i][c];
    split_store(x, dst + ((size_t)(k0 / 16 + kbl) * 2) * plane + (size_t)(c0 + c) * 32, plane);
  }
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) f();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 20;
}
int main() {
  for (int R : {16000, 32000}) {
    const int C = 2048;
    float *src; char *dst;
    (void)hipMalloc(&src, (size_t)R * C * 4); (void)hipMalloc(&dst, (size_t)R * C * 4 + (1 << 20));
    (void)hipMemset(src, 0x3c, (size_t)R * C * 4);
    const double gb = (double)R * C * 8 / 1e9;
    float t;
#define RUN(name, K, TW)                                                                                   \
    t = timeit([&] { hipLaunchKernelGGL(K<TW>, dim3(C / TW, R / 64), dim3(256), 0, 0, src, R, C, dst); }); \
    printf("[%d x %d] %s tile 64 x %3d: %7.1f us  %5.2f TB/s\n", R, C, name, TW, t * 1e3, gb / t);
    RUN("rows ", rows_k, 64) RUN("rows ", rows_k, 128) RUN("rows ", rows_k, 256)
    RUN("cols ", cols_k, 64) RUN("cols ", cols_k, 128) RUN("cols ", cols_k, 256)
    t = timeit([&] { (void)hipMemcpyAsync(dst, src, (size_t)R * C * 4, hipMemcpyDeviceToDevice, 0); });
    printf("[%d x %d] device copy of the fp32 array: %7.1f us  %5.2f TB/s\n", R, C, t * 1e3, gb / t);
    (void)hipFree(src); (void)hipFree(dst);
  }
  return 0;
}
