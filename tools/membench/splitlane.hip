// splitlane.hip -- would a tile with only 3 contiguous lane bits (128-B runs) and 3
// "high" lane bits stream as fast as the 6-contiguous-lane-bit tile (1-KiB runs)?
// tile = lane bits {0,1,2} + {h0,h1,h2} + 5 register bits [p, p+5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint64_t ins0(uint64_t j, int pos, int width) {
  const uint64_t low = (1ull << pos) - 1;
  return ((j & ~low) << width) | (j & low);
}

__global__ __launch_bounds__(256) void k_split(double2 *__restrict__ p, uint64_t ntiles, int h, int pbit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-lane offset: low 3 bits contiguous, high 3 lane bits at [h, h+3)
  const uint64_t loff = (uint64_t)(lane & 7) | ((uint64_t)(lane >> 3) << h);
  for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave; w < ntiles; w += (uint64_t)gridDim.x * 4) {
    uint64_t j = w << 3;                 // free bits above the 3 contiguous lane bits
    // insert zeros at [h,h+3) and [pbit,pbit+5)  (h < pbit assumed or h > pbit+4)
    if (h < pbit) { j = ins0(j, h, 3); j = ins0(j, pbit, 5); } else { j = ins0(j, pbit, 5); j = ins0(j, h, 3); }
    double2 a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[k] = p[(j | ((uint64_t)k << pbit)) + loff];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      double2 t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
      p[(j | ((uint64_t)k << pbit)) + loff] = t;
    }
  }
}

int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 30;
  uint64_t n = 1ull << nb; size_t bytes = n * 16;
  double2 *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint64_t ntiles = n >> 11;
  struct { int h, p; } cases[] = {{3, 6}, {3, 11}, {11, 14}, {6, 9}, {6, 16}, {11, 16}, {16, 19}, {8, 11}, {12, 20}, {16, 21}, {19, 22}, {22, 25}, {26, 19}, {27, 16}, {20, 6}, {27, 6}};
  for (auto c : cases) {
    if (c.p + 5 > nb || c.h + 3 > nb) continue;
    hipLaunchKernelGGL(k_split, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, p, ntiles, c.h, c.p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_split, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, p, ntiles, c.h, c.p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("lane-high [%2d..%2d] regbits [%2d..%2d] : %7.3f ms %7.1f GB/s\n", c.h, c.h + 2, c.p, c.p + 4, ms, 2.0 * bytes / ms / 1e6);
  }
  return 0;
}
