// stridebench.hip -- does an address swizzle cure the slow register-bit positions?
// Each wave RMWs 32 x 1 KiB runs whose addresses differ in 5 consecutive address bits
// [p, p+5) (the k_sweep access pattern).  Variant SWZ xors those 5 bits into the
// byte-address bits [swz_lo, swz_lo+5) (staying above the 1-KiB run), which spreads
// the 32 runs of a wave over different DRAM channels/banks if the mapping keys on
// those low bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_tiles(double2 *__restrict__ p, uint64_t ntiles, int pbit, int swz_lo) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t w = (uint64_t)blockIdx.x * 4 + wave; w < ntiles; w += (uint64_t)gridDim.x * 4) {
    // amplitude index of the tile base: insert 5 zero bits at [pbit, pbit+5) into (w << 6)
    const uint64_t j = w << 6;
    const uint64_t low = (1ull << pbit) - 1;
    const uint64_t base = ((j & ~low) << 5) | (j & low);
    double2 a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      uint64_t idx = base | ((uint64_t)k << pbit);
      if (swz_lo >= 0) idx ^= (uint64_t)k << swz_lo;
      a[k] = __builtin_nontemporal_load(&p[idx + lane].x) == 0.0 ? p[idx + lane] : p[idx + lane];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      uint64_t idx = base | ((uint64_t)k << pbit);
      if (swz_lo >= 0) idx ^= (uint64_t)k << swz_lo;
      double2 t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
      p[idx + lane] = t;
    }
  }
}

int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 30;
  uint64_t n = 1ull << nb; size_t bytes = n * 16;
  double2 *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint64_t ntiles = n >> 11;
  for (int pbit : {6, 11, 16, 19, 20, 21, 22, 23, 24, 25}) {
    if (pbit + 5 > nb) continue;
    for (int swz : {-1, 6, 8, 10, 12}) {
      if (swz >= 0 && swz + 5 > pbit) continue;
      hipLaunchKernelGGL(k_tiles, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, p, ntiles, pbit, swz);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_tiles, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, p, ntiles, pbit, swz);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("regbits [%2d..%2d] swizzle_into %3d : %7.3f ms %7.1f GB/s\n", pbit, pbit + 4, swz, ms, 2.0 * bytes / ms / 1e6);
    }
  }
  return 0;
}
