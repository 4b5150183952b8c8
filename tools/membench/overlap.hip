// overlap.hip -- does a register-tile sweep overlap one wave's arithmetic with the other waves' memory phases?
// Every wave loads a 32 KiB tile (32 x 1 KiB, non-temporal), runs ITERS rounds of 64 FP64 FMAs on it in
// place (= ITERS*64 VALU instructions = ITERS*256 issue cycles of its SIMD) and stores it back.
//   usage: overlap NBITS WAVES_PER_CU TILES_PER_WAVE ITERS...
//     WAVES_PER_CU is enforced through the workgroup's LDS size (2 waves per workgroup);
//     TILES_PER_WAVE = 1: one tile per wave, grid = all tiles; >1: consecutive tiles in a loop; 0 = persistent
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(128) void k_ov(v2d *__restrict__ psi, uint64_t ntiles, int tiles_per_wave, int iters, double cr, double ci) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63;
  uint64_t wave = (uint64_t)blockIdx.x * 2 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 2;
  if (lane == 99) lds[0] = 1;
  for (uint64_t t = (tiles_per_wave ? wave * tiles_per_wave : wave), cnt = 0; t < ntiles && (tiles_per_wave == 0 || cnt < (uint64_t)tiles_per_wave);
       t += (tiles_per_wave ? 1 : nwaves), ++cnt) {
    v2d *base = psi + t * 2048 + lane;
    v2d a[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) a[k] = __builtin_nontemporal_load(base + 64 * k);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        a[k].x = __builtin_fma(a[k].x, cr, ci);
        a[k].y = __builtin_fma(a[k].y, cr, -ci);
      }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) __builtin_nontemporal_store(a[k], base + 64 * k);
  }
}

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 30;
  const int wpc = argc > 2 ? atoi(argv[2]) : 12;
  const int tpw = argc > 3 ? atoi(argv[3]) : 1;
  const uint64_t n = 1ull << nb; const size_t bytes = n * 16;
  v2d *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t ntiles = n >> 11;
  const int blocks_per_cu = wpc / 2;
  const size_t lds = (160 * 1024 / blocks_per_cu) & ~1023u;       // so that exactly blocks_per_cu workgroups fit a CU
  CK(hipFuncSetAttribute((const void *)k_ov, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void *)k_ov));
  printf("# %d qubits, %d waves/CU (LDS %zu B per 2-wave workgroup), tiles/wave %d, kernel VGPRs %d\n", nb, wpc, lds, tpw, fa.numRegs);
  for (int a = 4; a < argc; ++a) {
    const int iters = atoi(argv[a]);
    const uint64_t grid = tpw ? (ntiles / tpw + 1) / 2 : (uint64_t)256 * blocks_per_cu;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_ov, dim3((unsigned)grid), dim3(128), lds, 0, p, ntiles, tpw, iters, 1.0000001, 1e-9);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) best = ms < best ? ms : best;
    }
    const double valu_ms = (double)ntiles * iters * 64 * 4 / (1024.0 * 2.4e9) * 1e3;
    printf("iters %4d  (%6d VALU instr per tile, VALU-only time %.2f ms)  %.3f ms  %.0f GB/s\n", iters, iters * 64, valu_ms, best, 2.0 * bytes / best * 1e-6);
  }
  return 0;
}
