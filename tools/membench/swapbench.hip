// swapbench.hip -- issue rate of v_swap_b32 against v_mov_b32 / DPP moves / v_permlane swaps on gfx950, and what
// v_permlane16_swap / v_permlane32_swap do when both operands are the SAME register (X on lane bit 4 / 5 in one instruction?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(64) void k_rate(unsigned *out, long long *cyc, int iters) {
  unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = threadIdx.x * 7 + 2, d = threadIdx.x * 11 + 3;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP64(asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 1) { REP64(asm volatile("v_swap_b32 %0, %1\n\tv_swap_b32 %2, %3\n\tv_swap_b32 %0, %2\n\tv_swap_b32 %1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 2) { REP64(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 3) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 4) { REP64(asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %1, %1, %0" : "+v"(*(double *)&a), "+v"(*(double *)&c)); asm volatile("" : "+v"(b), "+v"(d));) }
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = a ^ b ^ c ^ d;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_same(unsigned *out) {
  unsigned v = threadIdx.x, w = threadIdx.x;
  asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(v));
  asm volatile("v_permlane16_swap_b32 %0, %0" : "+v"(w));
  out[threadIdx.x] = v;
  out[64 + threadIdx.x] = w;
}

int main() {
  unsigned *out; long long *cyc;
  CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&cyc, 8 * 4096));
  const char *names[] = {"v_mov_b32", "v_swap_b32", "v_mov_b32_dpp (in place)", "v_permlane32/16_swap", "v_add_f64 (x2 per group)"};
  const int per_iter[] = {256, 256, 256, 256, 128};
  for (int mode = 0; mode < 5; ++mode) {
    // 12 waves per CU (3 per SIMD): 256 CUs x 12
    const int blocks = 256 * 12, iters = 64;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(k_rate<3>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
        case 4: hipLaunchKernelGGL(k_rate<4>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters); break;
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c0; CK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
    const double ninstr = (double)per_iter[mode] * iters;                 // per wave
    // 3 waves share a SIMD: SIMD-cycles per wave-instruction = wave cycles / (3 x instructions)  (clock64 = 100 MHz ref? print both)
    printf("%-28s %8.3f ms for %d waves x %.0f instructions: %.2f ns per wave-instruction per SIMD (3 waves each); clock64 delta %lld\n", names[mode], ms, blocks, ninstr,
           ms * 1e6 / (3.0 * ninstr), c0);
  }
  hipLaunchKernelGGL(k_same, dim3(1), dim3(64), 0, 0, out);
  unsigned h[128]; CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  printf("v_permlane32_swap v, v : lane l holds"); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[l]); printf("  (X on lane bit 5 would give l ^ 32)\n");
  printf("v_permlane16_swap v, v : lane l holds"); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[64 + l]); printf("  (X on lane bit 4 would give l ^ 16)\n");
  int ok32 = 1, ok16 = 1; for (int l = 0; l < 64; ++l) { ok32 &= h[l] == (unsigned)(l ^ 32); ok16 &= h[64 + l] == (unsigned)(l ^ 16); }
  printf("same-register swap == X on the lane bit: permlane32 %s, permlane16 %s\n", ok32 ? "yes" : "NO", ok16 ? "yes" : "NO");
  return 0;
}
