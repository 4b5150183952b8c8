// bpermbench.hip -- issue rate of ds_bpermute_b32 / DPP moves / v_permlane32_swap on gfx950:
// how many cycles per wave64 instruction per CU when all SIMDs shuffle continuously.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters) {
  int lane = threadIdx.x & 63;
  int addr = (lane ^ 8) << 2;
  int v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("ds_bpermute_b32 %0, %4, %0\n\tds_bpermute_b32 %1, %4, %1\n\tds_bpermute_b32 %2, %4, %2\n\tds_bpermute_b32 %3, %4, %3\n\ts_waitcnt lgkmcnt(0)"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(addr));
    } else if (MODE == 1) {   // xor 8 within a row: row_mirror then row_half_mirror
      asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_mov_b32_dpp %2, %2 row_mirror row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_mirror row_mask:0xf bank_mask:0xf"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    } else if (MODE == 2) {
      asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    } else {
      asm volatile("v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3;
}

template <int MODE> void run(const char *name, int per_iter) {
  int *out; const int blocks = 256 * 8, iters = 20000;
  CK(hipMalloc(&out, blocks * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(out, 100);
  CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double instr_per_cu = (double)blocks * 4 * iters * per_iter / 256.0;
  printf("%s: %.2f ms, %.2f cycles (at 2.4 GHz) per wave64 instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  CK(hipFree(out));
}

int main() {
  run<0>("ds_bpermute_b32", 4);
  run<1>("v_mov_b32_dpp row_mirror", 4);
  run<2>("v_permlane32_swap_b32", 2);
  run<3>("v_permlane16_swap_b32", 2);
  return 0;
}
