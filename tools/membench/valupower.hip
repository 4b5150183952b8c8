// valupower.hip -- what does a VALU instruction COST IN POWER?  The op-heavy sweeps are bound by the socket's 1 400 W
// (DESIGN 4.5), so the price of an instruction is its energy.  One instruction kind per run, all SIMDs busy with 3 waves
// each (k_sweep's occupancy), operands with varied mantissas; tools/probes/r04_valupower.py samples rocm-smi beside it.
//   usage: valupower MODE SECONDS     MODE: fma add mul mov64 mov32 dpp xor swap32 idle
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

#define REP8(s) s s s s s s s s
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k(double *out, int iters) {
  double a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (threadIdx.x * 8 + j) + 1e-9 * blockIdx.x;
  const double c = 0.99999991234, d = 1.2345678e-7;
  uint32_t *w = (uint32_t *)a;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a[j]) : "v"(c), "v"(d));
      if (MODE == 1) asm volatile(REP8("v_add_f64 %0, %0, %1\n\t") : "+v"(a[j]) : "v"(d));
      if (MODE == 2) asm volatile(REP8("v_mul_f64 %0, %0, %1\n\t") : "+v"(a[j]) : "v"(c));
      if (MODE == 3) asm volatile(REP8("v_mov_b64 %0, %1\n\tv_mov_b64 %1, %0\n\t") : "+v"(a[j]), "+v"(a[(j + 1) & 7]));
      if (MODE == 4) asm volatile(REP8("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %0\n\t") : "+v"(w[2 * j]), "+v"(w[2 * j + 1]));
      if (MODE == 5) asm volatile(REP8("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t") : "+v"(w[2 * j]), "+v"(w[2 * j + 1]));
      if (MODE == 6) asm volatile(REP8("v_xor_b32 %0, %1, %0\n\tv_xor_b32 %1, %0, %1\n\t") : "+v"(w[2 * j]), "+v"(w[2 * j + 1]));
      if (MODE == 7) asm volatile(REP8("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %1, %0\n\t") : "+v"(w[2 * j]), "+v"(w[2 * j + 1]));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j];
  if (s == 1.2345) out[0] = s;
}

template <int MODE> static void run(double secs, int per_iter) {
  double *out; CK(hipMalloc(&out, 64));
  const int blocks = 256 * 3, iters = 20000;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < secs) {
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double winstr = (double)launches * blocks * 4 * iters * per_iter;      // wave-instructions
  printf("wave-instructions/s %.4g  (%.3f per SIMD-cycle at 2.4 GHz)  seconds %.2f\n", winstr / el, winstr / el / (1024 * 2.4e9), el);
}

int main(int argc, char **argv) {
  const char *m = argc > 1 ? argv[1] : "fma";
  const double secs = argc > 2 ? atof(argv[2]) : 5.0;
  if (!strcmp(m, "fma")) run<0>(secs, 64);
  else if (!strcmp(m, "add")) run<1>(secs, 64);
  else if (!strcmp(m, "mul")) run<2>(secs, 64);
  else if (!strcmp(m, "mov64")) run<3>(secs, 128);
  else if (!strcmp(m, "mov32")) run<4>(secs, 128);
  else if (!strcmp(m, "dpp")) run<5>(secs, 128);
  else if (!strcmp(m, "xor")) run<6>(secs, 128);
  else if (!strcmp(m, "swap32")) run<7>(secs, 128);
  else { printf("idle\n"); }
  return 0;
}
