// valubench.hip -- issue rate of FP64 / FP32 VALU instructions on gfx950 (cycles per wave64
// instruction per CU with all SIMDs busy): is v_fma_f64 full rate or half rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
  double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double c = 0.999, d = 0.001;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                   "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 1) {
      asm volatile("v_add_f64 %0, %0, %8\n\tv_add_f64 %1, %1, %8\n\tv_add_f64 %2, %2, %8\n\tv_add_f64 %3, %3, %8\n\t"
                   "v_add_f64 %4, %4, %8\n\tv_add_f64 %5, %5, %8\n\tv_add_f64 %6, %6, %8\n\tv_add_f64 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
    } else if (MODE == 2) {
      asm volatile("v_mul_f64 %0, %0, %8\n\tv_mul_f64 %1, %1, %8\n\tv_mul_f64 %2, %2, %8\n\tv_mul_f64 %3, %3, %8\n\t"
                   "v_mul_f64 %4, %4, %8\n\tv_mul_f64 %5, %5, %8\n\tv_mul_f64 %6, %6, %8\n\tv_mul_f64 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else {
      float *f = (float *)&a0;
      asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(f[0]), "+v"(f[1]) : "v"(0.999f), "v"(0.001f));
      asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(f[0]), "+v"(f[1]) : "v"(0.999f), "v"(0.001f));
      asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(f[0]), "+v"(f[1]) : "v"(0.999f), "v"(0.001f));
      asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(f[0]), "+v"(f[1]) : "v"(0.999f), "v"(0.001f));
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name) {
  double *out; const int blocks = 256 * 12, iters = 20000;
  CK(hipMalloc(&out, blocks * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE><<<blocks, 256>>>(out, 100);
  CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double instr_per_cu = (double)blocks * 4 * iters * 8 / 256.0;
  printf("%s: %.2f ms, %.2f cycles (at 2.4 GHz) per wave64 instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
  CK(hipFree(out));
}

int main() {
  run<0>("v_fma_f64");
  run<1>("v_add_f64");
  run<2>("v_mul_f64");
  run<3>("v_fma_f32");
  return 0;
}
