// ldsxbench.hip -- what does a lane-bit <-> register-bit exchange THROUGH LDS cost a wave whose neighbours keep the
// VALU busy?  (k_sweep runs dense gates on lane bits 0..3 as DPP butterflies: 192-320 VALU instructions; an exchange
// through LDS costs no VALU issue at all: the lanes whose bit is 0 write their 16 slots with register bit r = 1, the
// other lanes their 16 slots with r = 0 -- ds_write_b128 under EXEC masks -- and read the partner lane's; the gate then
// runs as a 64-instruction register butterfly.)
//   tile = 32 complex128 per lane (128 VGPRs), 4 waves per workgroup, 3 workgroups per CU (12 waves / CU as k_sweep).
//   per iteration: K register butterflies (64 FP64 instructions each) + one of: nothing | LDS exchange on lane bit L |
//   DPP butterfly on lane bit L (192 / 320 instructions).
//   usage: ldsxbench ITER K
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int B> __device__ __forceinline__ void bfly(v2d (&a)[32]) {
#pragma unroll
  for (int h = 0; h < 16; ++h) {
    const int k0 = ((h >> B) << (B + 1)) | (h & ((1 << B) - 1)), k1 = k0 | (1 << B);
    a[k0] = a[k0] + a[k1];
    a[k1] = a[k0] - 2.0 * a[k1];
  }
}

// exchange lane bit L with register bit R through this wave's 8-KiB LDS region, two passes of 8 slot pairs; inline
// assembly (hipcc turns the two-sided `if` into v_cndmask copies of the tile and spills): lanes with bit L = 0 write
// their slots with register bit R = 1, the others their slots with R = 0, each reads the partner lane's.
#define XW(j) "ds_write_b128 %[mine], %[h" #j "] offset:" #j "*1024\n"
#define XV(j) "ds_write_b128 %[mine], %[l" #j "] offset:" #j "*1024\n"
#define XR(j) "ds_read_b128 %[h" #j "], %[part] offset:" #j "*1024\n"
#define XS(j) "ds_read_b128 %[l" #j "], %[part] offset:" #j "*1024\n"
#define XOPS(j) [l##j] "+v"(a[lo[j]]), [h##j] "+v"(a[hi[j]])
template <int L, int R> __device__ __forceinline__ void lds_xchg(v2d (&a)[32], uint32_t mine, uint32_t part, uint64_t m0, uint64_t m1) {
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    int lo[8], hi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int h = pass * 8 + j; lo[j] = ((h >> R) << (R + 1)) | (h & ((1 << R) - 1)); hi[j] = lo[j] | (1 << R); }
    asm volatile(
      "s_mov_b64 exec, %[m0]\n" XW(0) XW(1) XW(2) XW(3) XW(4) XW(5) XW(6) XW(7)
      "s_mov_b64 exec, %[m1]\n" XV(0) XV(1) XV(2) XV(3) XV(4) XV(5) XV(6) XV(7)
      "s_waitcnt lgkmcnt(0)\n"
      XS(0) XS(1) XS(2) XS(3) XS(4) XS(5) XS(6) XS(7)
      "s_mov_b64 exec, %[m0]\n" XR(0) XR(1) XR(2) XR(3) XR(4) XR(5) XR(6) XR(7)
      "s_mov_b64 exec, -1\n"
      "s_waitcnt lgkmcnt(0)\n"
      : XOPS(0), XOPS(1), XOPS(2), XOPS(3), XOPS(4), XOPS(5), XOPS(6), XOPS(7)
      : [mine] "v"(mine), [part] "v"(part), [m0] "s"(m0), [m1] "s"(m1) : "memory");
  }
}

template <int L> __device__ __forceinline__ void dpp_bfly(v2d (&a)[32], double s) {
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    v2d q;
    q.x = __shfl_xor(a[k].x, 1 << L);          // (the compiler picks DPP for xor 1, 2, 8; checked in the ISA)
    q.y = __shfl_xor(a[k].y, 1 << L);
    a[k].x = __builtin_fma(s, a[k].x, q.x);
    a[k].y = __builtin_fma(s, a[k].y, q.y);
  }
}

template <int MODE, int L> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_x(double *out, int iters, int K) {
  extern __shared__ v2d lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v2d a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) { a[k].x = 1e-30 * (lane + k); a[k].y = 1e-30 * (k - lane); }
  const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds;
  const uint32_t mine = lbase + wave * 8192 + lane * 16, partner = lbase + wave * 8192 + (lane ^ (1 << L)) * 16;
  const bool x = (lane >> L) & 1;
  const uint64_t m1 = __ballot(x), m0 = ~m1;
  const double s = x ? -1.0 : 1.0;
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < K; ++k) { bfly<0>(a); bfly<1>(a); bfly<2>(a); bfly<3>(a); }
    if (MODE == 1) { lds_xchg<L, 4>(a, mine, partner, m0, m1); }
    if (MODE == 2) { dpp_bfly<L>(a, s); }
#pragma unroll
    for (int k = 0; k < 32; ++k) { a[k].x *= 0.125; a[k].y *= 0.125; }
  }
  double acc = 0;
#pragma unroll
  for (int k = 0; k < 32; ++k) acc += a[k].x + a[k].y;
  if (acc == 12345.678) out[0] = acc;
}

template <int MODE, int L> static float run(double *out, int iters, int K) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(256 * 3), block(256);
  hipLaunchKernelGGL((k_x<MODE, L>), grid, block, 4 * 8192, 0, out, iters / 8 + 1, K);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_x<MODE, L>), grid, block, 4 * 8192, 0, out, iters, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  double *out; CK(hipMalloc(&out, 64));
  for (int K : {0, 1, 2, 4}) {
    const float base = run<0, 0>(out, iters, K);
    printf("K=%d (%3d FP64 instr + 64 scale per iteration)  base %.3f ms = %.0f ns/iter/wave-slot\n", 4 * K, 256 * K, base, base * 1e6 / iters);
    float t;
#define ROW(L) t = run<1, L>(out, iters, K); printf("  lane bit %d: LDS exchange +%.0f ns/iter", L, (t - base) * 1e6 / iters); \
               t = run<2, L>(out, iters, K); printf("   DPP butterfly +%.0f ns/iter\n", (t - base) * 1e6 / iters);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
  }
  return 0;
}
