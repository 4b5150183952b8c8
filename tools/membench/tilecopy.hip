// tilecopy.hip -- how much of a plain copy's rate does the SHAPE of a sweep cost?  placeprobe.hip: a full-grid copy of 16 GiB, one amplitude per
// thread, runs at 6.7-6.8 TB/s; k_sweep with an empty op stream at 6.2-6.4.  Here: each wave loads R x 1 KiB (a contiguous tile of R slots), waits
// for ALL of them (the ops need the whole tile), then stores R x 1 KiB to the second buffer -- R = 1 .. 32, W waves per workgroup, the waves of a
// workgroup on consecutive tiles; variant "pipe": the stores of slot k issued as soon as load k has arrived (what a copy does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int R, int W, bool PIPE, int ROT, int IL = 1, int IS = 1>
__global__ __launch_bounds__(64 * W) void k_tile(const double2 *__restrict__ s, double2 *__restrict__ d, unsigned blk_bits) {
  unsigned bi = blockIdx.x;
  if (ROT) bi = ((bi >> ROT) | (bi << (blk_bits - ROT))) & ((1u << blk_bits) - 1);
  const size_t tile = (size_t)bi * W + threadIdx.x / 64;
  // IL (loads) / IS (stores): groups of IL consecutive tiles interleave at 1-KiB granularity -- a wave's slots are IL KiB apart
  const size_t base = (tile / IL) * (size_t)(64 * R * IL) + (tile % IL) * 64 + (threadIdx.x & 63);
  const size_t sbase = (tile / IS) * (size_t)(64 * R * IS) + (tile % IS) * 64 + (threadIdx.x & 63);
  double2 v[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    v[k].x = __builtin_nontemporal_load(&s[base + (size_t)64 * IL * k].x);
    v[k].y = __builtin_nontemporal_load(&s[base + (size_t)64 * IL * k].y);
  }
  if (!PIPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (!PIPE) { asm volatile("" : "+v"(v[k].x), "+v"(v[k].y)); }
    __builtin_nontemporal_store(v[k].x, &d[sbase + (size_t)64 * IS * k].x);
    __builtin_nontemporal_store(v[k].y, &d[sbase + (size_t)64 * IS * k].y);
  }
}

template <typename F> static float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

template <int R, int W, bool PIPE, int ROT, int IL = 1, int IS = 1> static void run(const double2 *A, double2 *B, size_t n) {
  const unsigned blocks = (unsigned)(n / (64 * R) / W);
  unsigned bits = 0; while ((1u << bits) < blocks) ++bits;
  const float ms = timeit([&] { hipLaunchKernelGGL((k_tile<R, W, PIPE, ROT, IL, IS>), dim3(blocks), dim3(64 * W), 0, 0, A, B, bits); }, 7);
  printf("R=%2d slots/wave  W=%d waves/block  %s  rot %d  interleave load %3d store %3d : %.3f ms  %.0f GB/s\n", R, W, PIPE ? "pipelined" : "wait-all ", ROT, IL, IS, ms, 2.0 * n * 16 / ms / 1e6);
  fflush(stdout);
}

int main() {
  const size_t n = 1ull << 30, bytes = n * 16;
  double2 *A, *B;
  CK(hipExtMallocWithFlags((void **)&A, bytes, hipDeviceMallocContiguous));
  CK(hipExtMallocWithFlags((void **)&B, bytes, hipDeviceMallocContiguous));
  CK(hipMemset(A, 0, bytes)); CK(hipMemset(B, 0, bytes));
  run<1, 4, true, 0>(A, B, n);
  run<32, 2, false, 0>(A, B, n);
  run<32, 2, false, 3>(A, B, n);
  run<32, 2, false, 0, 16, 16>(A, B, n);
  run<32, 2, false, 0, 64, 64>(A, B, n);
  run<32, 2, false, 0, 256, 256>(A, B, n);
  run<32, 2, false, 0, 1024, 1024>(A, B, n);
  run<32, 2, false, 3, 16, 16>(A, B, n);
  run<32, 2, false, 3, 64, 64>(A, B, n);
  run<32, 2, false, 3, 256, 256>(A, B, n);
  run<32, 2, false, 0, 64, 1>(A, B, n);
  run<32, 2, false, 0, 1, 64>(A, B, n);
  run<32, 2, false, 3, 64, 1>(A, B, n);
  run<32, 2, false, 3, 1, 64>(A, B, n);
  run<32, 4, false, 0, 64, 64>(A, B, n);
  run<32, 1, false, 0, 64, 64>(A, B, n);
  run<16, 4, false, 0, 64, 64>(A, B, n);
  return 0;
}
