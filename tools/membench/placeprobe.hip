// placeprobe.hip -- does the physical placement of the two state buffers show in a plain copy, and does re-allocating ONE of them flip it?
// (DESIGN 7 "placement": a handle's step time is fixed by the pages its two buffers got -- two modes with contiguous buffers.)
// A (16 GiB contiguous) stays; B is allocated, timed (copy A->B and B->A, full grid, 16 B per thread, non-temporal), freed behind a spacer
// of varying size, allocated again ...  usage: placeprobe [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_copy_nt(const double2 *__restrict__ s, double2 *__restrict__ d) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  double2 v;
  v.x = __builtin_nontemporal_load(&s[i].x);
  v.y = __builtin_nontemporal_load(&s[i].y);
  __builtin_nontemporal_store(v.x, &d[i].x);
  __builtin_nontemporal_store(v.y, &d[i].y);
}

static float time_copy(const double2 *s, double2 *d, size_t n, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_copy_nt, dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d);
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_copy_nt, dim3((unsigned)(n / 256)), dim3(256), 0, 0, s, d);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 10;
  const size_t n = 1ull << 30, bytes = n * 16;
  double2 *A = nullptr; CK(hipExtMallocWithFlags((void **)&A, bytes, hipDeviceMallocContiguous));
  CK(hipMemset(A, 0, bytes));
  const size_t spacers[] = {0, 64ull << 20, 1ull << 30, 3ull << 30, 2ull << 20, 7ull << 30, 512ull << 20, 0, 5ull << 30, 12ull << 30};
  for (int r = 0; r < rounds; ++r) {
    void *S = nullptr;
    const size_t sp = spacers[r % 10];
    if (sp) CK(hipMalloc(&S, sp));
    double2 *B = nullptr; CK(hipExtMallocWithFlags((void **)&B, bytes, hipDeviceMallocContiguous));
    CK(hipMemset(B, 0, bytes));
    const float ab = time_copy(A, B, n, 7), ba = time_copy(B, A, n, 7);
    printf("round %d spacer %6zu MiB  B at %p (A at %p)  copy A->B %.3f ms  B->A %.3f ms  (%.0f / %.0f GB/s)\n", r, sp >> 20, (void *)B, (void *)A, ab, ba,
           2.0 * bytes / ab / 1e6, 2.0 * bytes / ba / 1e6);
    fflush(stdout);
    CK(hipFree(B));
    if (S) CK(hipFree(S));
  }
  return 0;
}
