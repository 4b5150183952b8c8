// membench.hip -- in-place streaming read-modify-write variants on MI355X.
// Finds the launch shape / cache policy that maximises HBM GB/s for the access
// pattern of the gate kernels (each amplitude read once and written once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U, int MODE>
__global__ __launch_bounds__(256) void k_rmw(double2 *__restrict__ p, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * 256 * U;
  for (uint64_t base = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
    double2 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (MODE == 1 || MODE == 3) {
        a[u].x = __builtin_nontemporal_load(&p[base + 256ull * u].x);
        a[u].y = __builtin_nontemporal_load(&p[base + 256ull * u].y);
      } else a[u] = p[base + 256ull * u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double2 t; t.x = a[u].x * 0.6 - a[u].y * 0.8; t.y = a[u].x * 0.8 + a[u].y * 0.6;
      if (MODE == 2 || MODE == 3) {
        __builtin_nontemporal_store(t.x, &p[base + 256ull * u].x);
        __builtin_nontemporal_store(t.y, &p[base + 256ull * u].y);
      } else p[base + 256ull * u] = t;
    }
  }
}

// pair pattern: two streams 2^pbit apart (the k_pair access pattern)
template <int U>
__global__ __launch_bounds__(256) void k_pairs(double2 *__restrict__ p, uint64_t npairs, int pbit) {
  const uint64_t stride = (uint64_t)gridDim.x * 256 * U;
  const uint64_t q2 = 1ull << pbit, low = q2 - 1;
  for (uint64_t base = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; base < npairs; base += stride) {
    double2 a[U], b[U]; uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { uint64_t j = base + 256ull * u; idx[u] = ((j & ~low) << 1) | (j & low); a[u] = p[idx[u]]; b[u] = p[idx[u] | q2]; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double2 s, d; s.x = (a[u].x + b[u].x) * 0.7071; s.y = (a[u].y + b[u].y) * 0.7071; d.x = (a[u].x - b[u].x) * 0.7071; d.y = (a[u].y - b[u].y) * 0.7071;
      p[idx[u]] = s; p[idx[u] | q2] = d;
    }
  }
}

__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ s, double2 *__restrict__ d, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ s, double *out, uint64_t n) {
  double acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { double2 v = s[i]; acc += v.x + v.y; }
  if (acc == 1.2345) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_write(double2 *__restrict__ d, uint64_t n) {
  double2 v; v.x = 1; v.y = 2;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = v;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 30;
  uint64_t n = 1ull << nb; size_t bytes = n * 16;
  double2 *p, *q; double *o; CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&o, 8));
  CK(hipMemset(p, 0, bytes)); CK(hipMemset(q, 0, bytes));
  hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, p, n);
  auto rep = [&](const char *name, float ms, double gb) { printf("%-44s %8.3f ms  %8.1f GB/s\n", name, ms, gb / ms * 1e3 / 1e9 * 1e-0); };
  const double GB2 = 2.0 * bytes;
  for (int grid : {2048, 4096, 8192, 16384, 65536, 0}) {
    unsigned g4 = grid ? grid : (unsigned)(n / 1024), g8 = grid ? grid : (unsigned)(n / 2048), g2 = grid ? grid : (unsigned)(n / 512), g1 = grid ? grid : (unsigned)(n / 256);
    char nm[96];
    snprintf(nm, 96, "rmw U=4 plain grid=%u", g4); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<4, 0>), dim3(g4), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=8 plain grid=%u", g8); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<8, 0>), dim3(g8), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=2 plain grid=%u", g2); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<2, 0>), dim3(g2), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=1 plain grid=%u", g1); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<1, 0>), dim3(g1), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=4 nt-load grid=%u", g4); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<4, 1>), dim3(g4), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=4 nt-store grid=%u", g4); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<4, 2>), dim3(g4), dim3(256), 0, 0, p, n); }, 5), GB2);
    snprintf(nm, 96, "rmw U=4 nt-both grid=%u", g4); rep(nm, timeit([&] { hipLaunchKernelGGL((k_rmw<4, 3>), dim3(g4), dim3(256), 0, 0, p, n); }, 5), GB2);
  }
  for (int pb : {0, 3, 8, 12, 20, 29}) {
    char nm[96]; snprintf(nm, 96, "pairs U=4 full grid pbit=%d", pb);
    rep(nm, timeit([&] { hipLaunchKernelGGL((k_pairs<4>), dim3((unsigned)(n / 2 / 1024)), dim3(256), 0, 0, p, n / 2, pb); }, 5), GB2);
    snprintf(nm, 96, "pairs U=2 full grid pbit=%d", pb);
    rep(nm, timeit([&] { hipLaunchKernelGGL((k_pairs<2>), dim3((unsigned)(n / 2 / 512)), dim3(256), 0, 0, p, n / 2, pb); }, 5), GB2);
  }
  rep("copy p->q grid=8192", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, p, q, n); }, 5), GB2);
  rep("copy p->q grid=65536", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(65536), dim3(256), 0, 0, p, q, n); }, 5), GB2);
  rep("read only grid=8192", timeit([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, p, o, n); }, 5), (double)bytes);
  rep("write only grid=8192", timeit([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, p, n); }, 5), (double)bytes);
  rep("hipMemcpyDtoD", timeit([&] { CK(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, 0)); }, 5), GB2);
  return 0;
}
