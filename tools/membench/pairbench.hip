// pairbench.hip -- launch shapes for the ONE-GATE-PER-LAUNCH kernels (kernels_gate.hip.h: k_pair, k_diag) on MI355X.
// Round 4: the per-gate kernels ran at 0.68 of the 8 TB/s HBM peak (profiles/r03/bench_unfused.json) while a fused
// sweep with one op -- the same bytes -- ran at 0.78; this probe times the candidate shapes on the same access pattern
// (in place, complex128, 2^nb amplitudes) for a set of target bits, so that the shape is chosen by measurement.
//   pair  variants: A  = round-3 shape (one pair per thread, 256-thread blocks, one chunk per block)
//                   Bu = U pairs per thread, the U chunks of a block 256 items apart (round-1 "U" variant)
//                   Tu = wave tiles: a wave owns 64*U consecutive work items (U KiB contiguous per stream), all 2U loads first
//                   Tu+rot = the same with the block index rotated by 3 bits (consecutive workgroups stream from 8 regions)
//                   Pu = persistent grid (4 waves / SIMD), wave tiles in a grid-stride loop
//   diag  variants: the same shapes on the one-stream pattern (bit-inserted enumeration of the touched quarter / half)
// Build: hipcc --offload-arch=gfx950 -O3 -o pairbench pairbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double2 ldnt(const double2 *p) {
  double2 v;
  v.x = __builtin_nontemporal_load(&p->x);
  v.y = __builtin_nontemporal_load(&p->y);
  return v;
}
__device__ __forceinline__ void stnt(double2 *p, double2 v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
}
__device__ __forceinline__ void bfly(double2 &a, double2 &b) {   // a generic complex 2x2 (same flops as k_pair)
  const double g0r = 0.6, g0i = 0.1, g1r = -0.3, g1i = 0.7, g2r = 0.2, g2i = -0.5, g3r = 0.4, g3i = 0.3;
  double2 t1, t2;
  t1.x = (g0r * a.x - g0i * a.y) + (g1r * b.x - g1i * b.y);
  t1.y = (g0r * a.y + g0i * a.x) + (g1r * b.y + g1i * b.x);
  t2.x = (g2r * a.x - g2i * a.y) + (g3r * b.x - g3i * b.y);
  t2.y = (g2r * a.y + g2i * a.x) + (g3r * b.y + g3i * b.x);
  a = t1;
  b = t2;
}
__device__ __forceinline__ uint64_t ins0(uint64_t j, int p) {
  const uint64_t low = (1ull << p) - 1ull;
  return ((j & ~low) << 1) | (j & low);
}

// B<U>: thread handles items base + 256*u (block chunk = 256*U items), one chunk per block
template <int U>
__global__ __launch_bounds__(256) void k_pair_B(double2 *__restrict__ psi, int p) {
  const uint64_t base = (uint64_t)blockIdx.x * (256ull * U) + threadIdx.x;
  const uint64_t q2 = 1ull << p;
  double2 a[U], b[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = ins0(base + 256ull * u, p);
    a[u] = ldnt(&psi[idx[u]]);
    b[u] = ldnt(&psi[idx[u] | q2]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    bfly(a[u], b[u]);
    stnt(&psi[idx[u]], a[u]);
    stnt(&psi[idx[u] | q2], b[u]);
  }
}

// T<U,WPB,ROT>: a wave owns 64*U consecutive items; block = WPB waves with consecutive tiles; optional block rotation
template <int U, int WPB, int ROT>
__global__ __launch_bounds__(64 * WPB) void k_pair_T(double2 *__restrict__ psi, int p, int blk_bits) {
  uint64_t bi = blockIdx.x;
  if (ROT) bi = ((bi >> ROT) | (bi << (blk_bits - ROT))) & ((1ull << blk_bits) - 1);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = ((bi * WPB + wave) * U) * 64ull + lane;
  const uint64_t q2 = 1ull << p;
  double2 a[U], b[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx[u] = ins0(base + 64ull * u, p);
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ldnt(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) b[u] = ldnt(&psi[idx[u] | q2]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    bfly(a[u], b[u]);
    stnt(&psi[idx[u]], a[u]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) stnt(&psi[idx[u] | q2], b[u]);
}

// P<U>: persistent grid, wave tiles in a grid-stride loop
template <int U>
__global__ __launch_bounds__(256) void k_pair_P(double2 *__restrict__ psi, uint64_t nwork, int p) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t q2 = 1ull << p;
  const uint64_t ntile = nwork / (64ull * U);
  for (uint64_t t = (uint64_t)blockIdx.x * 4 + wave; t < ntile; t += (uint64_t)gridDim.x * 4) {
    const uint64_t base = t * U * 64ull + lane;
    double2 a[U], b[U];
    uint64_t idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = ins0(base + 64ull * u, p);
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = ldnt(&psi[idx[u]]);
#pragma unroll
    for (int u = 0; u < U; ++u) b[u] = ldnt(&psi[idx[u] | q2]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bfly(a[u], b[u]);
      stnt(&psi[idx[u]], a[u]);
      stnt(&psi[idx[u] | q2], b[u]);
    }
  }
}

// ---- one-stream (diagonal) pattern: work item j -> index with ones inserted at bits c and t (c > t) --------------
__device__ __forceinline__ uint64_t ins2(uint64_t j, int lo, int hi) {   // insert a 1 at lo, then a 1 at hi (lo < hi)
  uint64_t m = (1ull << lo) - 1ull;
  j = ((j & ~m) << 1) | (j & m) | (1ull << lo);
  m = (1ull << hi) - 1ull;
  return ((j & ~m) << 1) | (j & m) | (1ull << hi);
}
template <int U, int WPB, int ROT>
__global__ __launch_bounds__(64 * WPB) void k_diag_T(double2 *__restrict__ psi, int lo, int hi, int blk_bits) {
  uint64_t bi = blockIdx.x;
  if (ROT) bi = ((bi >> ROT) | (bi << (blk_bits - ROT))) & ((1ull << blk_bits) - 1);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t base = ((bi * WPB + wave) * U) * 64ull + lane;
  double2 a[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) idx[u] = ins2(base + 64ull * u, lo, hi);
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ldnt(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    double2 t;
    t.x = 0.6 * a[u].x - 0.8 * a[u].y;
    t.y = 0.6 * a[u].y + 0.8 * a[u].x;
    stnt(&psi[idx[u]], t);
  }
}
template <int U>
__global__ __launch_bounds__(256) void k_diag_B(double2 *__restrict__ psi, int lo, int hi) {
  const uint64_t base = (uint64_t)blockIdx.x * (256ull * U) + threadIdx.x;
  double2 a[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    idx[u] = ins2(base + 256ull * u, lo, hi);
    a[u] = ldnt(&psi[idx[u]]);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    double2 t;
    t.x = 0.6 * a[u].x - 0.8 * a[u].y;
    t.y = 0.6 * a[u].y + 0.8 * a[u].x;
    stnt(&psi[idx[u]], t);
  }
}

// ---- big register tiles: does a wave that holds 64 KiB (RB = 6, complex128: 256 data registers, one wave per SIMD)
// still stream?  (verdict r3 #5: the membench to run before any 2-sweep QFT attempt)
template <int ROWS>
__global__ __launch_bounds__(64) void k_rmw_rows(double2 *__restrict__ psi, int blk_bits, int rot) {
  uint64_t bi = blockIdx.x;
  if (rot) bi = ((bi >> rot) | (bi << (blk_bits - rot))) & ((1ull << blk_bits) - 1);
  const uint64_t base = bi * (64ull * ROWS) + threadIdx.x;
  double2 a[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) a[r] = ldnt(&psi[base + 64ull * r]);
#pragma unroll
  for (int r = 0; r < ROWS; r += 2) {
    double2 s, d;
    s.x = a[r].x + a[r + 1].x; s.y = a[r].y + a[r + 1].y;
    d.x = a[r].x - a[r + 1].x; d.y = a[r].y - a[r + 1].y;
    a[r] = s; a[r + 1] = d;
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) stnt(&psi[base + 64ull * r], a[r]);
}

__global__ __launch_bounds__(256) void k_fill(double2 *d, uint64_t n) {
  double2 v; v.x = 1e-5; v.y = -2e-5;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = v;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}
static int log2u(uint64_t v) { int g = 0; while ((1ull << g) < v) ++g; return g; }

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 30;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  const uint64_t n = 1ull << nb;
  const size_t bytes = n * 16;
  double2 *psi;
  CK(hipMalloc(&psi, bytes));
  hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, psi, n);
  CK(hipDeviceSynchronize());
  const uint64_t npair = n / 2;
  auto rep = [&](const char *kind, const char *name, int p, float ms, double moved) {
    printf("%-5s %-22s p=%-5d %8.3f ms  %8.1f GB/s  frac %.3f\n", kind, name, p, ms, moved / ms / 1e6, moved / ms / 1e6 / 8000.0);
    fflush(stdout);
  };
  const double S2 = 2.0 * bytes;
  const int pbits[] = {0, 1, 2, 3, 5, 6, 8, 11, 14, 17, 20, 23, 26, 29};
  for (int p : pbits) {
    if (p >= nb) continue;
#define RUN_B(U) rep("pair", "B U=" #U, p, timeit([&] { hipLaunchKernelGGL((k_pair_B<U>), dim3((unsigned)(npair / (256 * U))), dim3(256), 0, 0, psi, p); }, reps), S2)
#define RUN_T(U, W, R) { const uint64_t blocks = npair / (64ull * U * W); rep("pair", "T U=" #U " wpb=" #W " rot=" #R, p, timeit([&] { hipLaunchKernelGGL((k_pair_T<U, W, R>), dim3((unsigned)blocks), dim3(64 * W), 0, 0, psi, p, log2u(blocks)); }, reps), S2); }
#define RUN_P(U, G) rep("pair", "P U=" #U " grid=" #G, p, timeit([&] { hipLaunchKernelGGL((k_pair_P<U>), dim3(G), dim3(256), 0, 0, psi, npair, p); }, reps), S2)
    RUN_B(1); RUN_B(2); RUN_B(4);
    RUN_T(2, 4, 0); RUN_T(4, 4, 0); RUN_T(8, 4, 0); RUN_T(16, 2, 0);
    RUN_T(4, 4, 3); RUN_T(8, 4, 3); RUN_T(8, 2, 3); RUN_T(16, 2, 3); RUN_T(16, 1, 3); RUN_T(8, 1, 6);
    RUN_P(4, 2048); RUN_P(8, 1024); RUN_P(8, 2048);
  }
  struct CT { int lo, hi; };
  const CT cts[] = {{2, 3}, {4, 5}, {3, 12}, {7, 8}, {10, 20}, {20, 25}, {28, 29}, {6, 29}};
  const double Sh = 2.0 * bytes / 4;     // a CU1 touches a quarter of the state: S/2 moved
  for (CT ct : cts) {
    if (ct.hi >= nb) continue;
    const uint64_t nw = n / 4;
    const int p = ct.lo * 100 + ct.hi;
#define RUN_DB(U) rep("diag", "B U=" #U, p, timeit([&] { hipLaunchKernelGGL((k_diag_B<U>), dim3((unsigned)(nw / (256 * U))), dim3(256), 0, 0, psi, ct.lo, ct.hi); }, reps), Sh)
#define RUN_DT(U, W, R) { const uint64_t blocks = nw / (64ull * U * W); rep("diag", "T U=" #U " wpb=" #W " rot=" #R, p, timeit([&] { hipLaunchKernelGGL((k_diag_T<U, W, R>), dim3((unsigned)blocks), dim3(64 * W), 0, 0, psi, ct.lo, ct.hi, log2u(blocks)); }, reps), Sh); }
    RUN_DB(1); RUN_DB(2); RUN_DB(4);
    RUN_DT(4, 4, 0); RUN_DT(8, 4, 0); RUN_DT(16, 2, 0); RUN_DT(8, 4, 3); RUN_DT(16, 2, 3); RUN_DT(32, 1, 3); RUN_DT(16, 1, 6);
  }
  // big register tiles
  {
#define RUN_R(ROWS, R) { const uint64_t blocks = n / (64ull * ROWS); rep("rows", "rows=" #ROWS " 1 wave/blk rot=" #R, ROWS, timeit([&] { hipLaunchKernelGGL((k_rmw_rows<ROWS>), dim3((unsigned)blocks), dim3(64), 0, 0, psi, log2u(blocks), R); }, reps), S2); }
    RUN_R(16, 3); RUN_R(32, 3); RUN_R(64, 3); RUN_R(32, 0); RUN_R(64, 0);
  }
  CK(hipFree(psi));
  return 0;
}
