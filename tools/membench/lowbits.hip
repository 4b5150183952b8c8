// lowbits.hip -- VERDICT r04 #4: a controlled phase (CU1 / CZ) whose control or target is one of the index bits 0..2, i.e.
// INSIDE the 128-byte line.  Algorithmically it changes the amplitudes with both bits set: N/4 amplitudes, S/2 bytes read +
// written.  Bits 0 and 1 share a 64-byte half line with their partners, bit 2 selects one half of the line.  Which way of
// touching them is fastest on MI355X (2^nb complex128, in place, k_diag_tile's shape: 8 items per thread, four-wave blocks,
// block index rotated by 3)?
//   pred   enumerate the amplitudes with HI set (N/2), multiply under a per-lane predicate on LO, rewrite whole lines
//          (what the engine does for LO = 0, 1)
//   ins    enumerate the amplitudes with HI and LO set (N/4): 16- / 32- / 64-byte pieces of every line with HI set
//          (what the engine does for LO = 2)
//   rdall  load like pred (whole lines, every sector fetched once), store only the lanes with LO set (EXEC-masked stores)
//   ins64  LO = 0, 1 only: enumerate 64-byte half lines (bit 2 inserted as 0 / 1 in two streams is the same as pred); here:
//          pred restricted to ... (not applicable -- every half line contains amplitudes with LO set)
// usage: lowbits NBITS [REPS]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
__device__ __forceinline__ uint64_t ins1(uint64_t j, int p) {      // insert a ONE at position p
  const uint64_t low = (1ull << p) - 1ull;
  return ((j & ~low) << 1) | (j & low) | (1ull << p);
}
__device__ __forceinline__ uint64_t rot3(uint64_t bi, int blk_bits) {
  return blk_bits > 3 ? (((bi >> 3) | (bi << (blk_bits - 3))) & ((1ull << blk_bits) - 1)) : bi;
}

// MODE 0 pred, 1 ins, 2 rdall
template <int MODE, int U, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_cphase(double2 *__restrict__ psi, int lo, int hi, int blk_bits) {
  const uint64_t bi = rot3(blockIdx.x, blk_bits);
  const uint64_t base = ((bi * WPB + (threadIdx.x >> 6)) * U) * 64ull + (threadIdx.x & 63);
  const double fr = 0.6, fi = 0.8;
  double2 a[U];
  uint64_t idx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    uint64_t j = base + 64ull * u;
    if (MODE == 1) j = ins1(j, lo);          // (lo < hi: insert the lower position first, hi then counts in the widened index)
    idx[u] = ins1(j, hi);
  }
#pragma unroll
  for (int u = 0; u < U; ++u) a[u] = ldnt(&psi[idx[u]]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const bool on = MODE == 1 || ((idx[u] >> lo) & 1ull);
    double2 t = a[u];
    if (on) { t.x = fr * a[u].x - fi * a[u].y; t.y = fr * a[u].y + fi * a[u].x; }
    if (MODE != 2 || on) stnt(&psi[idx[u]], t);
  }
}

template <int MODE> static float run(double2 *p, int nb, int lo, int hi, int reps) {
  const uint64_t nwork = 1ull << (nb - (MODE == 1 ? 2 : 1));
  const uint64_t blocks = nwork / (64 * 8 * 4);
  int blk_bits = 0; while ((1ull << blk_bits) < blocks) ++blk_bits;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { hipLaunchKernelGGL((k_cphase<MODE, 8, 4>), dim3((unsigned)blocks), dim3(256), 0, 0, p, lo, hi, blk_bits); };
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 30;
  const int reps = argc > 2 ? atoi(argv[2]) : 8;
  const uint64_t n = 1ull << nb;
  double2 *p; CK(hipMalloc(&p, n * 16)); CK(hipMemset(p, 0, n * 16));
  const double alg = (double)(n / 4) * 16 * 2;         // bytes whose value changes, read + written
  printf("2^%d complex128; algorithmic bytes of a controlled phase = S/2 = %.2f GB; times in ms, (GB/s algorithmic = frac of 8 TB/s)\n", nb, alg / 1e9);
  printf("%-10s %22s %22s %22s\n", "lo,hi", "pred (whole lines)", "ins (pieces)", "rdall (masked stores)");
  for (int round = 0; round < 2; ++round)
    for (int hi : {12, 25})
      for (int lo : {0, 1, 2, 3, 5}) {
        const float a = run<0>(p, nb, lo, hi, reps), b = run<1>(p, nb, lo, hi, reps), c = run<2>(p, nb, lo, hi, reps);
        printf("%d,%-8d %8.3f (%5.0f = %.2f) %8.3f (%5.0f = %.2f) %8.3f (%5.0f = %.2f)\n", lo, hi, a, alg / a / 1e6, alg / a / 8e9, b,
               alg / b / 1e6, alg / b / 8e9, c, alg / c / 1e6, alg / c / 8e9);
      }
  // both bits inside the line: pred on both (N amplitudes enumerated) vs insertion of the upper one
  return 0;
}
