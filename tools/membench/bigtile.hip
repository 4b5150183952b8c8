// bigtile.hip -- VERDICT r03 #5: could a 30-qubit QFT run in TWO sweeps (15 dense target bits each)?  A sweep gives a dense
// gate to the bits of its tile, and a tile must be resident on chip while its gates run: 2^15 amplitudes x 16 B = 512 KiB
// = the whole vector register file of a CU (4 SIMDs x 512 registers x 64 lanes x 4 B), with no register left for a
// temporary -- so 15 bits are out on capacity alone, and 14 + 14 < 30.  What this bench measures is the other half of the
// argument: what the memory stream loses when ONE workgroup owns a CU's register file (tiles of 2^14 amplitudes: 8 waves x
// 128 VGPRs, nothing else resident, all loads -> barrier -> all stores) against k_sweep's shape (12 independent waves).
//   usage: bigtile NBITS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int WAVES, bool BARRIER> __global__ __launch_bounds__(WAVES * 64) void k_big(v2d *__restrict__ p, int rot, int blk_bits) {
  extern __shared__ v2d lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t bi = blockIdx.x;
  if (rot) bi = ((bi >> rot) | (bi << (blk_bits - rot))) & ((1ull << blk_bits) - 1);
  v2d *base = p + ((bi * WAVES + wave) << 11) + lane;
  v2d a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) a[k] = __builtin_nontemporal_load(base + 64 * k);
  if (BARRIER) __syncthreads();
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    v2d t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
    __builtin_nontemporal_store(t, base + 64 * k);
  }
}

template <int WAVES, bool BARRIER> static void run(v2d *p, uint64_t n, size_t lds_bytes, const char *what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t nblk = (n >> 11) / WAVES;
  int blk_bits = 0; while ((1ull << blk_bits) < nblk) ++blk_bits;
  CK(hipFuncSetAttribute((const void *)k_big<WAVES, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  auto launch = [&]() { hipLaunchKernelGGL((k_big<WAVES, BARRIER>), dim3((unsigned)nblk), dim3(WAVES * 64), lds_bytes, 0, p, 3, blk_bits); };
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("%-86s %7.3f ms %7.1f GB/s\n", what, ms, 2.0 * n * 16 / ms / 1e6);
}

int main(int argc, char **argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 30;
  const uint64_t n = 1ull << nb;
  v2d *p; CK(hipMalloc(&p, n * 16)); CK(hipMemset(p, 0, n * 16));
  for (int rep = 0; rep < 2; ++rep) {
    run<2, false>(p, n, 0, "k_sweep's shape: 2-wave workgroups, 12 waves / CU, waves independent");
    run<4, false>(p, n, 0, "4-wave workgroups, 12 waves / CU, waves independent");
    run<4, true>(p, n, 48 * 1024, "4-wave workgroups, 3 / CU, barrier between loads and stores (2^13-amplitude tiles)");
    run<8, true>(p, n, 64 * 1024, "8-wave workgroups, 2 / CU (64 KiB LDS each), barrier (2^14-amplitude tiles, 16 waves / CU)");
    run<8, true>(p, n, 128 * 1024, "8-wave workgroups, 1 / CU (128 KiB LDS), barrier: ONE 2^14-amplitude tile per CU");
    run<16, true>(p, n, 128 * 1024, "16-wave workgroups, 1 / CU, barrier: ONE 2^15-amplitude tile per CU (no temporaries left)");
  }
  return 0;
}
