// mallbench.hip -- VERDICT r04 #1: can two DEPENDENT sweeps share one HBM round trip through the 256-MiB Infinity Cache?
//
// A QFT-30 is three sweeps of (read S, write S).  If sweep B could run over a block of the state while the block sweep A
// just stored is still on chip, the pair would cost one read and one write of HBM instead of two of each.  This bench
// moves bytes only (no arithmetic) in k_sweep's launch shape -- two-wave workgroups, 32 x 16 B per lane = 128 VGPRs of
// tile, three waves per SIMD:
//   phase A: tile = 4 096 contiguous amplitudes of X  ->  the same 64 KiB of Y           (a relayout sweep's store)
//   phase B: tile = 512 lines of 128 B of ONE block of Y, 2^(bb-9) amplitudes apart, in place (the next sweep's gather)
// B(block) may only start when all of A(block) is stored: a per-block counter in device memory.
//   mode 0  A over everything, then B over everything, two launches              (today: 2 HBM round trips)
//   mode 1  one launch, every workgroup does A(tile) -> waits for its block -> B(tile)        ("same workgroup")
//   mode 2  one launch, A(b) and B(b - lag) workgroups interleaved block by block             ("lagged")
//   mode 3  A alone;  mode 4  B alone                                                  (one round trip each)
// Cache-policy bits of the four access streams are template parameters (0 none, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0).
//   usage: mallbench NBITS BLOCKBITS MODE LAG XCDLOCAL LDA STA LDB STB [REPS]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int P> __device__ __forceinline__ void ld(v4f &d, const void *sb, unsigned vo) {
  if constexpr (P == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 1) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 2) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 3) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 4) asm volatile("global_load_dwordx4 %0, %1, %2 sc1 nt" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 5) asm volatile("global_load_dwordx4 %0, %1, %2 sc0" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
  if constexpr (P == 6) asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1 nt" : "=v"(d) : "v"(vo), "s"(sb) : "memory");
}
template <int P> __device__ __forceinline__ void st(void *sb, unsigned vo, const v4f &d) {
  if constexpr (P == 0) asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 5) asm volatile("global_store_dwordx4 %0, %1, %2 sc0" : : "v"(vo), "v"(d), "s"(sb) : "memory");
  if constexpr (P == 6) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" : : "v"(vo), "v"(d), "s"(sb) : "memory");
}
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr unsigned kErrSlot = 1u << 20;   // counter slot that collects give-ups
struct Geo {
  int bb;          // log2 amplitudes per block
  int nblk;        // blocks
  int lag;         // mode 2: B(b) is issued behind A(b + lag)
  int xcdlocal;    // 1: the workgroups of a block all run on one XCD (workgroup id mod 8)
  unsigned grid;   // workgroups launched
};

// tile copy A: X -> Y, contiguous 64 KiB per workgroup
template <int LD, int ST> __device__ __forceinline__ void phase_a(const v4f *X, v4f *Y, uint64_t tile) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t base = (tile << 12) + ((uint64_t)wave << 11);
  const unsigned vo = lane * 16;
  v4f a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) ld<LD>(a[k], X + base + 64 * k, vo);
  wait_vm();
#pragma unroll
  for (int k = 0; k < 32; ++k) st<ST>(Y + base + 64 * k, vo, a[k]);
}
// tile RMW B: 512 lines of the block, 2^(bb-9) amplitudes apart, in place
template <int LD, int ST> __device__ __forceinline__ void phase_b(v4f *Y, uint64_t block, uint64_t t, int bb) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sh = bb - 9;                     // <= 12: the lane part stays below 2^32 bytes
  v4f *base = Y + (block << bb) + (t << 3);
  const unsigned vo = (unsigned)((((uint64_t)(lane >> 3) << sh) + (lane & 7)) * 16);
  v4f a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) ld<LD>(a[k], base + ((uint64_t)((wave << 8) | (k << 3)) << sh), vo);
  wait_vm();
#pragma unroll
  for (int k = 0; k < 32; ++k) st<ST>(base + ((uint64_t)((wave << 8) | (k << 3)) << sh), vo, a[k]);
}

__device__ __forceinline__ unsigned virt_wg(const Geo &g) {
  const unsigned w = blockIdx.x;
  if (!g.xcdlocal) return w;
  return (w & 7) * (g.grid >> 3) + (w >> 3);
}
__device__ __forceinline__ void signal_block(unsigned *cnt, uint64_t block) {
  wait_vm();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt + block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_block(unsigned *cnt, uint64_t block, unsigned target) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;                                    // bounded: a bench must not hang the box
    while (__hip_atomic_load(cnt + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 21)) { __hip_atomic_fetch_add(cnt + kErrSlot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}

template <int LDA, int STA> __global__ __launch_bounds__(128) void k_a(const v4f *X, v4f *Y, Geo g) {
  phase_a<LDA, STA>(X, Y, virt_wg(g));
}
template <int LDB, int STB> __global__ __launch_bounds__(128) void k_b(v4f *Y, Geo g) {
  const unsigned v = virt_wg(g);
  const int tb = g.bb - 12;
  phase_b<LDB, STB>(Y, v >> tb, v & ((1u << tb) - 1), g.bb);
}
template <int LDA, int STA, int LDB, int STB> __global__ __launch_bounds__(128) void k_same(const v4f *X, v4f *Y, unsigned *cnt, Geo g) {
  const unsigned v = virt_wg(g);
  const int tb = g.bb - 12;
  const uint64_t block = v >> tb;
  phase_a<LDA, STA>(X, Y, v);
  signal_block(cnt, block);
  wait_block(cnt, block, 1u << tb);
  phase_b<LDB, STB>(Y, block, v & ((1u << tb) - 1), g.bb);
}
template <int LDA, int STA, int LDB, int STB> __global__ __launch_bounds__(128) void k_lag(const v4f *X, v4f *Y, unsigned *cnt, Geo g) {
  const unsigned v = virt_wg(g);
  const int tb = g.bb - 12;
  const unsigned t = v & ((1u << tb) - 1);
  const int e = (int)(v >> tb);
  int phase, block;
  if (e < g.lag) { phase = 0; block = e; }
  else {
    const int q = e - g.lag, pairs = g.nblk - g.lag;
    if (q < 2 * pairs) { phase = (q & 1) ? 0 : 1; block = (q & 1) ? g.lag + (q >> 1) : (q >> 1); }
    else { phase = 1; block = q - pairs; }
  }
  if (phase == 0) {
    phase_a<LDA, STA>(X, Y, ((uint64_t)block << tb) + t);
    signal_block(cnt, block);
  } else {
    wait_block(cnt, block, 1u << tb);
    phase_b<LDB, STB>(Y, block, t, g.bb);
  }
}

struct Args { int nb, bb, mode, lag, xcd, reps; };
static size_t g_lds = 0;   // MALL_LDS=bytes of dynamic LDS per workgroup: caps the workgroups per CU (160 KiB / bytes)
static v4f *X, *Y; static unsigned *cnt;

template <int LDA, int STA, int LDB, int STB> static float run(const Args &a) {
  const uint64_t n = 1ull << a.nb;
  Geo g; g.bb = a.bb; g.nblk = (int)(n >> a.bb); g.lag = a.lag < g.nblk ? a.lag : g.nblk; g.xcdlocal = a.xcd;
  const unsigned tiles = (unsigned)(n >> 12);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto once = [&]() {
    if (a.mode == 0) {
      g.grid = tiles;
      hipLaunchKernelGGL((k_a<LDA, STA>), dim3(tiles), dim3(128), g_lds, 0, X, Y, g);
      hipLaunchKernelGGL((k_b<LDB, STB>), dim3(tiles), dim3(128), g_lds, 0, Y, g);
    } else if (a.mode == 1) {
      g.grid = tiles;
      CK(hipMemsetAsync(cnt, 0, g.nblk * 4, 0));
      hipLaunchKernelGGL((k_same<LDA, STA, LDB, STB>), dim3(tiles), dim3(128), g_lds, 0, X, Y, cnt, g);
    } else if (a.mode == 2) {
      g.grid = 2 * tiles;
      CK(hipMemsetAsync(cnt, 0, g.nblk * 4, 0));
      hipLaunchKernelGGL((k_lag<LDA, STA, LDB, STB>), dim3(2 * tiles), dim3(128), g_lds, 0, X, Y, cnt, g);
    } else if (a.mode == 3) {
      g.grid = tiles;
      hipLaunchKernelGGL((k_a<LDA, STA>), dim3(tiles), dim3(128), g_lds, 0, X, Y, g);
    } else {
      g.grid = tiles;
      hipLaunchKernelGGL((k_b<LDB, STB>), dim3(tiles), dim3(128), g_lds, 0, Y, g);
    }
  };
  if (g_lds > 48 * 1024) {
    CK(hipFuncSetAttribute((const void *)k_a<LDA, STA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
    CK(hipFuncSetAttribute((const void *)k_b<LDB, STB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
    CK(hipFuncSetAttribute((const void *)k_same<LDA, STA, LDB, STB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
    CK(hipFuncSetAttribute((const void *)k_lag<LDA, STA, LDB, STB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_lds));
  }
  once(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < a.reps; ++r) once();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / a.reps;
}

// verification of the fused modes: Y must equal X afterwards (both phases copy)
static int verify(int nb) {
  const uint64_t n = 1ull << nb;
  std::vector<unsigned> hx(1 << 16), hy(1 << 16);
  int bad = 0;
  for (int w = 0; w < 64; ++w) {
    const uint64_t off = ((n / 64) * w + 4096ull * 13 * w) & (n - 1) & ~4095ull;
    CK(hipMemcpy(hx.data(), X + off, hx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), Y + off, hy.size() * 4, hipMemcpyDeviceToHost));
    bad += memcmp(hx.data(), hy.data(), hx.size() * 4) != 0;
  }
  return bad;
}
__global__ void k_fill(unsigned *p, uint64_t nwords, unsigned salt) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x)
    p[i] = (unsigned)(i * 2654435761u) ^ salt;
}

typedef float (*runfn)(const Args &);
template <int LDA, int STA, int LDB, int STB> static runfn pick() { return run<LDA, STA, LDB, STB>; }

int main(int argc, char **argv) {
  if (argc < 10) { printf("usage: mallbench NBITS BLOCKBITS MODE LAG XCDLOCAL LDA STA LDB STB [REPS]\n"); return 2; }
  Args a{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), argc > 10 ? atoi(argv[10]) : 5};
  const int lda = atoi(argv[6]), sta = atoi(argv[7]), ldb = atoi(argv[8]), stb = atoi(argv[9]);
  const uint64_t n = 1ull << a.nb;
  CK(hipMalloc(&X, n * 16)); CK(hipMalloc(&Y, n * 16)); CK(hipMalloc(&cnt, (kErrSlot + 1) * 4)); CK(hipMemset(cnt, 0, (kErrSlot + 1) * 4));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (unsigned *)X, n * 4, 0x1234567u);
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (unsigned *)Y, n * 4, 0x89abcdeu);
  CK(hipDeviceSynchronize());
  // the combinations that make sense: loads of A stream (nt / none), stores of A must reach the memory side when B may run
  // on another XCD (sc1 = agent scope), loads of B must not be served from a stale L2 line (sc1), stores of B stream.
  runfn f = nullptr;
#define COMBO(A_, B_, C_, D_) if (lda == A_ && sta == B_ && ldb == C_ && stb == D_) f = pick<A_, B_, C_, D_>();
  COMBO(1, 1, 1, 1) COMBO(1, 2, 2, 1) COMBO(1, 3, 3, 1) COMBO(1, 4, 4, 1) COMBO(1, 2, 4, 1) COMBO(1, 4, 2, 1)
  COMBO(1, 0, 0, 1) COMBO(0, 0, 0, 0) COMBO(1, 2, 2, 0) COMBO(0, 2, 2, 0) COMBO(0, 2, 2, 1) COMBO(1, 2, 2, 4)
  COMBO(1, 5, 5, 1) COMBO(1, 0, 2, 1) COMBO(1, 2, 0, 1) COMBO(1, 3, 2, 1) COMBO(1, 2, 2, 2) COMBO(1, 2, 3, 1)
  COMBO(3, 2, 2, 1) COMBO(6, 2, 2, 1) COMBO(1, 2, 2, 6) COMBO(1, 6, 6, 1) COMBO(6, 2, 2, 6)
  if (!f) { printf("policy combination not compiled in\n"); return 2; }
  if (getenv("MALL_LDS")) g_lds = (size_t)atol(getenv("MALL_LDS"));
  const float ms = f(a);
  int bad = (a.mode == 4) ? 0 : verify(a.nb);
  unsigned gaveup = 0; CK(hipMemcpy(&gaveup, cnt + kErrSlot, 4, hipMemcpyDeviceToHost));
  if (gaveup) { printf("GAVE UP WAITING %u times\n", gaveup); bad = 1; }
  static const char *pol[] = {"-", "nt", "sc1", "sc0sc1", "sc1nt", "sc0", "sc0sc1nt"};
  printf("lds=%zu n=%d bb=%d mode=%d lag=%d xcd=%d ldA=%s stA=%s ldB=%s stB=%s : %8.3f ms  (%.2f x S/5.5ms-pass)  %s\n", g_lds, a.nb, a.bb, a.mode,
         a.lag, a.xcd, pol[lda], pol[sta], pol[ldb], pol[stb], ms, ms / (2.0 * n * 16 / 6.25e9), bad ? "MISMATCH" : "ok");
  return bad != 0;
}
