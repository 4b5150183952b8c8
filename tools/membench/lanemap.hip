// lanemap.hip -- does the HBM rate of a register-tile sweep depend on WHICH lanes of a wave hold the amplitudes of one
// 128-byte line?  k_sweep keeps index bits 0,1,2 on lane bits 0,1,2 (lanes 0-7 = one line); a dense gate on those qubits
// then needs DPP partner fetches (192-320 VALU instructions), while lane bits 4 and 5 trade places with a register bit by
// v_permlane{16,32}_swap (64).  If the lines may sit on other lane bits at the same rate, the expensive lane seats can go to
// index bits the sweep never targets.
//   tile = 6 lane bits (index bits 0,1,2 + three "h" bits) + 5 register bits; one wave per tile, four tiles per workgroup.
//   variant = which lane bit carries i0,i1,i2,h0,h1,h2 (six digits), e.g. 012345 = today's map, 045123: i1,i2 on lanes 4,5
//   mode i = in place, o = gather -> contiguous block of a second buffer (lane l stores to the position its bit class has)
//   usage: lanemap NBITS H0 R0 variant:mode ...     (h bits = H0..H0+2, register bits = R0..R0+4)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
struct Geom { int sorted[11]; int reg[5]; uint32_t lsrc[64]; uint32_t ldst[64]; int inplace; };

__global__ __launch_bounds__(256) void k_lm(const v2d *__restrict__ src, v2d *__restrict__ dst, Geom g, int rot, int blk_bits) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t bi = blockIdx.x;
  if (rot) bi = ((bi >> rot) | (bi << (blk_bits - rot))) & ((1ull << blk_bits) - 1);
  const uint64_t w = bi * 4 + wave;
  uint64_t j = w;
#pragma unroll
  for (int k = 0; k < 11; ++k) { const uint64_t low = (1ull << g.sorted[k]) - 1; j = ((j & ~low) << 1) | (j & low); }
  const uint64_t jd = g.inplace ? j : (w << 11);
  const uint64_t ls = g.lsrc[lane], ld = g.inplace ? ls : g.ldst[lane];
  v2d a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << g.reg[b];
    a[k] = __builtin_nontemporal_load(&src[(j | o) + ls]);
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << (g.inplace ? g.reg[b] : 6 + b);
    v2d t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
    __builtin_nontemporal_store(t, &dst[(jd | o) + ld]);
  }
}

int main(int argc, char **argv) {
  if (argc < 5) { printf("usage: lanemap NBITS H0 R0 variant:mode ...\n"); return 1; }
  const int nb = atoi(argv[1]), h0 = atoi(argv[2]), r0 = atoi(argv[3]);
  const uint64_t n = 1ull << nb; const size_t bytes = n * 16;
  v2d *p, *q;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes));
  CK(hipMemset(p, 0, bytes)); CK(hipMemset(q, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t ntiles = n >> 11;
  int blk_bits = 0; while ((1ull << blk_bits) < ntiles / 4) ++blk_bits;
  for (int rep = 0; rep < 2; ++rep)
  for (int a = 4; a < argc; ++a) {
    const char *v = argv[a];
    Geom g{};
    const int srcbit[6] = {0, 1, 2, h0, h0 + 1, h0 + 2};     // i0 i1 i2 h0 h1 h2
    int lanebit[6];
    for (int k = 0; k < 6; ++k) lanebit[k] = v[k] - '0';
    g.inplace = v[7] == 'i';
    for (int l = 0; l < 64; ++l) {
      uint32_t s = 0, d = 0;
      for (int k = 0; k < 6; ++k) if ((l >> lanebit[k]) & 1) { s |= 1u << srcbit[k]; d |= 1u << k; }
      g.lsrc[l] = s; g.ldst[l] = d;
    }
    for (int b = 0; b < 5; ++b) g.reg[b] = r0 + b;
    int all[11] = {0, 1, 2, h0, h0 + 1, h0 + 2, r0, r0 + 1, r0 + 2, r0 + 3, r0 + 4};
    std::sort(all, all + 11); memcpy(g.sorted, all, sizeof all);
    const int rot = g.inplace ? 3 : 6;
    auto launch = [&]() { hipLaunchKernelGGL(k_lm, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, p, g.inplace ? p : q, g, rot, blk_bits); };
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("h %2d..%2d reg %2d..%2d lanes(i0 i1 i2 h0 h1 h2)=%.6s %s : %7.3f ms %7.1f GB/s\n", h0, h0 + 2, r0, r0 + 4, v,
           g.inplace ? "in place " : "gather->contiguous", ms, 2.0 * bytes / ms / 1e6);
  }
  return 0;
}
