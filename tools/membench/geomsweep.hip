// geomsweep.hip -- HBM rate of a read-modify-write register-tile sweep as a function of the
// tile geometry alone: lane bits {0,1,2} + three arbitrary "high" lane bits + five arbitrary
// register bits (the k_sweep<5> access pattern with a trivial body).
//   usage: geomsweep NBITS  l3,l4,l5,r0,r1,r2,r3,r4  [more geometries ...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
struct Geom { int pos[8]; int sorted[10]; int nins; int rot; int nblk_bits; int xs, xd, xw; int nwave; int wpos[2]; };

__global__ __launch_bounds__(256) void k_geom(v2d *__restrict__ p, uint64_t ntiles, Geom g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t loff = (uint64_t)(lane & 7);
#pragma unroll
  for (int k = 0; k < 3; ++k) loff |= (uint64_t)((lane >> (3 + k)) & 1) << g.pos[k];
  // optional rotation of the block index: the fastest-varying block bits land on the HIGHEST free bits
  uint64_t bi = blockIdx.x;
  if (g.rot) {
    const uint64_t m = (1ull << g.nblk_bits) - 1;
    bi = ((bi >> g.rot) | (bi << (g.nblk_bits - g.rot))) & m;
  }
  const uint64_t w = g.nwave ? bi : bi * 4 + wave;
  if (w >= (g.nwave ? (ntiles >> g.nwave) : ntiles)) return;
  uint64_t j = w << 3;
  for (int k = 0; k < g.nins; ++k) {           // insert a zero at each tile bit, ascending
    const uint64_t low = (1ull << g.sorted[k]) - 1;
    j = ((j & ~low) << 1) | (j & low);
  }
  for (int k = 0; k < g.nwave; ++k) j |= (uint64_t)((wave >> k) & 1) << g.wpos[k];
  v2d a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << g.pos[3 + b];
    uint64_t ix = (j | o) + loff;
    if (g.xw) ix ^= ((ix >> g.xs) & ((1ull << g.xw) - 1)) << g.xd;
    a[k] = __builtin_nontemporal_load(&p[ix]);
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << g.pos[3 + b];
    v2d t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
    uint64_t ix = (j | o) + loff;
    if (g.xw) ix ^= ((ix >> g.xs) & ((1ull << g.xw) - 1)) << g.xd;
    __builtin_nontemporal_store(t, &p[ix]);
  }
}

int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 30;
  uint64_t n = 1ull << nb; size_t bytes = n * 16;
  v2d *p;
  if (getenv("GEOM_ALIGN")) {   // over-allocate and align the base to the state size
    char *raw; CK(hipMalloc(&raw, 2 * bytes));
    uintptr_t a = ((uintptr_t)raw + bytes - 1) / bytes * bytes;
    p = (v2d *)a;
    printf("raw %p aligned %p\n", (void *)raw, (void *)p);
  } else {
    CK(hipMalloc(&p, bytes));
    printf("base %p\n", (void *)p);
  }
  CK(hipMemset(p, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  uint64_t ntiles = n >> 11;
  for (int a = 2; a < argc; ++a) {
    Geom g; int k = 0;
    char buf[256]; strncpy(buf, argv[a], 255); buf[255] = 0;
    int all[10];
    { char *cut = strpbrk(buf, ":^"); if (cut) *cut = 0; }
    for (char *t = strtok(buf, ","); t && k < 10; t = strtok(nullptr, ",")) all[k++] = atoi(t);
    g.nwave = k > 8 ? k - 8 : 0;
    for (int q = 0; q < 8 && q < k; ++q) g.pos[q] = all[q];
    for (int q = 0; q < g.nwave; ++q) g.wpos[q] = all[8 + q];
    g.nins = 8 + g.nwave;
    if (k > 8) k = 8;
    g.rot = 0; g.nblk_bits = g.nwave ? nb - 11 - g.nwave : nb - 13;
    g.xs = g.xd = g.xw = 0;
    if (k == 8) { char *c = strchr(argv[a], ':'); if (c) g.rot = atoi(c + 1); }
    { char *c = strchr(argv[a], '^'); if (c) sscanf(c + 1, "%d,%d,%d", &g.xs, &g.xd, &g.xw); }
    if (k != 8) { printf("bad geometry %s\n", argv[a]); continue; }
    memcpy(g.sorted, g.pos, sizeof(g.pos)); for (int q = 0; q < g.nwave; ++q) g.sorted[8 + q] = g.wpos[q]; std::sort(g.sorted, g.sorted + g.nins);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0));
      k_geom<<<dim3((unsigned)(g.nwave ? (ntiles >> g.nwave) : (ntiles + 3) / 4)), dim3(g.nwave ? (64 << g.nwave) : 256)>>>(p, ntiles, g);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
    }
    printf("%s  %.3f ms  %.0f GB/s\n", argv[a], best, 2.0 * bytes / best / 1e6);
  }
  return 0;
}
