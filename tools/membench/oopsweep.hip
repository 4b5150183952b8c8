// oopsweep.hip -- HBM rate of a register-tile sweep whose STORE goes to a second buffer with the
// tile's index bits moved to other positions (out-of-place "autosort" sweep), against the in-place
// sweep of the same tile.  A tile = lane bits {0,1,2} + 3 lane bits + 5 register bits + 1 wave bit
// (two waves per workgroup), as k_sweep<5> with one wave bit.
//   usage: oopsweep NBITS  SRC:DST[:i]  ...
//     flags after the second colon: i = in place, uN = N super-tiles per workgroup, rN = block rotation
//     SRC, DST = nine comma-separated bit positions (l3,l4,l5,r0..r4,w); every other index bit keeps
//     its relative order in the destination.  ":i" = in place (DST ignored, same buffer).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
struct Geom { int pos[9]; int sorted[9]; };

__device__ inline uint64_t expand(uint64_t j, const Geom &g) {
  for (int k = 0; k < 9; ++k) {
    const uint64_t low = (1ull << g.sorted[k]) - 1;
    j = ((j & ~low) << 1) | (j & low);
  }
  return j;
}

// U = super-tiles (2 waves each) per workgroup, consecutive unit indices; rot = block-index rotation
__global__ __launch_bounds__(512) void k_oop(const v2d *__restrict__ src, v2d *__restrict__ dst, Geom gs, Geom gd,
                                             int U, int rot, int blk_bits) {
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 1, sub = threadIdx.x >> 7;
  uint64_t ls = (uint64_t)(lane & 7), ld = ls;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    ls |= (uint64_t)((lane >> (3 + k)) & 1) << gs.pos[k];
    ld |= (uint64_t)((lane >> (3 + k)) & 1) << gd.pos[k];
  }
  uint64_t bi = blockIdx.x;
  if (rot) bi = ((bi >> rot) | (bi << (blk_bits - rot))) & ((1ull << blk_bits) - 1);
  const uint64_t w = bi * U + sub;
  const uint64_t js = expand(w << 3, gs) | ((uint64_t)wave << gs.pos[8]);
  const uint64_t jd = expand(w << 3, gd) | ((uint64_t)wave << gd.pos[8]);
  v2d a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << gs.pos[3 + b];
    a[k] = __builtin_nontemporal_load(&src[(js | o) + ls]);
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    uint64_t o = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) o |= (uint64_t)((k >> b) & 1) << gd.pos[3 + b];
    v2d t; t.x = a[k].x * 0.6 - a[k].y * 0.8; t.y = a[k].x * 0.8 + a[k].y * 0.6;
    __builtin_nontemporal_store(t, &dst[(jd | o) + ld]);
  }
}

static bool parse9(const char *s, Geom *g) {
  char buf[128]; strncpy(buf, s, 127); buf[127] = 0;
  int k = 0;
  for (char *t = strtok(buf, ","); t && k < 9; t = strtok(nullptr, ",")) g->pos[k++] = atoi(t);
  if (k != 9) return false;
  memcpy(g->sorted, g->pos, sizeof g->pos);
  std::sort(g->sorted, g->sorted + 9);
  return true;
}

int main(int argc, char **argv) {
  int nb = argc > 1 ? atoi(argv[1]) : 30;
  uint64_t n = 1ull << nb; size_t bytes = n * 16;
  v2d *p, *q;
  CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes));
  CK(hipMemset(p, 0, bytes)); CK(hipMemset(q, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint64_t nunits = n >> 12;
  for (int a = 2; a < argc; ++a) {
    char buf[256]; strncpy(buf, argv[a], 255); buf[255] = 0;
    char *c1 = strchr(buf, ':');
    if (!c1) { printf("bad spec %s\n", argv[a]); continue; }
    *c1 = 0;
    char *c2 = strchr(c1 + 1, ':');
    bool inplace = false;
    int U = 1, rot = 0;
    if (c2) {
      *c2 = 0;
      inplace = strchr(c2 + 1, 'i') != nullptr;
      if (const char *u = strchr(c2 + 1, 'u')) U = atoi(u + 1);
      if (const char *r = strchr(c2 + 1, 'r')) rot = atoi(r + 1);
    }
    Geom gs, gd;
    if (!parse9(buf, &gs) || !parse9(inplace && !c1[1] ? buf : c1 + 1, &gd)) { printf("bad spec %s\n", argv[a]); continue; }
    if (inplace) gd = gs;
    float best = 1e9f, sum = 0;
    const int reps = 6;
    for (int rep = 0; rep < reps; ++rep) {
      CK(hipEventRecord(e0));
      // ping-pong like the engine would: p -> q, q -> p
      const v2d *s = (rep & 1) ? q : p; v2d *d = inplace ? (v2d *)s : ((rep & 1) ? p : q);
      int blk_bits = 0;
      while ((1ull << blk_bits) < nunits / U) blk_bits++;
      k_oop<<<dim3((unsigned)(nunits / U)), dim3(128 * U)>>>(s, d, gs, gd, U, rot, blk_bits);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) { if (ms < best) best = ms; sum += ms; }
    }
    printf("%-60s best %.3f ms  avg %.3f ms  %.0f GB/s\n", argv[a], best, sum / (reps - 1), 2.0 * bytes / best / 1e6);
  }
  return 0;
}
