// mfmapower.hip -- round 6, VERDICT r05 #1(a): price the FP64 matrix pipe on the socket's power limit and try ONE fused complex
// 4x4 unitary on two lane bits as four v_mfma_f64_4x4x4_4b_f64 per slot (the data stays where it lies across the lanes, the
// matrix is a per-lane constant), beside the DPP lane butterflies it would replace.
//   usage: mfmapower layout            lane maps of v_mfma_f64_4x4x4_4b_f64 found by one-hot probing, and the self-check of the
//                                      fused group against a host 4x4 complex product
//          mfmapower MODE SECONDS      MODE: mfma4 mfma16 fma add dppbf grp grpmix salu smem   (tools/probes/r06_mfmapower.py samples rocm-smi)
// Occupancy as k_sweep: 256-thread blocks, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- layout probe (one wave)
__global__ void k_probe(double *out) {            // out[la][lb][lane]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}

// y = U x over the lanes, x complex per lane; data operand = B (DATA_B) or A
template <bool DATA_B>
__global__ void k_group_once(const double *tr, const double *ti, const double *xr, const double *xi, double *yr, double *yi) {
  const int lane = threadIdx.x;
  const double ur = tr[lane], ui = ti[lane], ar = xr[lane], ai = xi[lane];
  double r, i;
  if (DATA_B) {
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(ur, ar, 0.0, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(-ui, ai, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f64_4x4x4f64(ur, ai, 0.0, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f64_4x4x4f64(ui, ar, i, 0, 0, 0);
  } else {
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, ur, 0.0, 0, 0, 0);
    r = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, -ui, r, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, ur, 0.0, 0, 0, 0);
    i = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, ui, i, 0, 0, 0);
  }
  yr[lane] = r;
  yi[lane] = i;
}

static int layout_and_check() {
  double *out; CK(hipMalloc(&out, sizeof(double) * 64 * 64 * 64));
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out);
  CK(hipDeviceSynchronize());
  std::vector<double> h(64 * 64 * 64);
  CK(hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost));
  // R[la][lb] = set of ld with D != 0
  static int R[64][64];
  int multi = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      R[la][lb] = -1;
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.0) { if (R[la][lb] >= 0) ++multi; R[la][lb] = l; }
    }
  printf("v_mfma_f64_4x4x4_4b_f64: D lane <- (A lane, B lane) products  [%d pairs hit more than one D lane]\n", multi);
  for (int ld = 0; ld < 64; ++ld) {
    printf("  D%02d <-", ld);
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb)
        if (R[la][lb] == ld) printf(" A%02d*B%02d", la, lb);
    printf("\n");
  }
  int rc = 0;
  for (int variant = 0; variant < 2; ++variant) {           // 0: data in B, 1: data in A
    // sources of every D lane on the data side; the other side's lane is the coefficient seat
    int diffmask = 0, ok = 1;
    static int src[64][4], seat[64][4], ns[64];
    for (int ld = 0; ld < 64; ++ld) ns[ld] = 0;
    for (int la = 0; la < 64; ++la)
      for (int lb = 0; lb < 64; ++lb) {
        const int ld = R[la][lb];
        if (ld < 0) continue;
        const int s = variant == 0 ? lb : la, c = variant == 0 ? la : lb;
        if (ns[ld] < 4) { src[ld][ns[ld]] = s; seat[ld][ns[ld]] = c; }
        ++ns[ld];
        diffmask |= s ^ ld;
      }
    for (int ld = 0; ld < 64; ++ld) if (ns[ld] != 4) ok = 0;
    printf("data in %c: every D lane has 4 sources: %s; source lane ^ D lane stays inside lane-bit mask 0x%02x (%d bits)\n",
           variant == 0 ? 'B' : 'A', ok ? "yes" : "NO", diffmask, __builtin_popcount(diffmask));
    if (!ok || __builtin_popcount(diffmask) != 2) continue;
    const int p = __builtin_ctz(diffmask), q = 31 - __builtin_clz(diffmask);
    auto sub = [&](int lane) { return ((lane >> p) & 1) | (((lane >> q) & 1) << 1); };
    // random complex 4x4 (need not be unitary for the check)
    std::complex<double> U[4][4];
    srand(7);
    for (auto &row : U) for (auto &u : row) u = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
    std::vector<double> tr(64, 1e300), ti(64, 1e300), xr(64), xi(64);
    int consistent = 1;
    for (int ld = 0; ld < 64; ++ld)
      for (int s = 0; s < 4; ++s) {
        const std::complex<double> u = U[sub(ld)][sub(src[ld][s])];
        const int c = seat[ld][s];
        if (tr[c] != 1e300 && (tr[c] != u.real() || ti[c] != u.imag())) consistent = 0;
        tr[c] = u.real(); ti[c] = u.imag();
      }
    printf("  a 4x4 matrix on lane bits (%d,%d) as a per-lane constant: %s\n", p, q, consistent ? "consistent (one matrix for all lanes)" : "INCONSISTENT");
    if (!consistent) continue;
    for (int l = 0; l < 64; ++l) { xr[l] = rand() / (double)RAND_MAX - 0.5; xi[l] = rand() / (double)RAND_MAX - 0.5; }
    double *d; CK(hipMalloc(&d, 6 * 64 * sizeof(double)));
    CK(hipMemcpy(d, tr.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(d + 64, ti.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(d + 128, xr.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(d + 192, xi.data(), 512, hipMemcpyHostToDevice));
    if (variant == 0) hipLaunchKernelGGL(k_group_once<true>, dim3(1), dim3(64), 0, 0, d, d + 64, d + 128, d + 192, d + 256, d + 320);
    else hipLaunchKernelGGL(k_group_once<false>, dim3(1), dim3(64), 0, 0, d, d + 64, d + 128, d + 192, d + 256, d + 320);
    CK(hipDeviceSynchronize());
    std::vector<double> yr(64), yi(64);
    CK(hipMemcpy(yr.data(), d + 256, 512, hipMemcpyDeviceToHost));
    CK(hipMemcpy(yi.data(), d + 320, 512, hipMemcpyDeviceToHost));
    double err = 0;
    for (int l = 0; l < 64; ++l) {
      std::complex<double> want = 0;
      for (int c = 0; c < 4; ++c) {
        const int sl = (l & ~diffmask) | ((c & 1) << p) | ((c >> 1) << q);
        want += U[sub(l)][c] * std::complex<double>(xr[sl], xi[sl]);
      }
      err = fmax(err, std::abs(want - std::complex<double>(yr[l], yi[l])));
    }
    printf("  fused complex 4x4 on lane bits (%d,%d), 4 MFMAs, data in %c: max|mfma - host| = %.3e  %s\n", p, q, variant == 0 ? 'B' : 'A', err,
           err < 1e-14 ? "OK" : "MISMATCH");
    if (err >= 1e-14) rc = 1;
    CK(hipFree(d));
  }
  CK(hipFree(out));
  return rc;
}

// ---------------------------------------------------------------- power / rate loops
#define REP8(s) s s s s s s s s
// MODE 0 mfma4 (8 independent accumulators), 1 mfma16, 2 fma, 3 add, 4 dppbf (a DPP lane butterfly on 32 slots as the island
// emits it: 4 b32 DPP moves + 2 f64 adds per slot), 5 grp (128 MFMAs: the fused group over a 32-slot complex128 tile),
// 6 grpmix (grp with 256 independent v_add_f64 interleaved: does the matrix pipe run beside the VALU?)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k(double *out, int iters) {
  const double c = 0.99999991234, d = 1.2345678e-7;
  if (MODE <= 3) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 1.0 + 1e-3 * (threadIdx.x * 8 + j) + 1e-9 * blockIdx.x;
    double4_t q[2];
    q[0] = double4_t{a[0], a[1], a[2], a[3]};
    q[1] = double4_t{a[4], a[5], a[6], a[7]};
    const double ua = 0.25 + 1e-3 * threadIdx.x, ub = 0.5 - 1e-3 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
      if (MODE == 0) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(a[j]) : "v"(ua), "v"(ub));
      }
      if (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(q[j]) : "v"(ua), "v"(ub));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 2) asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a[j]) : "v"(c), "v"(d));
        if (MODE == 3) asm volatile(REP8("v_add_f64 %0, %0, %1\n\t") : "+v"(a[j]) : "v"(d));
      }
    }
    double s = q[0].x + q[0].y + q[0].z + q[0].w + q[1].x + q[1].y + q[1].z + q[1].w;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    if (s == 1.2345) out[0] = s;
  } else {
    double re[16], im[16], re2[16], im2[16];            // 16 slots twice per iteration = the 32 slots of a tile (the compiler spills a 32-slot copy)
#pragma unroll
    for (int j = 0; j < 16; ++j) { re[j] = 1e-3 * (threadIdx.x + j); im[j] = 1e-3 * (threadIdx.x - j) + 1e-9 * blockIdx.x; }
    // (H x H) e^{i theta}: symmetric, unitary -- the data keep their magnitude whichever of A's lane bits are i and k
    const int hs = __builtin_popcount((threadIdx.x & 3) & ((threadIdx.x >> 2) & 3)) & 1;
    const double ur = (hs ? -0.5 : 0.5) * 0.8, ui = (hs ? -0.5 : 0.5) * 0.6, sg = (threadIdx.x & 1) ? -1.0 : 1.0;
    double e0 = 1e-3 * threadIdx.x, e1 = 2e-3 * threadIdx.x;
    for (int i = 0; i < 2 * iters; ++i) {
      if (MODE == 4) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          double pr, pi;
          asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                       : "=&v"(((uint32_t *)&pr)[0]), "=&v"(((uint32_t *)&pr)[1]) : "v"(((uint32_t *)&re[j])[0]), "v"(((uint32_t *)&re[j])[1]));
          asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                       "v_mov_b32_dpp %1, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                       : "=&v"(((uint32_t *)&pi)[0]), "=&v"(((uint32_t *)&pi)[1]) : "v"(((uint32_t *)&im[j])[0]), "v"(((uint32_t *)&im[j])[1]));
          asm volatile("v_fma_f64 %0, %0, %2, %1" : "+v"(re[j]) : "v"(pr), "v"(sg));
          asm volatile("v_fma_f64 %0, %0, %2, %1" : "+v"(im[j]) : "v"(pi), "v"(sg));
        }
      } else {
        // ping-pong between two register sets: no copies in the loop
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(re2[j]) : "v"(ur), "v"(re[j]));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(im2[j]) : "v"(ur), "v"(im[j]));
          if (MODE == 6) asm volatile(REP8("v_add_f64 %0, %0, %1\n\t") : "+v"(e0) : "v"(d));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(re2[j]) : "v"(ui), "v"(im[j]));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(im2[j]) : "v"(ui), "v"(re[j]));
        }
        ++i;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(re[j]) : "v"(ur), "v"(re2[j]));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, 0" : "=&v"(im[j]) : "v"(ur), "v"(im2[j]));
          if (MODE == 6) asm volatile(REP8("v_add_f64 %0, %0, %1\n\t") : "+v"(e1) : "v"(d));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0 neg:[1,0,0]" : "+v"(re[j]) : "v"(ui), "v"(im2[j]));
          asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(im[j]) : "v"(ui), "v"(re2[j]));
        }
      }
    }
    double s = e0 + e1;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += re[j] + im[j] + (iters < 0 ? re2[j] + im2[j] : 0.0);
    if (s == 1.2345) out[0] = s;
  }
}

// scalar side of the island's interpreter: MODE 7 s_add_u32 / s_and_b32 / s_lshl_b32 chains, MODE 8 s_load_dwordx4 from 4 KiB (scalar cache hits)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void ks(const uint32_t *tab, uint32_t *out, int iters) {
  uint32_t a = blockIdx.x + 1, b = 0x9e3779b9u, c = 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 7)
      asm volatile(REP8("s_add_u32 %0, %0, %1\n\ts_and_b32 %2, %0, %1\n\ts_lshl_b32 %1, %1, 1\n\ts_xor_b32 %1, %1, %2\n\t"
                        "s_add_u32 %0, %0, %2\n\ts_or_b32 %2, %2, 1\n\ts_sub_u32 %1, %1, %0\n\ts_and_b32 %0, %0, 0xffff\n\t")
                   : "+s"(a), "+s"(b), "+s"(c) : : "scc");
    if (MODE == 8) {
      uint32_t off = (a * 16) & 0xff0;
      typedef uint32_t u4 __attribute__((ext_vector_type(4)));
      u4 r;
      asm volatile(REP8("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_add_u32 %2, %2, 16\n\ts_and_b32 %2, %2, 0xff0\n\t")
                   : "=&s"(r), "+s"(tab), "+s"(off) : : "scc", "memory");
      const uint32_t r0 = r.x, r3 = r.w;
      a += r0 + r3;
    }
  }
  if (a + b + c == 0x12345) out[0] = a;
}

template <int MODE> static void run_s(double secs, int per_iter, int iters, const char *unit) {
  uint32_t *tab, *out; CK(hipMalloc(&tab, 8192)); CK(hipMalloc(&out, 64));
  CK(hipMemset(tab, 1, 8192));
  const int blocks = 256 * 3;
  hipLaunchKernelGGL(ks<MODE>, dim3(blocks), dim3(256), 0, 0, tab, out, 10);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < secs) {
    for (int r = 0; r < 32; ++r) hipLaunchKernelGGL(ks<MODE>, dim3(blocks), dim3(256), 0, 0, tab, out, iters);
    CK(hipDeviceSynchronize());
    launches += 32;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double n = (double)launches * blocks * 4 * iters * per_iter;
  printf("%s/s %.4g  (%.4f per SIMD-cycle at 2.4 GHz = one per %.1f cycles)  seconds %.2f\n", unit, n / el, n / el / (1024 * 2.4e9),
         1024 * 2.4e9 * el / n, el);
}

template <int MODE> static void run(double secs, int per_iter, int iters, const char *unit) {
  double *out; CK(hipMalloc(&out, 64));
  const int blocks = 256 * 3;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  double el = 0;
  while (el < secs) {
    for (int r = 0; r < 32; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipDeviceSynchronize());
    launches += 32;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double n = (double)launches * blocks * 4 * iters * per_iter;
  printf("%s/s %.4g  (%.4f per SIMD-cycle at 2.4 GHz = one per %.1f cycles)  seconds %.2f\n", unit, n / el, n / el / (1024 * 2.4e9),
         1024 * 2.4e9 * el / n, el);
}

int main(int argc, char **argv) {
  const char *m = argc > 1 ? argv[1] : "layout";
  const double secs = argc > 2 ? atof(argv[2]) : 5.0;
  if (!strcmp(m, "layout")) return layout_and_check();
  if (!strcmp(m, "mfma4")) run<0>(secs, 64, 4000, "wave-instructions");
  else if (!strcmp(m, "mfma16")) run<1>(secs, 16, 4000, "wave-instructions");
  else if (!strcmp(m, "fma")) run<2>(secs, 64, 20000, "wave-instructions");
  else if (!strcmp(m, "add")) run<3>(secs, 64, 20000, "wave-instructions");
  else if (!strcmp(m, "dppbf")) run<4>(secs, 1, 480, "lane-butterflies-of-32-slots");
  else if (!strcmp(m, "grp")) run<5>(secs, 1, 1000, "fused-groups-of-32-slots");
  else if (!strcmp(m, "grpmix")) run<6>(secs, 1, 1000, "fused-groups-of-32-slots(+256 v_add_f64)");
  else if (!strcmp(m, "salu")) run_s<7>(secs, 64, 20000, "scalar-ALU wave-instructions");
  else if (!strcmp(m, "smem")) run_s<8>(secs, 8, 4000, "s_load_dwordx4 (+ wait + 2 scalar ALU)");
  else printf("idle\n");
  return 0;
}
