// libq_facade.cc -- implementation of include/libq.h on top of the C-ABI
// (include/qcc_hip.h).  Plain host C++: no HIP in this file.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/libq.h"

namespace libq {
namespace {

void check(int rc, const char *what) {
  if (rc != QH_OK) {
    fprintf(stderr, "libq facade: %s failed: %s\n", what, qh_last_error());
    exit(EXIT_FAILURE);
  }
}

void gate(qureg *reg, uint64_t ctl_mask, int target, const double g[8]) {
  check(qh_apply_bits(reg->handle, ctl_mask, target, g), "gate");
}

const double kS = 0.70710678118654752440;
const double GX[8] = {0, 0, 1, 0, 1, 0, 0, 0};
const double GY[8] = {0, 0, 0, -1, 0, 1, 0, 0};
const double GZ[8] = {1, 0, 0, 0, 0, 0, -1, 0};
const double GH[8] = {kS, 0, kS, 0, kS, 0, -kS, 0};
const double GV[8] = {0.5, 0.5, 0.5, -0.5, 0.5, -0.5, 0.5, 0.5};      // sqrt(X)   ops.py:152-154
const double GVA[8] = {0.5, -0.5, 0.5, 0.5, 0.5, 0.5, 0.5, -0.5};     // its adjoint
const double GYR[8] = {0.5, 0.5, -0.5, -0.5, 0.5, 0.5, 0.5, 0.5};     // sqrt(Y)   ops.py:158-162

void phase(qureg *reg, uint64_t ctl_mask, int target, double gamma) {
  const double g[8] = {1, 0, 0, 0, 0, 0, std::cos(gamma), std::sin(gamma)};
  gate(reg, ctl_mask, target, g);
}

// download and visit the amplitudes that a sparse libq register would hold:
// probability >= 1e-6 / 2^width (apply.cc:149-171)
template <typename F> int visit(qureg *reg, F f) {
  const uint64_t n = 1ull << reg->width;
  std::vector<std::complex<double>> amp(n);
  check(qh_download(reg->handle, amp.data(), 0, n), "download");
  const double limit = 1e-6 / (double)n;
  int count = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const double p = std::norm(amp[i]);
    if (p >= limit) {
      f(i, amp[i], p);
      ++count;
    }
  }
  reg->size = count;
  if (count > reg->maxsize) reg->maxsize = count;
  return count;
}

}  // namespace

float probability(cmplx a) { return a.real() * a.real() + a.imag() * a.imag(); }

// src/libq/libq.h:69, apply.cc:78-176: an arbitrary 2x2 on index bit `target` (libq's bit order), row-major m
void libq_gate1(int target, cmplx m[4], qureg *reg) {
  double g[8];
  for (int k = 0; k < 4; ++k) {
    g[2 * k] = m[k].real();
    g[2 * k + 1] = m[k].imag();
  }
  gate(reg, 0, target, g);
}

qureg *new_qureg(state_t initval, int width) {
  qureg *reg = new qureg;
  reg->width = width;
  reg->size = 1;
  reg->maxsize = 0;
  reg->hash_computes = 0;
  check(qh_create(width, 128, 0, &reg->handle), "qh_create");
  check(qh_set_fusion(reg->handle, QH_FUSE_SWEEP), "qh_set_fusion");
  check(qh_init_basis(reg->handle, initval), "qh_init_basis");
  return reg;
}

void delete_qureg(qureg *reg) {
  if (!reg) return;
  qh_destroy(reg->handle);
  delete reg;
}

void print_qureg(qureg *reg) {
  printf("States with non-zero probability:\n");
  const int width = reg->width;
  visit(reg, [&](uint64_t i, std::complex<double> a, double p) {
    printf("  % f %+fi|%llu> (%e) (|", a.real(), a.imag(), (unsigned long long)i, p);
    for (int j = width - 1; j >= 0; --j) {
      if (j % 4 == 3) printf(" ");
      printf("%i", (int)((i >> j) & 1ull));
    }
    printf(">)\n");
  });
}

void print_qureg_stats(qureg *reg) {
  visit(reg, [](uint64_t, std::complex<double>, double) {});
  printf("# of qubits        : %d\n", reg->width);
  printf("# of hash computes : %d\n", reg->hash_computes);
  printf("Maximum # of states: %d, theoretical: %d, %.3f%%\n", reg->maxsize, 2 << reg->width,
         100.0 * reg->maxsize / (2 << reg->width));
}

void flush(qureg *reg) {
  check(qh_sync(reg->handle), "qh_sync");
  print_qureg_stats(reg);
}

void x(int target, qureg *reg) { gate(reg, 0, target, GX); }
void y(int target, qureg *reg) { gate(reg, 0, target, GY); }
void z(int target, qureg *reg) { gate(reg, 0, target, GZ); }
void h(int target, qureg *reg) { gate(reg, 0, target, GH); }
void t(int target, qureg *reg) { phase(reg, 0, target, M_PI / 4.0); }
void v(int target, qureg *reg) { gate(reg, 0, target, GV); }
void yroot(int target, qureg *reg) { gate(reg, 0, target, GYR); }
void walsh(int width, qureg *reg) {
  for (int i = 0; i < width; ++i) h(i, reg);
}
void cx(int control, int target, qureg *reg) { gate(reg, 1ull << control, target, GX); }
void cz(int control, int target, qureg *reg) { gate(reg, 1ull << control, target, GZ); }
void ccx(int c0, int c1, int target, qureg *reg) { gate(reg, (1ull << c0) | (1ull << c1), target, GX); }
void u1(int target, float gamma, qureg *reg) { phase(reg, 0, target, gamma); }
void cu1(int control, int target, float gamma, qureg *reg) { phase(reg, 1ull << control, target, gamma); }
void cv(int control, int target, qureg *reg) { gate(reg, 1ull << control, target, GV); }
void cv_adj(int control, int target, qureg *reg) { gate(reg, 1ull << control, target, GVA); }

}  // namespace libq
