// libq_driver.cc -- runs small libq programs and dumps the resulting register DENSE.
//
// Test infrastructure for SURVEY 8a row A7 (the libq gate set, src/libq/gates.cc:9-151): the
// SAME driver is compiled twice,
//   * against the reference's src/libq/libq.h + oracle/_ref/libq.a (tools/make_golden.py, in the
//     builder container only) to record what the reference's sparse complex<float> libq computes,
//   * against include/libq.h + libqcc_hip.so (tests/test_gpu_libq_facade.py, on the GPU box) to
//     run the identical calls through the facade on the MI355X,
// and the two dense vectors are compared.
//
// Input (text file argv[1]):   ncases, then per case  "width initval nops"  and nops lines
//   "name a b c gamma"  (unused arguments are 0; "gate1 target 0 0 0" is followed by the eight numbers re im of m[0..3] and
//   goes to libq::libq_gate1, src/libq/libq.h:69).  Output (binary file argv[2]): per case
//   2^width complex<double>, index = libq basis state.
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "libq.h"

static void apply(const char *name, int a, int b, int c, double gamma, libq::qureg *q) {
  if (!strcmp(name, "x")) libq::x(a, q);
  else if (!strcmp(name, "y")) libq::y(a, q);
  else if (!strcmp(name, "z")) libq::z(a, q);
  else if (!strcmp(name, "h")) libq::h(a, q);
  else if (!strcmp(name, "t")) libq::t(a, q);
  else if (!strcmp(name, "v")) libq::v(a, q);
  else if (!strcmp(name, "yroot")) libq::yroot(a, q);
  else if (!strcmp(name, "walsh")) libq::walsh(a, q);
  else if (!strcmp(name, "cx")) libq::cx(a, b, q);
  else if (!strcmp(name, "cz")) libq::cz(a, b, q);
  else if (!strcmp(name, "ccx")) libq::ccx(a, b, c, q);
  else if (!strcmp(name, "u1")) libq::u1(a, (float)gamma, q);
  else if (!strcmp(name, "cu1")) libq::cu1(a, b, (float)gamma, q);
  else if (!strcmp(name, "cv")) libq::cv(a, b, q);
  else if (!strcmp(name, "cv_adj")) libq::cv_adj(a, b, q);
  else { fprintf(stderr, "unknown gate %s\n", name); exit(2); }
}

static void dump(libq::qureg *q, FILE *out) {
  const size_t n = (size_t)1 << q->width;
  std::vector<std::complex<double>> dense(n);
#ifdef QCC_LIBQ_FACADE_H_
  if (qh_download(q->handle, dense.data(), 0, n) != QH_OK) { fprintf(stderr, "%s\n", qh_last_error()); exit(3); }
#else
  for (int i = 0; i < q->size; ++i) dense[q->state[i]] += std::complex<double>(q->amplitude[i].real(), q->amplitude[i].imag());
#endif
  fwrite(dense.data(), sizeof(dense[0]), n, out);
}

int main(int argc, char **argv) {
  if (argc < 3) return 1;
  FILE *in = fopen(argv[1], "r"), *out = fopen(argv[2], "wb");
  if (!in || !out) return 1;
  int ncases = 0;
  if (fscanf(in, "%d", &ncases) != 1) return 1;
  for (int k = 0; k < ncases; ++k) {
    int width, nops;
    unsigned long long init;
    if (fscanf(in, "%d %llu %d", &width, &init, &nops) != 3) return 1;
    libq::qureg *q = libq::new_qureg(init, width);
    for (int i = 0; i < nops; ++i) {
      char name[16];
      int a, b, c;
      double gamma;
      if (fscanf(in, "%15s %d %d %d %lf", name, &a, &b, &c, &gamma) != 5) return 1;
      if (!strcmp(name, "gate1")) {
        double v[8];
        for (double &x : v)
          if (fscanf(in, "%lf", &x) != 1) return 1;
        libq::cmplx m[4];
        for (int k = 0; k < 4; ++k) m[k] = libq::cmplx((float)v[2 * k], (float)v[2 * k + 1]);
        libq::libq_gate1(a, m, q);
        continue;
      }
      apply(name, a, b, c, gamma, q);
    }
    dump(q, out);
    libq::delete_qureg(q);
  }
  fclose(out);
  return 0;
}
