/*
 * xgates_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99, 64-bit indices, no UB) of the reference's dense
 * gate-application algorithm.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker
 * or the timed CPU baseline -- never as the product path.  The product path is
 * qcc_amd/csrc (HIP, gfx950) and fails loudly when its library is missing.
 *
 * What is restated (reference = /root/reference, read-only):
 *   oracle_apply1_*  <- src/lib/xgates.cc:23-41   apply1<cmplx_type>()
 *                       (same loop nest as src/lib/state.py:80-100 State.apply1)
 *   oracle_applyc_*  <- src/lib/xgates.cc:45-67   applyc<cmplx_type>()
 *                       (same loop nest as src/lib/state.py:102-125 State.applyc)
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against
 *   (1) golden vectors produced by the reference's own compiled xgates.cc and
 *       by the reference's Python fallback loops (tools/make_golden.py ->
 *       tests/golden/), and
 *   (2) when oracle/_ref/libxgates.so exists (built from the reference source
 *       by oracle/Makefile), against live calls into that build.
 *
 * Conventions kept from the reference:
 *   - psi is interleaved (re,im), length 2^nbits, updated IN PLACE, both new
 *     values of a pair computed from the OLD values (xgates.cc:35-38).
 *   - gate is row-major [a b c d] (xgates.cc:18-21).
 *   - qubit numbers are big-endian: qubit q lives at index bit nbits-1-q
 *     (xgates.cc:26,48-49).
 *   - the control predicate is evaluated on (g * 2^nbits + i), where g is the
 *     base of the current 2^(tgt'+1) block (xgates.cc:58-59, state.py:119-121).
 *     For an in-range control that is simply bit ctl' of i.  For a control
 *     index outside [0, nbits) the Python fallback (arbitrary precision) tests
 *     bit (ctl'-nbits) of g; the reference's own test suite does exactly that
 *     (src/lib/circuit_test.py:97-102, negative control indices).  We restate
 *     the Python semantics (the C++ one is signed-overflow UB).
 *
 * Deliberate differences (documented in DESIGN.md):
 *   - 64-bit index arithmetic (reference: 32-bit int, breaks at nbits >= 31).
 *   - returns a status code instead of exit(EXIT_FAILURE) (xgates.cc:28-32).
 *   - complex products are written out in real arithmetic in the order
 *     (ar*br - ai*bi, ar*bi + ai*br); the reference is compiled -ffast-math,
 *     which permits the same naive formula (no NaN/Inf recovery path).
 */
#include <stdint.h>
#include <stddef.h>

#define ORACLE_OK 0
#define ORACLE_BAD_QUBIT 1
#define ORACLE_NO_CTL INT32_MIN /* negative controls are legal inputs (Q7) */

#define DEFINE_ORACLE(SUFFIX, REAL)                                            \
  int oracle_apply1_##SUFFIX(REAL *psi, const REAL *gate, int nbits,           \
                             int tgt) {                                        \
    /* xgates.cc:26 */                                                         \
    int p = nbits - tgt - 1;                                                   \
    if (nbits < 1 || nbits > 62 || p < 0 || p >= nbits) return ORACLE_BAD_QUBIT; \
    const uint64_t q2 = (uint64_t)1 << p;                                      \
    const uint64_t n = (uint64_t)1 << nbits;                                   \
    const REAL g0r = gate[0], g0i = gate[1], g1r = gate[2], g1i = gate[3];     \
    const REAL g2r = gate[4], g2i = gate[5], g3r = gate[6], g3i = gate[7];     \
    /* xgates.cc:33-40 */                                                      \
    for (uint64_t g = 0; g < n; g += q2 << 1) {                                \
      for (uint64_t i = g; i < g + q2; ++i) {                                  \
        REAL *a = psi + 2 * i, *b = psi + 2 * (i + q2);                        \
        const REAL ar = a[0], ai = a[1], br = b[0], bi = b[1];                 \
        const REAL t1r = (g0r * ar - g0i * ai) + (g1r * br - g1i * bi);        \
        const REAL t1i = (g0r * ai + g0i * ar) + (g1r * bi + g1i * br);        \
        const REAL t2r = (g2r * ar - g2i * ai) + (g3r * br - g3i * bi);        \
        const REAL t2i = (g2r * ai + g2i * ar) + (g3r * bi + g3i * br);        \
        a[0] = t1r; a[1] = t1i; b[0] = t2r; b[1] = t2i;                        \
      }                                                                        \
    }                                                                          \
    return ORACLE_OK;                                                          \
  }                                                                            \
                                                                               \
  int oracle_applyc_##SUFFIX(REAL *psi, const REAL *gate, int nbits, int ctl,  \
                             int tgt) {                                        \
    /* xgates.cc:48-49 */                                                      \
    int p = nbits - tgt - 1;                                                   \
    int c = nbits - ctl - 1;                                                   \
    if (nbits < 1 || nbits > 62 || p < 0 || p >= nbits) return ORACLE_BAD_QUBIT; \
    if (c < 0) return ORACLE_BAD_QUBIT; /* Python: negative shift ValueError */ \
    const uint64_t q2 = (uint64_t)1 << p;                                      \
    const uint64_t n = (uint64_t)1 << nbits;                                   \
    const REAL g0r = gate[0], g0i = gate[1], g1r = gate[2], g1i = gate[3];     \
    const REAL g2r = gate[4], g2i = gate[5], g3r = gate[6], g3i = gate[7];     \
    for (uint64_t g = 0; g < n; g += q2 << 1) {                                \
      for (uint64_t i = g; i < g + q2; ++i) {                                  \
        /* xgates.cc:58-59 / state.py:119-121: bit c of (g*2^nbits + i) */     \
        int on;                                                                \
        if (c < nbits) on = (int)((i >> c) & 1u);                              \
        else if (c - nbits < 64) on = (int)((g >> (c - nbits)) & 1u);          \
        else on = 0;                                                           \
        if (!on) continue;                                                     \
        REAL *a = psi + 2 * i, *b = psi + 2 * (i + q2);                        \
        const REAL ar = a[0], ai = a[1], br = b[0], bi = b[1];                 \
        const REAL t1r = (g0r * ar - g0i * ai) + (g1r * br - g1i * bi);        \
        const REAL t1i = (g0r * ai + g0i * ar) + (g1r * bi + g1i * br);        \
        const REAL t2r = (g2r * ar - g2i * ai) + (g3r * br - g3i * bi);        \
        const REAL t2i = (g2r * ai + g2i * ar) + (g3r * bi + g3i * br);        \
        a[0] = t1r; a[1] = t1i; b[0] = t2r; b[1] = t2i;                        \
      }                                                                        \
    }                                                                          \
    return ORACLE_OK;                                                          \
  }

DEFINE_ORACLE(c128, double)
DEFINE_ORACLE(c64, float)

/* Gate-stream driver used by the CPU-baseline leg of bench.py and by tests
 * that replay a recorded gate trace: ops[k] = {ctl (or ORACLE_NO_CTL), tgt}, gates[k] =
 * 8 doubles.  Qubit indices are the reference's big-endian numbers. */
int oracle_run_stream_c128(double *psi, int nbits, int64_t ngates,
                           const int32_t *ops, const double *gates) {
  for (int64_t k = 0; k < ngates; ++k) {
    int ctl = ops[2 * k], tgt = ops[2 * k + 1];
    int rc = (ctl == ORACLE_NO_CTL)
                 ? oracle_apply1_c128(psi, gates + 8 * k, nbits, tgt)
                 : oracle_applyc_c128(psi, gates + 8 * k, nbits, ctl, tgt);
    if (rc) return rc;
  }
  return ORACLE_OK;
}

/* All-core variant of the same stream driver, for the "all cores, for context" figure of the CPU
 * baseline: the pair loop of xgates.cc:33-40 / :56-65 flattened over the
 * pair index j (pair j = index with a zero inserted at bit p) so that OpenMP can split it whatever
 * the target bit is; same arithmetic per pair, in-range controls only.  Compiled with -fopenmp into
 * _build/liboracle_omp.so; checked against the serial functions in tests/test_oracle_golden.py. */
int oracle_run_stream_c128_mt(double *psi, int nbits, int64_t ngates,
                              const int32_t *ops, const double *gates) {
  if (nbits < 1 || nbits > 62) return ORACLE_BAD_QUBIT;
  const int64_t npairs = (int64_t)1 << (nbits - 1);
  for (int64_t k = 0; k < ngates; ++k) {
    const int ctl = ops[2 * k], tgt = ops[2 * k + 1];
    const int p = nbits - tgt - 1;
    const int c = (ctl == ORACLE_NO_CTL) ? -1 : nbits - ctl - 1;
    if (p < 0 || p >= nbits || (ctl != ORACLE_NO_CTL && (c < 0 || c >= nbits || c == p))) return ORACLE_BAD_QUBIT;
    const double *g = gates + 8 * k;
    const double g0r = g[0], g0i = g[1], g1r = g[2], g1i = g[3], g2r = g[4], g2i = g[5], g3r = g[6], g3i = g[7];
    const uint64_t q2 = (uint64_t)1 << p, low = q2 - 1;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < npairs; ++j) {
      const uint64_t i = (((uint64_t)j & ~low) << 1) | ((uint64_t)j & low);
      if (c >= 0 && !((i >> c) & 1u)) continue;
      double *a = psi + 2 * i, *b = psi + 2 * (i + q2);
      const double ar = a[0], ai = a[1], br = b[0], bi = b[1];
      a[0] = (g0r * ar - g0i * ai) + (g1r * br - g1i * bi);
      a[1] = (g0r * ai + g0i * ar) + (g1r * bi + g1i * br);
      b[0] = (g2r * ar - g2i * ai) + (g3r * br - g3i * bi);
      b[1] = (g2r * ai + g2i * ar) + (g3r * bi + g3i * br);
    }
  }
  return ORACLE_OK;
}

/* First-touch initialisation by the threads that will work on the pages (NUMA placement of the
 * all-core baseline): psi := |index>. */
int oracle_init_basis_c128_mt(double *psi, int nbits, uint64_t index) {
  if (nbits < 1 || nbits > 62) return ORACLE_BAD_QUBIT;
  const int64_t n2 = (int64_t)2 << nbits;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n2; ++i) psi[i] = 0.0;
  psi[2 * index] = 1.0;
  return ORACLE_OK;
}
