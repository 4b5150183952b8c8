#!/usr/bin/env python3
"""Round-6 addition to tests/golden/ (run in the build container only; the reference is RUN, never copied):

  g10_libq_gate1.npz   the reference's own libq::libq_gate1 (src/libq/libq.h:69, body src/libq/apply.cc:78-176; compiled
                       unmodified into oracle/_ref/libq.a by oracle/Makefile) on EVERY target of 6-, 8- and 10-qubit
                       registers -- a dense entangled state (prep below) and a single basis state -- with a random unitary,
                       a random NON-unitary matrix with four non-zero entries, and H; run by tests/libq_driver.cc
                       ("gate1" lines).  The fixture stores width / initval / prep flag / target / the four float matrix
                       entries per case and the dense results (complex64: the reference's libq is complex<float>).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
from tests.test_oracle_golden import libq_gate1_prep  # noqa: E402


def main():
  from scipy.stats import unitary_group
  rng = np.random.default_rng(606)
  s = 1 / np.sqrt(2)
  mats = [unitary_group.rvs(2, random_state=61).reshape(4), (rng.standard_normal(4) + 1j * rng.standard_normal(4)) * 0.7,
          np.array([s, s, s, -s], dtype=np.complex128), unitary_group.rvs(2, random_state=62).reshape(4)]
  mats = [m.astype(np.complex64) for m in mats]          # what the C interface takes (cmplx = complex<float>)
  cases = []
  for w in (6, 8, 10):
    for dense_prep in (1, 0):
      for mi, m in enumerate(mats):
        for t in range(w):
          init = int(rng.integers(0, 1 << w))
          cases.append((w, init, dense_prep, t, m))
  tmp = tempfile.mkdtemp()
  inp, outp = os.path.join(tmp, 'cases.txt'), os.path.join(tmp, 'dense.bin')
  with open(inp, 'w') as f:
    f.write(f'{len(cases)}\n')
    for w, init, dp, t, m in cases:
      prep = libq_gate1_prep(w) if dp else []
      f.write(f'{w} {init} {len(prep) + 1}\n')
      for name, a, b, c, gamma in prep:
        f.write(f'{name} {a} {b} {c} {gamma!r}\n')
      f.write(f'gate1 {t} 0 0 0.0 ' + ' '.join(f'{float(x)!r}' for z in m for x in (z.real, z.imag)) + '\n')
  exe = os.path.join(tmp, 'libq_driver_ref')
  subprocess.check_call(['g++', '-O2', '-std=c++11', '-I' + REF + '/src/libq', os.path.join(ROOT, 'tests', 'libq_driver.cc'),
                         os.path.join(ROOT, 'oracle', '_ref', 'libq.a'), '-o', exe])
  subprocess.check_call([exe, inp, outp], stdout=subprocess.DEVNULL)
  raw = np.fromfile(outp, dtype=np.complex128)
  assert raw.size == sum(1 << c[0] for c in cases)
  path = os.path.join(ROOT, 'tests', 'golden', 'g10_libq_gate1.npz')
  np.savez_compressed(path, width=np.array([c[0] for c in cases]), init=np.array([c[1] for c in cases], dtype=np.uint64),
                      dense_prep=np.array([c[2] for c in cases]), target=np.array([c[3] for c in cases]),
                      m=np.array([c[4] for c in cases], dtype=np.complex64), dense=raw.astype(np.complex64),
                      note='dense[k] = amplitude of libq basis state k (little-endian), cases concatenated; m row-major')
  print('wrote', path, os.path.getsize(path), 'bytes;', len(cases), 'cases')


if __name__ == '__main__':
  main()
