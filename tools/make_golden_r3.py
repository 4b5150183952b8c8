#!/usr/bin/env python3
"""Round-3 addition to tests/golden/ (run in the build container only; the reference is RUN, never copied):

  g4_pairs_n9_n10.npz   G4 of SURVEY 8(c) for EVERY ordered (ctl, tgt) at n = 9 and n = 10 (rounds 1-2 held sampled
                        pairs there): dense random unitary, U1, X, Z through the reference's own applyc
                        (src/lib/xgates.cc:45-67 compiled unmodified by oracle/Makefile -> oracle/_ref/libxgates.so).
                        648 cases.  To keep the fixture small each 2^n-amplitude output is stored as its 16 inner
                        products with fixed random probe vectors (stored too) -- 16 complex numbers per case pin all
                        2^n amplitudes: a wrong amplitude changes every product -- and the outputs of the first and
                        last pair of each gate in full.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_lib  # noqa: E402


def main():
  xg = oracle_lib.load_ref_xgates()
  if xg is None:
    raise SystemExit('oracle/_ref/libxgates.so missing: run `make -C oracle` where /root/reference exists')
  from scipy.stats import unitary_group
  rng = np.random.default_rng(93)
  u = unitary_group.rvs(2, random_state=29)
  x = np.array([[0, 1], [1, 0]], dtype=np.complex128)
  z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
  u1 = np.array([[1, 0], [0, np.exp(0.3j)]], dtype=np.complex128)
  out = {}
  for n in (9, 10):
    psi0 = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
    psi0 /= np.linalg.norm(psi0)
    probes = (rng.standard_normal((16, 1 << n)) + 1j * rng.standard_normal((16, 1 << n))) / np.sqrt(2 << n)
    names, gmat, proj, full_names, full = [], [], [], [], []
    pairs = [(c, t) for c in range(n) for t in range(n) if c != t]
    for gname, g in (('u', u), ('u1', u1), ('x', x), ('z', z)):
      for k, (c, t) in enumerate(pairs):
        p = psi0.copy()
        xg.applyc(p, np.ascontiguousarray(g, dtype=np.complex128).reshape(4), n, c, t, 128)
        names.append(f'{gname}:{c}:{t}')
        gmat.append(np.asarray(g, dtype=np.complex128).reshape(4))
        proj.append(probes.conj() @ p)
        if k in (0, len(pairs) - 1):
          full_names.append(names[-1])
          full.append(p)
    out.update({f'psi0_n{n}': psi0, f'probes_n{n}': probes, f'names_n{n}': np.array(names), f'gates_n{n}': np.array(gmat),
                f'proj_n{n}': np.array(proj), f'full_names_n{n}': np.array(full_names), f'full_n{n}': np.array(full)})
  path = os.path.join(ROOT, 'tests', 'golden', 'g4_pairs_n9_n10.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes;', sum(len(out[f'names_n{n}']) for n in (9, 10)), 'cases')


if __name__ == '__main__':
  main()
