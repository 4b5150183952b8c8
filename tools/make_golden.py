#!/usr/bin/env python3
"""Generate tests/golden/* from the REFERENCE itself (run in the build container only).

This script is the provenance of every committed golden vector.  It never
copies reference source: it *runs* the reference and stores inputs/outputs.

Two reference executables are used:
  (1) oracle/_ref/libxgates.so -- the reference's src/lib/xgates.cc compiled
      unmodified by oracle/Makefile (g++ -O3 -ffast-math, the flags of the
      reference's make_libxgates.sh:63-65).  Called directly with NumPy arrays.
  (2) the reference's Python package /root/reference/src/lib (circuit.qc, ops,
      state), imported read-only.  It imports `absl.flags`, which is not
      installed in this image and cannot be; a 3-module flags/app stub is
      written to a temp dir so the import succeeds.  The stub supplies only the
      flag *values* (tensor_width=128/64, empty dump paths); no arithmetic of
      the reference is replaced.  circuit.apply1/applyc (module globals bound
      at src/lib/circuit.py:40-41 to libxgates) are wrapped to RECORD every
      native call before forwarding it to (1).

Fixtures (see SURVEY.md section 8c for the G-numbering):
  g1_qft12.npz        12q QFT of |101100101110>, complex128 (config 1 pin)
  g2_libq_qft12.npz   same circuit through dumpers.libq + the reference libq
                      sources (float, printed) -> plumbing pin at 1e-6
  g3_single.npz       8q random state x {20 gates} x every target
  g4_ctl.npz          applyc for every ordered (ctl,tgt), n=6 (+ sampled n=9)
  g5_*.npz            recorded gate traces + final states of reference circuits
  g6_qft22.npz        22q QFT: 4096 sampled amplitudes + norm + argmax
  g7_c64.npz          complex64 variants (reference default tensor_width=64)
  py_fallback.npz     State.apply1/applyc pure-Python loops (state.py:80-125),
                      incl. the negative-control sequence of circuit_test.py:97-102
"""
import importlib.util
import math
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
NO_CTL = -(2 ** 31)


def load_ref_xgates():
  path = os.path.join(ROOT, 'oracle', '_ref', 'libxgates.so')
  spec = importlib.util.spec_from_file_location('libxgates', path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def install_absl_stub(tmp):
  os.makedirs(os.path.join(tmp, 'absl', 'testing'))
  open(os.path.join(tmp, 'absl', '__init__.py'), 'w').close()
  open(os.path.join(tmp, 'absl', 'testing', '__init__.py'), 'w').close()
  with open(os.path.join(tmp, 'absl', 'flags.py'), 'w') as f:
    f.write(
        'class _F:\n  pass\nFLAGS = _F()\n'
        'def _d(name, default, help=None, **kw):\n  setattr(FLAGS, name, default)\n'
        'DEFINE_integer = DEFINE_string = DEFINE_bool = DEFINE_boolean = DEFINE_float = _d\n')
  with open(os.path.join(tmp, 'absl', 'app.py'), 'w') as f:
    f.write('class UsageError(Exception):\n  pass\n'
            'def run(main):\n  import sys\n  main(sys.argv[:1])\n')
  sys.path.insert(0, tmp)


class Recorder:
  """Wraps the reference's native entry points and records the call stream."""

  def __init__(self, xg):
    self.xg = xg
    self.ops, self.gates = [], []

  def reset(self):
    self.ops, self.gates = [], []

  def apply1(self, psi, gate, nbits, tgt, bw):
    self.ops.append((NO_CTL, int(tgt)))
    self.gates.append(np.asarray(gate, dtype=np.complex128).copy())
    return self.xg.apply1(psi, gate, nbits, tgt, bw)

  def applyc(self, psi, gate, nbits, ctl, tgt, bw):
    self.ops.append((int(ctl), int(tgt)))
    self.gates.append(np.asarray(gate, dtype=np.complex128).copy())
    return self.xg.applyc(psi, gate, nbits, ctl, tgt, bw)

  def arrays(self):
    ops = np.array(self.ops, dtype=np.int32).reshape(-1, 2)
    g = np.array(self.gates, dtype=np.complex128).reshape(-1, 4)
    return ops, g.view(np.float64).reshape(-1, 8)


def main():
  os.makedirs(OUT, exist_ok=True)
  xg = load_ref_xgates()
  sys.modules['libxgates'] = xg  # so the reference's circuit.py picks it up
  tmp = tempfile.mkdtemp(prefix='qcc_golden_')
  install_absl_stub(tmp)
  sys.path.insert(0, REF)
  from absl import flags
  from src.lib import circuit, ops, state, dumpers  # the reference
  assert circuit.apply1 is xg.apply1, 'reference did not bind libxgates'
  flags.FLAGS.tensor_width = 128
  rec = Recorder(xg)
  circuit.apply1, circuit.applyc = rec.apply1, rec.applyc

  def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in kw.items()})

  def trace(name, qc_builder, **extra):
    rec.reset()
    qc, init = qc_builder()
    o, g = rec.arrays()
    save(name, nbits=qc.psi.nbits, init=init, ops=o, gates=g,
         final=np.asarray(qc.psi, dtype=np.complex128), **extra)

  # ---- G1: 12-qubit QFT, complex128 ------------------------------------
  bits12 = (1, 0, 1, 1, 0, 0, 1, 0, 1, 1, 1, 0)

  def qft12():
    qc = circuit.qc('qft12')
    reg = qc.reg(12, bits12)
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    qc.qft(reg)
    return qc, init
  trace('g1_qft12.npz', qft12, bits=np.array(bits12))

  # ---- G2: same circuit through dumpers.libq + reference libq ----------
  flags.FLAGS.libq = ''
  qcn = circuit.qc('qft12_libq', eager=False)
  regn = qcn.reg(12, bits12)
  qcn.qft(regn)
  src = dumpers.libq(qcn.ir)
  cc = os.path.join(tmp, 'qft12_libq.cc')
  with open(cc, 'w') as f:
    f.write(src)
  exe = os.path.join(tmp, 'qft12_libq')
  subprocess.check_call(['g++', '-O3', '-ffast-math', '-I' + REF + '/src/libq', cc,
                         os.path.join(ROOT, 'oracle', '_ref', 'libq.a'), '-o', exe])
  txt = subprocess.check_output([exe]).decode()
  idx, amp = [], []
  for line in txt.splitlines():
    line = line.strip()
    if '|' not in line or 'i|' not in line:
      continue
    head, rest = line.split('i|', 1)
    re_s, im_s = head.split()
    idx.append(int(rest.split('>')[0]))
    amp.append(complex(float(re_s), float(im_s)))
  save('g2_libq_qft12.npz', nbits=12, libq_state=np.array(idx, dtype=np.int64),
       amp=np.array(amp, dtype=np.complex64), bits=np.array(bits12),
       note='libq is little-endian: libq state s <-> qcc index bitreverse_12(s)')

  # ---- G3: single-qubit gates on every target, direct native calls -----
  rng = np.random.default_rng(7)
  n3 = 8
  psi0 = rng.standard_normal(1 << n3) + 1j * rng.standard_normal(1 << n3)
  psi0 /= np.linalg.norm(psi0)
  glist = [('h', ops.Hadamard()), ('x', ops.PauliX()), ('y', ops.PauliY()),
           ('z', ops.PauliZ()), ('s', ops.Sgate()), ('t', ops.Tgate()),
           ('v', ops.Vgate()), ('yroot', ops.Yroot()), ('u1', ops.U1(1.1)),
           ('rx', ops.RotationX(0.7)), ('ry', ops.RotationY(0.7)),
           ('rz', ops.RotationZ(0.7))]
  glist += [(n + 'dag', g.adjoint()) for n, g in list(glist)]
  names, gmat, outs = [], [], []
  for name, g in glist:
    for t in range(n3):
      p = psi0.copy()
      xg.apply1(p, np.asarray(g, dtype=np.complex128).reshape(4), n3, t, 128)
      names.append(f'{name}:{t}')
      gmat.append(np.asarray(g, dtype=np.complex128).reshape(4))
      outs.append(p)
  save('g3_single.npz', nbits=n3, psi0=psi0, names=np.array(names),
       gates=np.array(gmat), outs=np.array(outs))

  # ---- G4: controlled gates, every ordered (ctl,tgt) -------------------
  from scipy.stats import unitary_group
  u = unitary_group.rvs(2, random_state=11)
  cg = [('u', u), ('u1', np.asarray(ops.U1(0.3))), ('x', np.asarray(ops.PauliX())),
        ('z', np.asarray(ops.PauliZ())), ('h', np.asarray(ops.Hadamard()))]
  for n4, pairs in ((6, None), (9, [(0, 8), (8, 0), (3, 4), (4, 3), (1, 7), (6, 2)])):
    p0 = rng.standard_normal(1 << n4) + 1j * rng.standard_normal(1 << n4)
    p0 /= np.linalg.norm(p0)
    if pairs is None:
      pairs = [(c, t) for c in range(n4) for t in range(n4) if c != t]
    names, gmat, outs = [], [], []
    for gname, g in cg:
      for c, t in pairs:
        p = p0.copy()
        xg.applyc(p, np.asarray(g, dtype=np.complex128).reshape(4), n4, c, t, 128)
        names.append(f'{gname}:{c}:{t}')
        gmat.append(np.asarray(g, dtype=np.complex128).reshape(4))
        outs.append(p)
    save(f'g4_ctl_n{n4}.npz', nbits=n4, psi0=p0, names=np.array(names),
         gates=np.array(gmat), outs=np.array(outs))

  # ---- G5: recorded traces of reference circuits -----------------------
  for n in (4, 7, 10):
    def qft_n(n=n):
      qc = circuit.qc('qft')
      reg = qc.reg(n, (0b1011001110 >> (10 - n)))
      init = np.asarray(qc.psi, dtype=np.complex128).copy()
      qc.qft(reg)
      qc.inverse_qft(list(reg)[: n // 2])
      return qc, init
    trace(f'g5_qft_iqft_n{n}.npz', qft_n)

  def accel_block2():
    # circuit_test.py:92-107 -- includes NEGATIVE control indices (quirk Q7)
    qc = circuit.qc()
    qc.bitstring(1, 0, 1, 0, 1)
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    for n in range(5):
      qc.h(n)
      for i in range(0, 5):
        qc.cu1(n - (i + 1), n, math.pi / float(2 ** (i + 1)))
      qc.h(n)
    return qc, init
  trace('g5_negctl.npz', accel_block2)

  def accel_block1():
    qc = circuit.qc()
    qc.bitstring(1, 0, 1, 0)
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    for i in range(4):
      qc.x(i); qc.y(i); qc.z(i); qc.h(i)
      if i:
        qc.cu1(0, i, 1.1)
    return qc, init
  trace('g5_accel1.npz', accel_block1)

  def mctl():
    qc = circuit.qc('mc')
    qc.reg(4, (1, 0, 1, 1))
    aux = qc.reg(4, 0)
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    qc.h([0, 1, 2, 3])
    qc.multi_control([0, [1], 2], 3, aux, ops.PauliX(), 'mc-x')
    qc.multi_control([0, 1, [2], 3], 7, aux, ops.Hadamard(), 'mc-h')
    qc.cswap(0, 1, 2)
    qc.swap(0, 3)
    qc.ccu1(0, 1, 2, 0.77)
    qc.crx(1, 2, 0.3); qc.cry([0], 3, 0.4); qc.crz(3, 0, 0.5)
    return qc, init
  trace('g5_multi_control.npz', mctl)

  # supremacy circuit (supremacy.py:123-158,208-253) with seeded Python RNG
  sys.argv = sys.argv[:1]
  import io, contextlib
  from src import supremacy
  for n, seed in ((12, 0), (14, 1)):
    def sup(n=n, seed=seed):
      random.seed(seed)
      with contextlib.redirect_stdout(io.StringIO()):
        states = supremacy.build_circuit(n, 20)
      # sim_circuit creates its own qc; replicate its body through the API so
      # that we can keep the qc (same calls, same RNG consumption order).
      qc = circuit.qc('Supremacy Circuit')
      qc.reg(n)
      init = np.asarray(qc.psi, dtype=np.complex128).copy()
      G = supremacy.Gate
      for d in range(20):
        s = states[d]
        for i in range(n):
          if s[i] == G.UNK:
            continue
          if s[i] == G.T:
            qc.t(i)
          if s[i] == G.H:
            qc.h(i)
          if s[i] == G.U:
            if random.randint(0, 1) == 0:
              qc.v(i)
            else:
              qc.yroot(i)
          if s[i] == G.CZ:
            if i < n - 1 and s[i + 1] == G.CZ:
              qc.cz(i, i + 1)
              s[i + 1] = G.UNK
            if i < n - 6 and s[i + 6] == G.CZ:
              qc.cz(i, i + 6)
              s[i + 6] = G.UNK
      return qc, init
    trace(f'g5_supremacy_n{n}_s{seed}.npz', sup, seed=seed, depth=20)

  # Grover circuit (grover.py:124-168), nbits=6 -> 12 qubits, fixed marked string
  def grover6():
    nb = 6
    bits = [1, 0, 1, 0, 1, 1]
    qc = circuit.qc('Grover')
    reg = qc.reg(nb, 0)
    qc.reg(1, 1)
    aux = qc.reg(nb - 1, 0)
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    idx = list(range(nb))
    iterations = int(math.pi / 4 * math.sqrt(2 ** nb))
    qc.h([i for i in range(nb + 1)])
    for _ in range(iterations):
      for i in idx:
        if bits[i] == 0:
          qc.apply1(ops.PauliX(), i, 'x')
      qc.multi_control(reg, nb, aux, ops.PauliX(), 'Phase Inversion')
      for i in idx:
        if bits[i] == 0:
          qc.apply1(ops.PauliX(), i, 'x')
      qc.h(idx); qc.x(idx)
      qc.multi_control(reg, nb, aux, ops.PauliZ(), 'Mean Inversion')
      qc.x(idx); qc.h(idx)
    return qc, init
  trace('g5_grover6.npz', grover6, marked=np.array([1, 0, 1, 0, 1, 1]))

  # ---- G6: 22-qubit QFT, sampled ---------------------------------------
  n6 = 22
  x6 = 0x2CB9A5 & ((1 << n6) - 1)
  psi = np.zeros(1 << n6, dtype=np.complex128)
  psi[x6] = 1
  h = np.asarray(ops.Hadamard(), dtype=np.complex128).reshape(4)
  for i in reversed(range(n6)):
    xg.apply1(psi, h, n6, i, 128)
    for j in reversed(range(i)):
      g = np.asarray(ops.U1(np.pi / 2 ** (i - j)), dtype=np.complex128).reshape(4)
      xg.applyc(psi, g, n6, i, j, 128)
  samp = np.random.default_rng(22).integers(0, 1 << n6, size=4096)
  save('g6_qft22.npz', nbits=n6, x=x6, idx=samp, amp=psi[samp],
       norm2=float(np.vdot(psi, psi).real), sum=psi.sum())

  # ---- G7: complex64 (reference default) -------------------------------
  flags.FLAGS.tensor_width = 64
  n7 = 8
  p0 = (rng.standard_normal(1 << n7) + 1j * rng.standard_normal(1 << n7)).astype(np.complex64)
  p0 /= np.linalg.norm(p0)
  names, gmat, outs = [], [], []
  for name, g in glist[:12]:
    g64 = np.asarray(g, dtype=np.complex64).reshape(4)
    for t in (0, 3, 7):
      p = p0.copy()
      xg.apply1(p, g64, n7, t, 64)
      names.append(f'{name}:{t}'); gmat.append(g64); outs.append(p)
    for c, t in ((0, 7), (7, 0), (3, 4)):
      p = p0.copy()
      xg.applyc(p, g64, n7, c, t, 64)
      names.append(f'c{name}:{c}:{t}'); gmat.append(g64); outs.append(p)
  save('g7_c64.npz', nbits=n7, psi0=p0, names=np.array(names), gates=np.array(gmat),
       outs=np.array(outs))
  flags.FLAGS.tensor_width = 128

  # ---- Python fallback loops (state.py:80-125), the spec of the native path
  psi = state.bitstring(1, 0, 1, 0, 1)
  init = np.asarray(psi, dtype=np.complex128).copy()
  o, g = [], []
  for n in range(5):
    psi.apply1(ops.Hadamard(), n); o.append((NO_CTL, n)); g.append(np.asarray(ops.Hadamard()).reshape(4))
    for i in range(0, 5):
      gg = ops.U1(math.pi / float(2 ** (i + 1)))
      psi.applyc(gg, n - (i + 1), n); o.append((n - (i + 1), n)); g.append(np.asarray(gg).reshape(4))
  rs = np.random.default_rng(5)
  for _ in range(40):
    c, t = rs.choice(5, size=2, replace=False)
    gg = ops.Operator(unitary_group.rvs(2, random_state=int(rs.integers(1 << 30))))
    if rs.integers(2):
      psi.apply1(gg, int(t)); o.append((NO_CTL, int(t)))
    else:
      psi.applyc(gg, int(c), int(t)); o.append((int(c), int(t)))
    g.append(np.asarray(gg, dtype=np.complex128).reshape(4))
  save('py_fallback.npz', nbits=5, init=init, ops=np.array(o, dtype=np.int32),
       gates=np.array(g, dtype=np.complex128).view(np.float64).reshape(-1, 8),
       final=np.asarray(psi, dtype=np.complex128))


if __name__ == '__main__':
  main()
