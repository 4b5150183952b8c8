#!/usr/bin/env python3
"""Round-2 additions to tests/golden/ (run in the build container only; same provenance rules as
tools/make_golden.py, whose helpers it uses: the reference is RUN, never copied).

Completes the fixture list of SURVEY 8(c):
  g3_single_n10.npz          G3 at 10 qubits (every gate x every target, direct native calls)
  g4_ctl_n{7,8,10}.npz       G4 for n = 7, 8 (every ordered pair) and 10 (sampled pairs)
  g5_arith_quantum6.npz      recorded trace of arith_quantum(6, a, b) (src/arith_quantum.py:59-75)
  g5_supremacy_n16_s2.npz    supremacy trace, 16 qubits depth 20
  g5_grover7.npz / g5_grover8.npz   Grover circuit, nbits 7 and 8 (14 / 16 qubits)
  g9_supremacy_n20_s0.npz    supremacy 20 qubits depth 20: trace + 4096 sampled amplitudes + norm
  g6_qft24.npz, g6_qft26.npz 24/26-qubit QFT through the reference build: samples + norm + argmax
  g8_libq_gates.npz          the reference's libq (src/libq/*.cc, float, sparse) run on small
                             programs covering x,y,z,h,t,u1,cu1,cx,cz,ccx,walsh for every target /
                             ordered pair at 6 qubits and a QFT adder at reduced width
                             (tests/libq_driver.cc linked with oracle/_ref/libq.a)
"""
import contextlib
import io
import math
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

REF, ROOT, OUT, NO_CTL = mg.REF, mg.ROOT, mg.OUT, mg.NO_CTL


def libq_cases():
  """(width, initval, [(name, a, b, c, gamma)]) programs: a dense, entangled preparation, then ONE
  gate under test -- for every target / ordered pair -- so that each gate sees every amplitude."""
  w = 6
  prep = [('walsh', w, 0, 0, 0.0)] + [('u1', i, 0, 0, 0.3 * (i + 1)) for i in range(w)]
  prep += [('cu1', 0, 3, 0, 0.5), ('cu1', 2, 5, 0, 1.1), ('h', 1, 0, 0, 0.0), ('cu1', 1, 4, 0, -0.7), ('h', 4, 0, 0, 0.0)]
  cases = []
  for name in ('x', 'y', 'z', 'h', 't'):
    for t in range(w):
      cases.append((w, 0b100110, prep + [(name, t, 0, 0, 0.0)]))
  for t in range(w):
    cases.append((w, 0b100110, prep + [('u1', t, 0, 0, 0.77)]))
  for name in ('cx', 'cz', 'cu1'):
    for c in range(w):
      for t in range(w):
        if c != t:
          cases.append((w, 0b010011, prep + [(name, c, t, 0, 0.9 if name == 'cu1' else 0.0)]))
  for c0 in range(w):
    for c1 in range(c0 + 1, w):
      for t in range(w):
        if t not in (c0, c1):
          cases.append((w, 0b110101, prep + [('ccx', c0, c1, t, 0.0)]))
  for init in (0, 0b101101):
    cases.append((w, init, [('walsh', w, 0, 0, 0.0)]))
  # sparse inputs too (the reference's hash path adds/removes basis states)
  for t in range(w):
    cases.append((w, 0b000111, [('h', t, 0, 0, 0.0), ('cx', t, (t + 1) % w, 0, 0.0), ('y', (t + 2) % w, 0, 0, 0.0)]))
  return cases


def write_cases(path, cases):
  with open(path, 'w') as f:
    f.write(f'{len(cases)}\n')
    for w, init, ops in cases:
      f.write(f'{w} {init} {len(ops)}\n')
      for name, a, b, c, gamma in ops:
        f.write(f'{name} {a} {b} {c} {gamma!r}\n')


def encode_cases(cases):
  names = sorted({op[0] for _, _, ops in cases for op in ops})
  head = np.array([(w, init, len(ops)) for w, init, ops in cases], dtype=np.int64)
  body = np.array([(names.index(n), a, b, c) for _, _, ops in cases for (n, a, b, c, _) in ops], dtype=np.int32)
  gam = np.array([g for _, _, ops in cases for (*_, g) in ops], dtype=np.float64)
  return np.array(names), head, body, gam


def main():
  os.makedirs(OUT, exist_ok=True)
  xg = mg.load_ref_xgates()
  sys.modules['libxgates'] = xg
  tmp = tempfile.mkdtemp(prefix='qcc_golden_r2_')
  mg.install_absl_stub(tmp)
  sys.path.insert(0, REF)
  from absl import flags
  from src.lib import circuit, ops, helper  # the reference
  assert circuit.apply1 is xg.apply1
  flags.FLAGS.tensor_width = 128
  rec = mg.Recorder(xg)
  circuit.apply1, circuit.applyc = rec.apply1, rec.applyc

  def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name), **kw)
    print('wrote', name, {k: getattr(v, 'shape', v) for k, v in kw.items()})

  def trace(name, qc_builder, **extra):
    rec.reset()
    qc, init = qc_builder()
    o, g = rec.arrays()
    save(name, nbits=qc.psi.nbits, init=init, ops=o, gates=g, final=np.asarray(qc.psi, dtype=np.complex128), **extra)

  # ---- G3 at 10 qubits ----------------------------------------------------------------
  rng = np.random.default_rng(7)
  n3 = 10
  psi0 = rng.standard_normal(1 << n3) + 1j * rng.standard_normal(1 << n3)
  psi0 /= np.linalg.norm(psi0)
  glist = [('h', ops.Hadamard()), ('x', ops.PauliX()), ('y', ops.PauliY()), ('z', ops.PauliZ()), ('s', ops.Sgate()),
           ('t', ops.Tgate()), ('v', ops.Vgate()), ('yroot', ops.Yroot()), ('u1', ops.U1(1.1)),
           ('rx', ops.RotationX(0.7)), ('ry', ops.RotationY(0.7)), ('rz', ops.RotationZ(0.7))]
  names, gmat, outs = [], [], []    # (the adjoints are in the 8-qubit set)
  for name, g in glist:
    for t in range(n3):
      p = psi0.copy()
      xg.apply1(p, np.asarray(g, dtype=np.complex128).reshape(4), n3, t, 128)
      names.append(f'{name}:{t}')
      gmat.append(np.asarray(g, dtype=np.complex128).reshape(4))
      outs.append(p)
  save('g3_single_n10.npz', nbits=n3, psi0=psi0, names=np.array(names), gates=np.array(gmat), outs=np.array(outs))

  # ---- G4 at 7, 8 (all ordered pairs) and 10 (sampled) ----------------------------------
  from scipy.stats import unitary_group
  u = unitary_group.rvs(2, random_state=11)
  cg = [('u', u), ('u1', np.asarray(ops.U1(0.3))), ('x', np.asarray(ops.PauliX())), ('z', np.asarray(ops.PauliZ()))]
  for n4, pairs in ((7, None), (8, None), (10, [(0, 9), (9, 0), (4, 5), (5, 4), (1, 8), (7, 2), (3, 9), (9, 6)])):
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
    save(f'g4_ctl_n{n4}.npz', nbits=n4, psi0=p0, names=np.array(names), gates=np.array(gmat), outs=np.array(outs))

  # ---- G5 arith_quantum(6, a, b) -----------------------------------------------------
  sys.argv = sys.argv[:1]
  from src import arith_quantum as aq

  def arith():
    n, a_val, b_val = 6, 21, 38
    qc = circuit.qc('qadd')
    a = qc.reg(n + 1, helper.val2bits(a_val, n)[::-1], name='a')
    b = qc.reg(n + 1, helper.val2bits(b_val, n)[::-1], name='b')
    init = np.asarray(qc.psi, dtype=np.complex128).copy()
    for i in range(n + 1):
      aq.qft(qc, a, n - i)
    for i in range(n + 1):
      aq.evolve(qc, a, b, n - i, 1.0)
    for i in range(n + 1):
      aq.inverse_qft(qc, a, i)
    aq.check_result(qc.psi, a_val, b_val, n + 1, 1.0)
    return qc, init
  trace('g5_arith_quantum6.npz', arith, a=21, b=38)

  # ---- supremacy n = 16 (full) and 20 (sampled) -------------------------------------------
  from src import supremacy

  def sup_builder(n, seed):
    def sup():
      random.seed(seed)
      with contextlib.redirect_stdout(io.StringIO()):
        states = supremacy.build_circuit(n, 20)
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
    return sup
  trace('g5_supremacy_n16_s2.npz', sup_builder(16, 2), seed=2, depth=20)
  rec.reset()
  qc20, _ = sup_builder(20, 0)()
  o, g = rec.arrays()
  psi = np.asarray(qc20.psi, dtype=np.complex128)
  samp = np.random.default_rng(20).integers(0, 1 << 20, size=4096)
  save('g9_supremacy_n20_s0.npz', nbits=20, init_index=0, ops=o, gates=g, idx=samp, amp=psi[samp],
       norm2=float(np.vdot(psi, psi).real), seed=0, depth=20)

  # ---- Grover nbits 7, 8 -----------------------------------------------------------------
  def grover_builder(nb, bits):
    def grover():
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
    return grover
  trace('g5_grover7.npz', grover_builder(7, [1, 0, 1, 0, 1, 1, 0]), marked=np.array([1, 0, 1, 0, 1, 1, 0]))
  trace('g5_grover8.npz', grover_builder(8, [0, 1, 1, 0, 1, 0, 0, 1]), marked=np.array([0, 1, 1, 0, 1, 0, 0, 1]))

  # ---- G6 at 24 and 26 qubits ----------------------------------------------------------
  h = np.asarray(ops.Hadamard(), dtype=np.complex128).reshape(4)
  for n6 in (24, 26):
    x6 = 0x2CB9A5E & ((1 << n6) - 1)
    psi = np.zeros(1 << n6, dtype=np.complex128)
    psi[x6] = 1
    for i in reversed(range(n6)):
      xg.apply1(psi, h, n6, i, 128)
      for j in reversed(range(i)):
        g = np.asarray(ops.U1(np.pi / 2 ** (i - j)), dtype=np.complex128).reshape(4)
        xg.applyc(psi, g, n6, i, j, 128)
    samp = np.random.default_rng(n6).integers(0, 1 << n6, size=4096)
    p = psi.real ** 2 + psi.imag ** 2
    save(f'g6_qft{n6}.npz', nbits=n6, x=x6, idx=samp, amp=psi[samp], norm2=float(p.sum()), argmax=int(np.argmax(p)),
         pmax=float(p.max()))
    del psi, p

  # ---- G8: the reference's libq on small programs ------------------------------------------
  cases = libq_cases()
  # a QFT adder at reduced width, transpiled by the reference's own dumper (src/lib/dumpers.py:40-86)
  flags.FLAGS.libq = ''
  from src.lib import dumpers

  def adder_program(n, a_val, b_val):
    qc = circuit.qc('qadd', eager=False)
    a = qc.reg(n + 1, helper.val2bits(a_val, n)[::-1], name='a')
    b = qc.reg(n + 1, helper.val2bits(b_val, n)[::-1], name='b')
    for i in range(n + 1):
      aq.qft(qc, a, n - i)
    for i in range(n + 1):
      aq.evolve(qc, a, b, n - i, 1.0)
    for i in range(n + 1):
      aq.inverse_qft(qc, a, i)
    prog = []
    pos = 0
    for _, _, reg in qc.ir.regset:       # what dumpers.libq emits: x on the set bits, then the gates
      for v in reg.val:
        if v == 1:
          prog.append(('x', pos, 0, 0, 0.0))
        pos += 1
    for node in qc.ir.gates:
      if not node.is_gate():
        continue
      if node.is_single():
        prog.append((node.name, node.idx0, 0, 0, float(node.val or 0.0)))
      else:
        prog.append((node.name, node.ctl, node.idx1, 0, float(node.val or 0.0)))
    text = dumpers.libq(qc.ir)           # cross-check: one libq:: call per program entry
    assert text.count('libq::') - 5 == len(prog), (text.count('libq::'), len(prog))
    return (pos, 0, prog)
  cases.append(adder_program(5, 2, 3))     # "addition of 2, 3" (src/libq/libq_arith_test.cc) at 12 qubits
  cases.append(adder_program(6, 21, 38))
  inp = os.path.join(tmp, 'cases.txt')
  outp = os.path.join(tmp, 'dense.bin')
  write_cases(inp, cases)
  exe = os.path.join(tmp, 'libq_driver_ref')
  subprocess.check_call(['g++', '-O2', '-std=c++11', '-I' + REF + '/src/libq', os.path.join(ROOT, 'tests', 'libq_driver.cc'),
                         os.path.join(ROOT, 'oracle', '_ref', 'libq.a'), '-o', exe])
  subprocess.check_call([exe, inp, outp], stdout=subprocess.DEVNULL)
  raw = np.fromfile(outp, dtype=np.complex128)
  sizes = [1 << w for w, _, _ in cases]
  assert raw.size == sum(sizes)
  names, head, body, gam = encode_cases(cases)
  save('g8_libq_gates.npz', op_names=names, case_head=head, case_ops=body, case_gamma=gam,
       dense=raw.astype(np.complex64), note='dense[k] = amplitude of libq basis state k (little-endian), cases concatenated')
  norms = []
  off = 0
  for s in sizes:
    norms.append(float(np.sum(np.abs(raw[off:off + s]) ** 2)))
    off += s
  print('libq cases:', len(cases), 'norm range', min(norms), max(norms))


if __name__ == '__main__':
  main()
