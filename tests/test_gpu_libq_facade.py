"""a qcc circuit transpiled to a libq C++ program (the text format of
src/lib/dumpers.py:40-86) compiles against include/libq.h and runs on the GPU;
its print_qureg output matches the reference libq's own output (golden G2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from qcc_amd.lib import circuit, helper, tensor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def emit_libq_program(qc):
  """Same program text shape as the reference's `--libq` dumper produces."""
  lines = ['#include <math.h>', '#include <stdio.h>', '#include <stdlib.h>', '', '#include "libq.h"', '',
           'int main(int argc, char* argv[]) {', '']
  total = sum(size for _, size, _ in qc.ir.regset)
  lines.append(f'  libq::qureg* q = libq::new_qureg(0, {total});')
  pos = 0
  for _, _, reg in qc.ir.regset:
    for v in reg.val:
      if v == 1:
        lines.append(f'  libq::x({pos}, q);')
      pos += 1
  for node in qc.ir.gates:
    if not node.is_gate():
      continue
    args = [str(node.idx0)] if node.is_single() else [str(node.ctl), str(node.idx1)]
    if node.val is not None:
      args.append(helper.pi_fractions(node.val, 'M_PI'))
    lines.append(f'  libq::{node.name}({", ".join(args)}, q);')
  lines += ['', '  libq::flush(q);', '  libq::print_qureg(q);', '  libq::delete_qureg(q);', '  return EXIT_SUCCESS;', '}']
  return '\n'.join(lines) + '\n'


def parse_print_qureg(text):
  idx, amp = [], []
  for line in text.splitlines():
    line = line.strip()
    if 'i|' not in line:
      continue
    head, rest = line.split('i|', 1)
    re_s, im_s = head.split()
    idx.append(int(rest.split('>')[0]))
    amp.append(complex(float(re_s), float(im_s)))
  return np.array(idx), np.array(amp)


def test_transpiled_qft12_runs_on_gpu_and_matches_reference_libq(golden_dir, tmp_path):
  g2 = np.load(os.path.join(golden_dir, 'g2_libq_qft12.npz'))
  tensor.set_tensor_width(128)
  qc = circuit.qc('qft12_libq', eager=False)
  reg = qc.reg(12, tuple(int(b) for b in g2['bits']))
  qc.qft(reg)
  src = tmp_path / 'prog.cc'
  src.write_text(emit_libq_program(qc))
  exe = tmp_path / 'prog'
  libdir = os.path.join(ROOT, 'qcc_amd')
  subprocess.check_call(['g++', '-std=c++17', '-O2', str(src), '-I' + os.path.join(ROOT, 'include'),
                         '-L' + libdir, '-lqcc_hip', '-Wl,-rpath,' + libdir, '-o', str(exe)])
  out = subprocess.check_output([str(exe)], text=True)
  assert '# of qubits        : 12' in out
  idx, amp = parse_print_qureg(out)
  assert len(idx) == 4096
  got = np.zeros(4096, dtype=np.complex128)
  got[idx] = amp
  ref = np.zeros(4096, dtype=np.complex128)
  ref[g2['libq_state']] = g2['amp']
  assert np.max(np.abs(got - ref)) < 2e-6          # reference libq is float and prints 6 decimals


def _build_driver(tmp_path):
  exe = tmp_path / 'libq_driver'
  libdir = os.path.join(ROOT, 'qcc_amd')
  subprocess.check_call(['g++', '-std=c++17', '-O2', os.path.join(ROOT, 'tests', 'libq_driver.cc'),
                         '-I' + os.path.join(ROOT, 'include'), '-L' + libdir, '-lqcc_hip',
                         '-Wl,-rpath,' + libdir, '-o', str(exe)])
  return exe


def _run_driver(exe, tmp_path, cases):
  inp, outp = tmp_path / 'cases.txt', tmp_path / 'dense.bin'
  with open(inp, 'w') as f:
    f.write(f'{len(cases)}\n')
    for w, init, ops in cases:
      f.write(f'{w} {init} {len(ops)}\n')
      for op in ops:
        name, a, b, c, gamma = op[:5]
        f.write(f'{name} {a} {b} {c} {gamma!r}' + ''.join(f' {float(x)!r}' for x in op[5:]) + '\n')   # gate1: + 8 numbers
  subprocess.check_call([str(exe), str(inp), str(outp)], stdout=subprocess.DEVNULL)
  raw = np.fromfile(outp, dtype=np.complex128)
  out, off = [], 0
  for w, _, _ in cases:
    out.append(raw[off:off + (1 << w)])
    off += 1 << w
  assert off == raw.size
  return out


def test_libq_gate_set_matches_reference_libq(golden_dir, tmp_path):
  """SURVEY 8a row A7 (src/libq/gates.cc:9-151): x, y, z, h, t, u1, cu1, cx, cz, ccx, walsh through
  include/libq.h on the GPU -- every target / every ordered pair at 6 qubits on a dense entangled
  state, sparse inputs, and two QFT adders as the reference's dumper transpiles them -- against what
  the reference's own libq (oracle/_ref/libq.a, float) computed for the identical calls (golden G8)."""
  from tests.test_oracle_golden import libq_cases
  cases = libq_cases(golden_dir)
  exe = _build_driver(tmp_path)
  got = _run_driver(exe, tmp_path, [(w, init, ops) for w, init, ops, _ in cases])
  names = set()
  for (w, init, ops, dense), g in zip(cases, got):
    names |= {o[0] for o in ops}
    err = np.max(np.abs(g - dense))
    assert err < 3e-6, (w, init, ops[-1], err)          # the reference is complex<float>
    assert abs(np.vdot(g, g).real - 1) < 1e-12          # ours is complex128
  assert names >= {'x', 'y', 'z', 'h', 't', 'u1', 'cu1', 'cx', 'cz', 'ccx', 'walsh'}


def test_libq_root_gates_match_ops_matrices(oracle, tmp_path):
  """v, yroot, cv, cv_adj: the reference's libq versions are broken (gates.cc:9-15,48-54,96-118 apply
  the gate once per stored state), so the facade implements the matrices of
  src/lib/ops.py:152-162 (sqrt(X), sqrt(Y), controlled sqrt(X) and its adjoint) and is checked
  against the dense oracle with those matrices."""
  from qcc_amd import gates
  w = 6
  prep = [('walsh', w, 0, 0, 0.0)] + [('u1', i, 0, 0, 0.21 * (i + 1)) for i in range(w)] + [('cu1', 0, 4, 0, 0.6), ('h', 2, 0, 0, 0.0)]
  cases = []
  for t in range(w):
    cases.append((w, 0b011010, prep + [('v', t, 0, 0, 0.0)]))
    cases.append((w, 0b011010, prep + [('yroot', t, 0, 0, 0.0)]))
  for c in range(w):
    for t in range(w):
      if c != t:
        cases.append((w, 0b110001, prep + [('cv', c, t, 0, 0.0)]))
        cases.append((w, 0b110001, prep + [('cv_adj', c, t, 0, 0.0)]))
  got = _run_driver(_build_driver(tmp_path), tmp_path, cases)
  import cmath
  s = 1 / np.sqrt(2)
  v = np.asarray(gates.vgate(), dtype=np.complex128).reshape(2, 2)
  mats = {'h': np.array([[s, s], [s, -s]]), 'v': v, 'yroot': np.asarray(gates.yroot(), dtype=np.complex128).reshape(2, 2),
          'cv': v, 'cv_adj': v.conj().T}
  rev = np.array([int(format(k, f'0{w}b')[::-1], 2) for k in range(1 << w)])
  for (w_, init, ops), g in zip(cases, got):
    psi = np.zeros(1 << w, dtype=np.complex128)
    psi[int(format(init, f'0{w}b')[::-1], 2)] = 1
    for name, a, b, c, gamma in ops:
      if name == 'walsh':
        for i in range(a):
          oracle.apply1(psi, mats['h'].reshape(4), w, i)
      elif name == 'u1':
        oracle.apply1(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, a)
      elif name == 'cu1':
        oracle.applyc(psi, np.array([1, 0, 0, cmath.exp(1j * np.float32(gamma))]), w, a, b)
      elif name in ('h', 'v', 'yroot'):
        oracle.apply1(psi, mats[name].reshape(4), w, a)
      else:
        oracle.applyc(psi, mats[name].reshape(4), w, a, b)
    assert np.max(np.abs(g - psi[rev])) < 1e-7, ops[-1]   # float angles in the u1 calls


def test_libq_gate1_matches_oracle_and_reference_libq(oracle, golden_dir, tmp_path):
  """SURVEY 8a row A6 (src/libq/libq.h:69, apply.cc:78-176): an arbitrary 2x2 through the libq boundary.  libq_gate1 of
  include/libq.h on the GPU -- random unitaries, non-unitary matrices and H on EVERY target of dense 6-, 8- and 10-qubit
  registers and of single basis states -- against the oracle at 1e-10 (same float matrix entries, complex128 arithmetic)
  and against what the reference's own libq_gate1 computed for the identical calls (golden G10, complex<float>)."""
  from tests.test_oracle_golden import libq_gate1_cases, libq_gate1_expected, libq_gate1_prep
  cases = libq_gate1_cases(golden_dir)
  progs = []
  for w, init, dp, t, m, _ in cases:
    nums = [float(x) for z in m for x in (z.real, z.imag)]
    progs.append((w, init, (libq_gate1_prep(w) if dp else []) + [('gate1', t, 0, 0, 0.0, *nums)]))
  # three more matrices the fixture does not hold, oracle only: a projector, a nilpotent matrix, a scaled rotation
  extra = [np.array([1, 0, 0, 0], dtype=np.complex64), np.array([0, 1, 0, 0], dtype=np.complex64),
           np.array([1.5, -0.5j, 0.5j, 1.5], dtype=np.complex64)]
  extra_cases = []
  for w in (6, 9):
    for m in extra:
      for t in range(w):
        extra_cases.append((w, 5, 1, t, m))
        nums = [float(x) for z in m for x in (z.real, z.imag)]
        progs.append((w, 5, libq_gate1_prep(w) + [('gate1', t, 0, 0, 0.0, *nums)]))
  got = _run_driver(_build_driver(tmp_path), tmp_path, progs)
  for (w, init, dp, t, m, dense), g in zip(cases, got):
    want = libq_gate1_expected(oracle, w, init, dp, t, m)
    assert np.max(np.abs(g - want)) < 1e-10, (w, init, dp, t)
    assert np.max(np.abs(g - dense)) < 3e-6 * max(1.0, float(np.max(np.abs(want)))), (w, init, dp, t)
  for (w, init, dp, t, m), g in zip(extra_cases, got[len(cases):]):
    assert np.max(np.abs(g - libq_gate1_expected(oracle, w, init, dp, t, m))) < 1e-10, (w, t, m)
