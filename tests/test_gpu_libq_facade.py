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
