"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports
every symbol include/qcc_hip.h declares, and fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from qcc_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
  native.build()
  return native.load()


def _declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'qcc_hip.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(qh_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
  syms = _declared_symbols()
  assert len(syms) >= 30
  for s in syms:
    assert hasattr(lib, s), f'{s} declared in include/qcc_hip.h but not exported'
    assert s in native.SIGNATURES, f'{s} has no ctypes signature in qcc_amd/native.py'
  assert sorted(native.SIGNATURES) == syms


def test_no_cpu_fallback(lib):
  if native.device_count() > 0:
    pytest.skip('a GPU is visible')
  h = ctypes.c_void_p()
  rc = lib.qh_create(10, 128, 0, ctypes.byref(h))
  assert rc == native.QH_ERR_NO_DEVICE
  assert b'no CPU fallback' in lib.qh_last_error()
  psi = np.zeros(4, dtype=np.complex128)
  g = np.eye(2, dtype=np.complex128).view(np.float64).reshape(8)
  rc = lib.qh_host_apply1(psi.ctypes.data, g.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 2, 0, 128)
  assert rc == native.QH_ERR_NO_DEVICE
  from qcc_amd import device
  with pytest.raises(native.QhError):
    device.DeviceState(8)


def test_argument_errors_are_status_codes(lib):
  h = ctypes.c_void_p()
  assert lib.qh_create_dry(0, 128, ctypes.byref(h)) == native.QH_ERR_ARG
  assert lib.qh_create_dry(10, 32, ctypes.byref(h)) == native.QH_ERR_BAD_DTYPE
  assert lib.qh_create_dry(10, 128, ctypes.byref(h)) == native.QH_OK
  g = (ctypes.c_double * 8)(1, 0, 0, 0, 0, 0, 1, 0)
  assert lib.qh_apply1(h, 10, g) == native.QH_ERR_BAD_QUBIT    # xgates.cc:28-32 would exit()
  assert lib.qh_apply1(h, -1, g) == native.QH_ERR_BAD_QUBIT
  assert lib.qh_applyc(h, 3, 3, g) == native.QH_ERR_SAME_QUBIT
  assert lib.qh_applyc(h, 10, 3, g) == native.QH_ERR_BAD_QUBIT  # positive out-of-range control
  assert lib.qh_applyc(h, -1, 3, g) == native.QH_OK             # quirk Q7: legal, maybe no-op
  assert lib.qh_apply_bits(h, 1 << 4, 4, g) == native.QH_ERR_SAME_QUBIT
  assert lib.qh_apply_bits(h, 1 << 10, 4, g) == native.QH_ERR_BAD_QUBIT
  assert lib.qh_apply1(None, 0, g) == native.QH_ERR_ARG
  assert lib.qh_destroy(h) == native.QH_OK


def test_shard_semantics_dry(lib):
  """Shard-bit controls / diagonal targets never need the GPU to decide."""
  h = ctypes.c_void_p()
  assert lib.qh_create_dry(10, 128, ctypes.byref(h)) == 0
  assert lib.qh_set_shard(h, 12, 0b10) == 0
  had = (ctypes.c_double * 8)(*(np.array([1, 1, 1, -1], dtype=np.complex128) / np.sqrt(2)).view(np.float64))
  cu1 = (ctypes.c_double * 8)(1, 0, 0, 0, 0, 0, 0.6, 0.8)
  # dense gate on a shard bit (physical bit 11 = qubit 0): must be refused
  assert lib.qh_apply1(h, 0, had) == native.QH_ERR_NONLOCAL
  assert b'exchange' in lib.qh_last_error()
  # control on shard bit 10 (qubit 1): this shard has it clear -> no-op
  assert lib.qh_applyc(h, 1, 5, had) == 0
  # control on shard bit 11 (qubit 0): set -> runs as an uncontrolled local gate
  assert lib.qh_applyc(h, 0, 5, had) == 0
  # diagonal target on a shard bit is legal
  assert lib.qh_applyc(h, 5, 0, cu1) == 0
  s = native.QhStats()
  lib.qh_get_stats(h, ctypes.byref(s))
  assert s.gates_submitted == 3 and s.gates_noop == 1 and s.kernels_launched == 2
  # S = 2^10*16 B: dense under a satisfied shard-bit control = 2S; CU1 whose target is a
  # set shard bit scales the control-set half of the shard = S
  assert s.bytes_algorithmic == 2 * 16384 + 16384
  lib.qh_destroy(h)


def test_bitmap_roundtrip_dry(lib):
  h = ctypes.c_void_p()
  assert lib.qh_create_dry(6, 128, ctypes.byref(h)) == 0
  assert lib.qh_remap_swap(h, 0, 5) == 0
  assert lib.qh_remap_swap(h, 2, 3) == 0
  o = ctypes.c_uint64()
  for v in (0b000001, 0b100110, 0b111111, 0b001000):
    lib.qh_logical_to_phys(h, v, ctypes.byref(o))
    p = o.value
    lib.qh_phys_to_logical(h, p, ctypes.byref(o))
    assert o.value == v
  lib.qh_logical_to_phys(h, 0b000001, ctypes.byref(o))
  assert o.value == 0b100000
  lib.qh_destroy(h)


def test_failed_init_product_leaves_the_layout_alone_dry(lib):
  """qh_init_product reset the bit map BEFORE validating; an error return then scrambled a live state."""
  h = ctypes.c_void_p()
  assert lib.qh_create_dry(10, 128, ctypes.byref(h)) == 0
  assert lib.qh_remap_swap(h, 3, 8) == 0 and lib.qh_remap_swap(h, 4, 9) == 0
  assert lib.qh_set_fusion(h, native.QH_FUSE_SWEEP) == 0
  def bitmap():
    bm = (ctypes.c_int32 * 64)()
    assert lib.qh_get_bitmap(h, bm) == 0
    return list(bm)[:10]
  before = bitmap()
  assert before != list(range(10))
  had = (ctypes.c_double * 8)(*(np.array([1, 1, 1, -1], dtype=np.complex128) / np.sqrt(2)).view(np.float64))
  assert lib.qh_apply1(h, 2, had) == 0
  pend = ctypes.c_uint64()
  assert lib.qh_pending_gates(h, ctypes.byref(pend)) == 0 and pend.value == 1
  # factor sizes that do not add up, an out-of-range basis index, a basis factor without basis[]: all refused ...
  tab = np.array([0.6, 0.8j], dtype=np.complex128)
  for nq, amps, basis in (([4, 4], [None, None], [1, 2]), ([5, 5], [None, None], [1, 99]), ([1, 8], [tab.ctypes.data, None], [0, 7])):
    c_nq = (ctypes.c_int32 * len(nq))(*nq)
    c_amps = (ctypes.c_void_p * len(nq))(*amps)
    c_basis = (ctypes.c_uint64 * len(nq))(*basis)
    assert lib.qh_init_product(h, len(nq), c_nq, c_amps, c_basis) == native.QH_ERR_ARG
    # ... and nothing moved: same bit map, the queued gate still queued
    assert bitmap() == before
    assert lib.qh_pending_gates(h, ctypes.byref(pend)) == 0 and pend.value == 1
  # a valid call replaces the state: canonical order, empty queue
  c_nq = (ctypes.c_int32 * 2)(1, 9)
  c_amps = (ctypes.c_void_p * 2)(tab.ctypes.data, None)
  c_basis = (ctypes.c_uint64 * 2)(0, 7)
  assert lib.qh_init_product(h, 2, c_nq, c_amps, c_basis) == 0
  assert bitmap() == list(range(10))
  assert lib.qh_pending_gates(h, ctypes.byref(pend)) == 0 and pend.value == 0
  lib.qh_destroy(h)


def test_libq_facade_header_links(tmp_path, lib):
  """include/libq.h: every declared function resolves against libqcc_hip.so (link only)."""
  import re
  import subprocess
  text = open(os.path.join(ROOT, 'include', 'libq.h')).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  text = re.sub(r'//.*', '', text)
  decl = re.findall(r'^\s*(?:void|qureg \*|float)\s*\*?(\w+)\s*\(([^)]*)\)\s*;', text, flags=re.M)
  names = [n for n, _ in decl]
  assert {'new_qureg', 'print_qureg', 'h', 'cu1', 'ccx', 'cv_adj', 'flush', 'libq_gate1'} <= set(names) and len(names) >= 22
  body = '\n'.join(f'  (void)&libq::{n};' for n in names)
  # take the address of each function through a volatile sink so the linker must resolve it
  src = tmp_path / 'link_all.cc'
  src.write_text('#include "libq.h"\nvolatile const void* sink;\nint main() {\n' +
                 '\n'.join(f'  sink = (const void*)&libq::{n};' for n in names if n not in ('x', 'y', 'z', 'h', 't', 'v')) +
                 '\n  void (*f1)(int, libq::qureg*) = &libq::x; sink = (const void*)f1;'
                 '\n  f1 = &libq::y; sink = (const void*)f1; f1 = &libq::z; sink = (const void*)f1;'
                 '\n  f1 = &libq::h; sink = (const void*)f1; f1 = &libq::t; sink = (const void*)f1;'
                 '\n  f1 = &libq::v; sink = (const void*)f1;\n  return 0;\n}\n')
  libdir = os.path.join(ROOT, 'qcc_amd')
  subprocess.check_call(['g++', '-std=c++17', str(src), '-I' + os.path.join(ROOT, 'include'), '-L' + libdir,
                         '-lqcc_hip', '-Wl,-rpath,' + libdir, '-o', str(tmp_path / 'link_all')])
