"""ctypes loader for the CPU oracle (oracle/xgates_oracle.c).  TEST-ONLY.

Nothing under qcc_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_CTL = -(2 ** 31)


class Oracle:
  def __init__(self, lib):
    self.lib = lib
    for sfx, real in (('c128', ctypes.c_double), ('c64', ctypes.c_float)):
      f = getattr(lib, 'oracle_apply1_' + sfx)
      f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
      f.restype = ctypes.c_int
      f = getattr(lib, 'oracle_applyc_' + sfx)
      f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
      f.restype = ctypes.c_int
    lib.oracle_run_stream_c128.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p]
    lib.oracle_run_stream_c128.restype = ctypes.c_int
    lib.oracle_run_stream_c128_mt.argtypes = lib.oracle_run_stream_c128.argtypes
    lib.oracle_run_stream_c128_mt.restype = ctypes.c_int
    lib.oracle_init_basis_c128_mt.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64]
    lib.oracle_init_basis_c128_mt.restype = ctypes.c_int

  @staticmethod
  def _sfx(psi):
    assert psi.flags.c_contiguous and psi.dtype in (np.complex128, np.complex64)
    return 'c128' if psi.dtype == np.complex128 else 'c64'

  def apply1(self, psi, gate, nbits, tgt):
    sfx = self._sfx(psi)
    g = np.ascontiguousarray(np.asarray(gate, dtype=psi.dtype).reshape(4))
    rc = getattr(self.lib, 'oracle_apply1_' + sfx)(psi.ctypes.data, g.ctypes.data, nbits, tgt)
    if rc:
      raise ValueError(f'oracle_apply1 rc={rc}')

  def applyc(self, psi, gate, nbits, ctl, tgt):
    sfx = self._sfx(psi)
    g = np.ascontiguousarray(np.asarray(gate, dtype=psi.dtype).reshape(4))
    rc = getattr(self.lib, 'oracle_applyc_' + sfx)(psi.ctypes.data, g.ctypes.data, nbits, ctl, tgt)
    if rc:
      raise ValueError(f'oracle_applyc rc={rc}')

  def run_stream(self, psi, nbits, ops, gates):
    """ops: int32 [G,2] (ctl or NO_CTL, tgt); gates: float64 [G,8]."""
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    gates = np.ascontiguousarray(gates, dtype=np.float64)
    rc = self.lib.oracle_run_stream_c128(psi.ctypes.data, nbits, len(ops),
                                         ops.ctypes.data, gates.ctypes.data)
    if rc:
      raise ValueError(f'oracle_run_stream rc={rc}')


  def init_basis_mt(self, psi, nbits, index):
    """psi := |index>, pages first touched by the OpenMP threads that will work on them."""
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous and psi.size == 1 << nbits
    rc = self.lib.oracle_init_basis_c128_mt(psi.ctypes.data, nbits, int(index))
    if rc:
      raise ValueError(f'oracle_init_basis_mt rc={rc}')

  def run_stream_mt(self, psi, nbits, ops, gates):
    """All-core variant (OpenMP when the library was built with it): in-range controls only."""
    assert psi.dtype == np.complex128 and psi.flags.c_contiguous
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    gates = np.ascontiguousarray(gates, dtype=np.float64)
    rc = self.lib.oracle_run_stream_c128_mt(psi.ctypes.data, nbits, len(ops), ops.ctypes.data, gates.ctypes.data)
    if rc:
      raise ValueError(f'oracle_run_stream_mt rc={rc}')


def load(fast=False, omp=False):
  name = 'liboracle_omp.so' if omp else ('liboracle_fast.so' if fast else 'liboracle.so')
  path = os.path.join(ROOT, 'oracle', '_build', name)
  if not os.path.exists(path):
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '_build/' + name])
  return Oracle(ctypes.CDLL(path))


def load_ref_xgates():
  """The reference's own xgates.cc build (oracle/_ref), or None."""
  import importlib.util
  path = os.path.join(ROOT, 'oracle', '_ref', 'libxgates.so')
  if not os.path.exists(path):
    return None
  spec = importlib.util.spec_from_file_location('libxgates', path)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod
