"""ctypes binding of include/qcc_hip.h (libqcc_hip.so) + the build recipe.

This is the only place the shared library is loaded.  It fails loudly: a
missing library raises ImportError-like RuntimeError from load(); a missing GPU
makes qh_create return QH_ERR_NO_DEVICE, surfaced as QhError.
"""
import ctypes
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB_PATH = os.environ.get('QCC_HIP_LIB') or os.path.join(PKG, 'libqcc_hip.so')  # env: A/B builds only
SOURCES = [os.path.join(PKG, 'csrc', f) for f in
           ('engine.hip', 'kernels_gate.hip.h', 'kernels_sweep.hip.h', 'planner.h', 'exchange.hip.h',
            'sweep_island_rb2.inc', 'sweep_island_rb3.inc', 'sweep_island_rb4.inc',
            'sweep_island_rb5.inc', 'sweep_island_f32_rb2.inc', 'sweep_island_f32_rb3.inc',
            'sweep_island_f32_rb4.inc', 'sweep_island_f32_rb5.inc', 'sweep_island_f32_rb6.inc', 'sweep_handlers.inc',
            'libq_facade.cc')]
HEADERS = [os.path.join(ROOT, 'include', 'libq.h')]
HEADER = os.path.join(ROOT, 'include', 'qcc_hip.h')

QH_OK = 0
QH_ERR_BAD_QUBIT, QH_ERR_SAME_QUBIT, QH_ERR_BAD_DTYPE, QH_ERR_HIP = 1, 2, 3, 4
QH_ERR_ARG, QH_ERR_NOMEM, QH_ERR_NO_DEVICE, QH_ERR_NONLOCAL, QH_ERR_COMM = 5, 6, 7, 8, 9
QH_FUSE_OFF, QH_FUSE_SWEEP = 0, 1

_u64, _i32, _vp, _dp = ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)


class QhStats(ctypes.Structure):
  _fields_ = [('gates_submitted', _u64), ('kernels_launched', _u64), ('sweeps', _u64),
              ('bytes_algorithmic', _u64), ('bytes_swept', _u64), ('gates_noop', _u64)]

  def as_dict(self):
    return {k: int(getattr(self, k)) for k, _ in self._fields_}


# name -> (restype, argtypes); every symbol declared in include/qcc_hip.h
SIGNATURES = {
    'qh_last_error': (ctypes.c_char_p, []),
    'qh_version': (_i32, []),
    'qh_device_count': (_i32, [ctypes.POINTER(_i32)]),
    'qh_create': (_i32, [_i32, _i32, _i32, ctypes.POINTER(_vp)]),
    'qh_attach': (_i32, [_i32, _i32, _i32, _vp, _vp, ctypes.POINTER(_vp)]),
    'qh_create_dry': (_i32, [_i32, _i32, ctypes.POINTER(_vp)]),
    'qh_create_host_mapped': (_i32, [_i32, _i32, _i32, ctypes.POINTER(_vp)]),
    'qh_host_ptr': (_i32, [_vp, ctypes.POINTER(_vp)]),
    'qh_destroy': (_i32, [_vp]),
    'qh_set_shard': (_i32, [_vp, _i32, _u64]),
    'qh_device_ptr': (_i32, [_vp, ctypes.POINTER(_vp)]),
    'qh_stream': (_i32, [_vp, ctypes.POINTER(_vp)]),
    'qh_nbits': (_i32, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    'qh_init_basis': (_i32, [_vp, _u64]),
    'qh_init_product': (_i32, [_vp, _i32, ctypes.POINTER(_i32), ctypes.POINTER(_vp), ctypes.POINTER(_u64)]),
    'qh_upload': (_i32, [_vp, _vp, _u64, _u64]),
    'qh_download': (_i32, [_vp, _vp, _u64, _u64]),
    'qh_apply1': (_i32, [_vp, _i32, _dp]),
    'qh_applyc': (_i32, [_vp, _i32, _i32, _dp]),
    'qh_apply_bits': (_i32, [_vp, _u64, _i32, _dp]),
    'qh_set_fusion': (_i32, [_vp, _i32]),
    'qh_set_relayout': (_i32, [_vp, _i32, ctypes.POINTER(_i32)]),
    'qh_apply_stream': (_i32, [_vp, _u64, ctypes.POINTER(ctypes.c_int32), _dp]),
    'qh_flush': (_i32, [_vp]),
    'qh_sync': (_i32, [_vp]),
    'qh_pending_gates': (_i32, [_vp, ctypes.POINTER(_u64)]),
    'qh_discard_pending': (_i32, [_vp]),
    'qh_remap_swap': (_i32, [_vp, _i32, _i32]),
    'qh_get_bitmap': (_i32, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    'qh_phys_to_logical': (_i32, [_vp, _u64, ctypes.POINTER(_u64)]),
    'qh_logical_to_phys': (_i32, [_vp, _u64, ctypes.POINTER(_u64)]),
    'qh_norm2': (_i32, [_vp, _dp]),
    'qh_amplitude': (_i32, [_vp, _u64, _dp]),
    'qh_argmax': (_i32, [_vp, ctypes.POINTER(_u64), _dp]),
    'qh_prob_bit': (_i32, [_vp, _i32, _dp]),
    'qh_prob_bit_value': (_i32, [_vp, _i32, _i32, _dp]),
    'qh_scale': (_i32, [_vp, ctypes.c_double, ctypes.c_double]),
    'qh_project_bit': (_i32, [_vp, _i32, _i32]),
    'qh_get_stats': (_i32, [_vp, ctypes.POINTER(QhStats)]),
    'qh_reset_stats': (_i32, [_vp]),
    'qh_timer_begin': (_i32, [_vp]),
    'qh_timer_end': (_i32, [_vp, ctypes.POINTER(ctypes.c_float)]),
    'qh_timer_lap': (_i32, [_vp]),
    'qh_timer_laps': (_i32, [_vp, ctypes.POINTER(ctypes.c_float), _i32, ctypes.POINTER(_i32)]),
    'qh_plan_json': (_i32, [_vp, ctypes.c_char_p, _u64, ctypes.POINTER(_u64)]),
    'qh_plan_export': (_i32, [_vp, _vp, _u64, ctypes.POINTER(_u64)]),
    'qh_host_apply1': (_i32, [_vp, _dp, _i32, _i32, _i32]),
    'qh_host_applyc': (_i32, [_vp, _dp, _i32, _i32, _i32, _i32]),
    'qh_host_release': (_i32, []),
}


class QhXStats(ctypes.Structure):
  _fields_ = [('exchanges', _u64), ('rounds', _u64), ('bytes_sent', _u64), ('slabs', _u64),
              ('sweeps_overlapped', _u64), ('span_ms', ctypes.c_double), ('rounds_packed', _u64),
              ('geometry_checks', _u64), ('comm_ranks', ctypes.c_uint32), ('comm_rank', ctypes.c_uint32)]

  def as_dict(self):
    return {k: getattr(self, k) for k, _ in self._fields_}


class QhXGeom(ctypes.Structure):
  """include/qcc_hip.h qh_xgeom: how the last exchange of a handle was cut (equal on every rank, or the exchange fails)."""
  _fields_ = [('signature', _u64), ('slab_mask', _u64), ('block_bits', _u64), ('rounds_per_slab', _u64), ('staging_bytes', _u64),
              ('slabs', ctypes.c_uint32), ('chunk_bits', ctypes.c_uint32), ('packed', ctypes.c_uint32), ('peers', ctypes.c_uint32),
              ('sweeps_before', ctypes.c_uint32), ('last_sweep_split', ctypes.c_uint32)]

  def as_dict(self):
    return {k: int(getattr(self, k)) for k, _ in self._fields_}


# one round of the host-staged transport (include/qcc_hip.h: qh_round_fn)
ROUND_FN = ctypes.CFUNCTYPE(ctypes.c_int, _vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(_vp),
                            ctypes.POINTER(_vp), _u64)

SIGNATURES.update({
    'qh_comm_unique_id': (_i32, [_vp]),
    'qh_comm_init': (_i32, [_vp, _i32, _i32, _vp]),
    'qh_comm_init_custom': (_i32, [_vp, _i32, _i32, ROUND_FN, _vp]),
    'qh_comm_init_dry': (_i32, [_vp, _i32, _i32]),
    'qh_exchange_geometry': (_i32, [_vp, ctypes.POINTER(QhXGeom)]),
    'qh_comm_destroy': (_i32, [_vp]),
    'qh_exchange_alltoall': (_i32, [_vp, _i32, _u64]),
    'qh_exchange_pair': (_i32, [_vp, _i32, _i32, _u64]),
    'qh_exchange_loopback': (_i32, [_vp, _i32, _u64]),
    'qh_exchange_wait': (_i32, [_vp]),
    'qh_exchange_stats': (_i32, [_vp, ctypes.POINTER(QhXStats)]),
    'qh_comm_allreduce_sum': (_i32, [_vp, _dp, _i32]),
})


class QhError(RuntimeError):
  def __init__(self, code, msg):
    super().__init__(f'qcc_hip error {code}: {msg}')
    self.code = code


def build(force=False, verbose=False):
  """Compile the HIP engine for gfx950 in-tree (qcc_amd/libqcc_hip.so)."""
  deps = SOURCES + [HEADER] + HEADERS
  if (not force and os.path.exists(LIB_PATH)
      and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps)):
    return LIB_PATH
  cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC',
         '-o', LIB_PATH, SOURCES[0], SOURCES[-1]]  # engine.hip (+ its headers) and the libq facade
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd, cwd=ROOT)
  return LIB_PATH


_lib = None


def _preload_torch_runtime():
  """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own
  libamdhip64.so; if ours (/opt/rocm) is mapped first, torch afterwards sees no
  GPU (measured on the MI355X box: tools/order_probe.py).  Whenever this process
  is going to use torch.distributed (multi-GPU runs launched by torchrun set
  WORLD_SIZE), import torch BEFORE dlopen()ing the engine so that the engine's
  DT_NEEDED libamdhip64.so.7 resolves to the copy torch already mapped."""
  import sys
  if 'torch' in sys.modules:
    return
  if os.environ.get('WORLD_SIZE') or os.environ.get('QCC_PRELOAD_TORCH') == '1':
    import torch  # noqa: F401  pylint: disable=import-outside-toplevel,unused-import


def load():
  """Load libqcc_hip.so and bind every declared symbol.  No fallback."""
  global _lib
  if _lib is not None:
    return _lib
  _preload_torch_runtime()
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
        '(hipcc --offload-arch=gfx950).  qcc_amd has no CPU fallback.')
  lib = ctypes.CDLL(LIB_PATH)
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)  # AttributeError if the symbol is not exported
    fn.restype, fn.argtypes = res, args
  _lib = lib
  return lib


def check(rc):
  if rc != QH_OK:
    raise QhError(rc, load().qh_last_error().decode(errors='replace'))


def device_count():
  n = _i32(0)
  load().qh_device_count(ctypes.byref(n))
  return n.value
