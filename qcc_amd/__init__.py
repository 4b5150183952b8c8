"""qcc_amd -- MI355X-native gate-application engine behind the qcc API.

Hot path: hand-written HIP kernels (qcc_amd/csrc) behind the C-ABI declared in
include/qcc_hip.h, loaded with ctypes (qcc_amd.native).  There is NO CPU
fallback in this package: without the built library or without a GPU the
device classes raise.
"""
from qcc_amd import native  # noqa: F401

__all__ = ['native']
__version__ = '0.1.0'
