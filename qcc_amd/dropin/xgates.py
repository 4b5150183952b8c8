"""``xgates``: the second name the reference's extension is importable under (xgates.cc:165-175 exports both
PyInit_xgates and PyInit_libxgates).  One implementation: libxgates.py, next to this file."""
try:                                            # imported as qcc_amd.dropin.xgates (a package module)
    from .libxgates import apply1, applyc       # noqa: F401
except ImportError:                             # imported as top-level `xgates` with this directory on sys.path (the reference's way)
    from libxgates import apply1, applyc        # noqa: F401
