"""``xgates``: the second name the reference's extension is importable under (xgates.cc:165-175 exports both
PyInit_xgates and PyInit_libxgates).  One implementation: libxgates.py, next to this file."""
from libxgates import apply1, applyc  # noqa: F401  (this directory is on sys.path whenever either module is importable)
