"""qcc_amd.lib -- host-side mirror of the reference's ``src.lib`` package for the
gate-application hot path.

Same module names, class names, function names and argument meaning as
/root/reference/src/lib/{tensor,helper,state,ops,ir,circuit,bell}.py, so code
written against ``from src.lib import circuit, ops, state`` runs against this
package after :func:`install_as_src_lib` (or with ``qcc_amd/compat`` on
``sys.path``).  What differs is underneath: ``circuit.qc`` keeps the amplitude
vector in MI355X HBM behind the C-ABI (qcc_amd.device) instead of calling the
``libxgates`` CPU extension on a NumPy buffer.
"""
import importlib
import sys

_MODULES = ('tensor', 'helper', 'state', 'ops', 'ir', 'circuit', 'bell')


def install_as_src_lib():
    """Register this package as ``src.lib`` (and ``src``) in sys.modules."""
    import types
    if 'src' not in sys.modules:
        pkg = types.ModuleType('src')
        pkg.__path__ = []
        sys.modules['src'] = pkg
    me = sys.modules[__name__]
    sys.modules['src.lib'] = me
    setattr(sys.modules['src'], 'lib', me)
    for name in _MODULES:
        mod = importlib.import_module(f'{__name__}.{name}')
        sys.modules[f'src.lib.{name}'] = mod
        setattr(me, name, mod)
    return me
