"""Import alias: the package directory is `online-3d-bpp-pct_amd/` (not a valid identifier),
so `import pct_amd` loads it through importlib."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("online-3d-bpp-pct_amd")
globals().update({k: getattr(_pkg, k) for k in _pkg.__all__})
_lib = _pkg._lib
__all__ = list(_pkg.__all__)
