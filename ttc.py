"""`import ttc` == the package in `sentinel-tree-cover_amd/` (hyphenated directory name).

`ttc.x` and `sentinel-tree-cover_amd.x` resolve to the SAME module objects (an alias
finder, not a second import), so the native library is loaded once.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "sentinel-tree-cover_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, mod):
        self._mod = mod

    def create_module(self, spec):
        return self._mod

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if not fullname.startswith("ttc."):
            return None
        mod = importlib.import_module(_REAL + fullname[3:])
        return importlib.util.spec_from_loader(fullname, _AliasLoader(mod))


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
