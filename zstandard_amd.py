"""Importable alias of the ``python-zstandard_amd`` package directory (a hyphen is not valid in an import statement).

``import zstandard_amd`` and ``import zstandard_amd.device`` / ``.sharded`` / ... resolve to the ONE set of module objects of the real
package (a second import of the extension under another name would give a second ZstdError class that ``except`` clauses miss).
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL, _ALIAS = "python-zstandard_amd", __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(name, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
