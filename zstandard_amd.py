"""Importable alias of the ``python-zstandard_amd`` package directory (a hyphen is not valid in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("python-zstandard_amd")
sys.modules[__name__] = _pkg
