"""Importable alias of the package directory `vulkan-rtiow_amd/` (hyphenated names
cannot appear in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vulkan-rtiow_amd")
sys.modules[__name__] = _pkg
