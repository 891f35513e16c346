"""Importable alias for the package directory `global-lvba_amd/` (a hyphen is not a valid identifier)."""
import importlib
import sys

_pkg = importlib.import_module("global-lvba_amd")
sys.modules[__name__] = _pkg
