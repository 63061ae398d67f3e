"""Import alias: the package directory is named ``diff-vits_amd`` (not a valid
Python identifier), so ``import diff_vits_amd`` loads it from here."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diff-vits_amd")
_spec = importlib.util.spec_from_file_location(
    "diff_vits_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["diff_vits_amd"] = _mod
_spec.loader.exec_module(_mod)
