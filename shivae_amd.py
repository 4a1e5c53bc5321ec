"""Import shim: the package directory is ``shi-vae_amd/`` (the name the project brief fixes), which
is not a valid Python identifier.  ``import shivae_amd`` loads that directory as the package
``shivae_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shi-vae_amd")
_spec = importlib.util.spec_from_file_location(
    "shivae_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["shivae_amd"] = _mod
_spec.loader.exec_module(_mod)
