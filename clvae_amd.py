"""Importable alias for the package directory ``classifying-vae-lstm_amd/``.

The directory name is fixed by the build contract and is not a valid Python
identifier, so ``import clvae_amd`` loads that directory as a package under this
name (``clvae_amd.cl_vae.model``, ``clvae_amd.utils.pianoroll`` ...).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "classifying-vae-lstm_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
