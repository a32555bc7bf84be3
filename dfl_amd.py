"""Import shim: ``import dfl_amd`` loads the package in ``deepfluorolabeling-ipcai2020_amd/`` (a directory name that
is not a valid Python identifier) under the module name ``dfl_amd``; ``dfl_amd.unet`` etc. resolve inside it."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'deepfluorolabeling-ipcai2020_amd')
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_pkg_dir, '__init__.py'),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
