"""`waveforms` — import alias of :mod:`waveforms_amd`.

Scripts written against mcdiarmid/waveforms (``from waveforms.cpm.modulate import
cpm_modulate`` ...) run unchanged: every ``waveforms.x.y`` import is served by the
module object ``waveforms_amd.x.y`` (one module, two names — stateful module globals
such as ``waveforms.noise.DEFAULT_RNG`` are therefore shared).
"""
import importlib
import importlib.abc
import importlib.util
import sys

import waveforms_amd

_SRC, _DST = "waveforms_amd", "waveforms"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)

    def exec_module(self, module):  # already executed under its real name
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _DST and not fullname.startswith(_DST + "."):
            return None
        real = _SRC + fullname[len(_DST):]
        try:
            real_spec = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            return None
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real),
                                               is_package=real_spec.submodule_search_locations is not None)
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__version__ = waveforms_amd.__version__
__path__ = []  # submodules come from the finder above, never from this directory
