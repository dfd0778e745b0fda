"""`import geotransformer` served by this package: the reference's import paths as aliases of geotransformer_amd's modules.

The reference's experiment scripts import `geotransformer.modules.*`, `geotransformer.utils.*`, `geotransformer.ext`,
`geotransformer.datasets.*` (experiments/*/model.py:6-16, backbone.py:5, config.py:7, dataset.py:1-6, demo.py:6-9).  After
`install()` -- or with `compat/` on PYTHONPATH, whose `geotransformer/__init__.py` calls it -- every `geotransformer.X.Y` import
resolves to THE SAME module object as `geotransformer_amd.X.Y`, so those scripts run unchanged on the HIP hot path.  A name the
replacement does not provide (the training engine, open3d visualisation, losses: outside the hot-path scope) raises the usual
ModuleNotFoundError naming the missing `geotransformer_amd` module.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

ALIAS = 'geotransformer'
TARGET = 'geotransformer_amd'


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self.module = module

    def create_module(self, spec):
        return self.module  # the already-imported geotransformer_amd module object itself

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != ALIAS and not fullname.startswith(ALIAS + '.'):
            return None
        real = TARGET + fullname[len(ALIAS):]
        module = importlib.import_module(real)  # ModuleNotFoundError names what is missing
        spec = importlib.machinery.ModuleSpec(fullname, _AliasLoader(module), is_package=hasattr(module, '__path__'))
        spec.submodule_search_locations = getattr(module, '__path__', None)
        return spec


_finder = None


def install():
    """Idempotent.  Refuses to shadow a real `geotransformer` package that is already imported."""
    global _finder
    if _finder is not None:
        return
    present = sys.modules.get(ALIAS)
    if present is not None and getattr(present, '__name__', ALIAS) != TARGET and not getattr(present, '_geotr_alias_stub', False):
        raise RuntimeError('a different `geotransformer` package is already imported; geotransformer_amd.compat.install() must run first')
    _finder = _AliasFinder()
    sys.meta_path.insert(0, _finder)
    root = importlib.import_module(TARGET)
    sys.modules[ALIAS] = root  # the package object itself; submodules resolve through the finder
    importlib.invalidate_caches()


def uninstall():
    """Remove the aliases again (tests that also import the real reference in the same process)."""
    global _finder
    if _finder is None:
        return
    sys.meta_path.remove(_finder)
    _finder = None
    for name in [n for n in sys.modules if n == ALIAS or n.startswith(ALIAS + '.')]:
        del sys.modules[name]
