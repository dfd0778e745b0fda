"""`import geotransformer` served by this package: the reference's import paths as aliases of geotransformer_amd's modules.

The reference's experiment scripts import `geotransformer.modules.*`, `geotransformer.utils.*`, `geotransformer.ext`,
`geotransformer.datasets.*` (experiments/*/model.py:6-16, backbone.py:5, config.py:7, dataset.py:1-6, demo.py:6-9).  After
`install()` -- or with `compat/` on PYTHONPATH, whose `geotransformer/__init__.py` calls it -- every `geotransformer.X.Y` import
resolves to THE SAME module object as `geotransformer_amd.X.Y`, so those scripts run unchanged on the HIP hot path.  A name the
replacement does not provide (the training engine, open3d visualisation, losses: outside the hot-path scope) raises the usual
ModuleNotFoundError for `geotransformer.<name>`; `importlib.util.find_spec` on it returns None.
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
        # the already-imported geotransformer_amd module object itself; importlib then stamps the ALIAS spec onto it
        # (module.__spec__ / __loader__ / __package__), so its own import-system identity is saved here ...
        self.saved = {k: getattr(self.module, k) for k in ('__spec__', '__loader__', '__package__') if hasattr(self.module, k)}
        return self.module

    def exec_module(self, module):
        # ... and restored here: geotransformer_amd.X.__spec__.name stays 'geotransformer_amd.X' (importlib.reload, pickling of
        # classes by module name and find_spec on the real name keep working)
        for k, v in self.saved.items():
            setattr(module, k, v)


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != ALIAS and not fullname.startswith(ALIAS + '.'):
            return None
        real = TARGET + fullname[len(ALIAS):]
        try:
            module = importlib.import_module(real)
        except ModuleNotFoundError as exc:
            if exc.name == real:   # the replacement has no such module: "not found" for finders / find_spec probes (-> None);
                return None        # a plain `import` then raises ModuleNotFoundError for the alias name as usual
            raise                  # a missing dependency INSIDE an existing module is a real error
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
