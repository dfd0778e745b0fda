"""Alias package: put this directory's parent (`<repo>/compat`) on PYTHONPATH and `import geotransformer...` resolves to
geotransformer_amd (the MI355X-native hot path) -- the reference's experiments/*/{model,backbone,config,dataset}.py and demo.py
then import unchanged.  See geotransformer_amd/compat.py."""
import os
import sys

_stub = sys.modules[__name__]
_stub._geotr_alias_stub = True
_repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _repo not in sys.path:
    sys.path.insert(0, _repo)
from geotransformer_amd import compat as _compat  # noqa: E402

_compat.install()  # replaces sys.modules['geotransformer'] by the geotransformer_amd package object
