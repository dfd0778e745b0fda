"""CPU, build container only (needs /root/reference): the reference's OWN experiments/*/{config,backbone,model}.py executed
unchanged with `geotransformer` resolving to the replacement package (compat/ on PYTHONPATH -> geotransformer_amd.compat) --
SURVEY.md section 2 #14 / section 8b boundary 2.  For each of the three experiments the reference's `create_model(make_cfg())`
must construct on the replacement modules, and under the reference's seeds its state_dict must be the reference model's:
same keys, same shapes, same bytes (so released checkpoints load with strict=True and seeded runs start from identical
weights).  Absent third-party imports of those scripts (easydict, IPython) are stubbed exactly as oracle/ref_harness.py does.
The FORWARD of those scripts runs on the GPU box from byte-for-byte snapshots (tests/golden/reference_scripts.npz ->
tests/test_reference_forward_gpu.py); here the snapshots are checked against the live tree."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

CHILD = r'''
import hashlib, json, logging, os, sys, types, importlib.util
import numpy as np, torch
sys.dont_write_bytecode = True
exp_dir, mode = sys.argv[1], sys.argv[2]

class AttrDict(dict):
    __getattr__ = lambda self, k: self[k] if k in self else (_ for _ in ()).throw(AttributeError(k))
    __setattr__ = dict.__setitem__
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m
stub('IPython', embed=lambda *a, **k: None)
stub('easydict', EasyDict=AttrDict)

if mode == 'replacement':
    import geotransformer                      # compat/geotransformer -> geotransformer_amd
    assert geotransformer.__name__ == 'geotransformer_amd', geotransformer.__name__
    import geotransformer.utils.common as common
else:
    sys.path.insert(0, os.path.join(%(root)r))
    from oracle import ref_harness as rh
    rh.setup()
    import geotransformer
    import geotransformer.utils.common as common
    assert geotransformer.__file__.startswith('/root/reference')
common.ensure_dir = lambda p: None             # config.py creates output directories at import time: not under /root/reference

mods = {}
for short in ('config', 'backbone', 'model'):
    spec = importlib.util.spec_from_file_location(short, os.path.join(exp_dir, short + '.py'))
    m = importlib.util.module_from_spec(spec); sys.modules[short] = m; spec.loader.exec_module(m); mods[short] = m
cfg = mods['config'].make_cfg()
torch.manual_seed(cfg.seed); np.random.seed(cfg.seed)
model = mods['model'].create_model(cfg)
sd = model.state_dict()
h = hashlib.sha256()
for k in sorted(sd):
    h.update(k.encode()); h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
classes = sorted({type(m).__module__.split('.')[0] for m in model.modules()})
print('RESULT ' + json.dumps({'keys': {k: list(v.shape) for k, v in sd.items()}, 'sha256': h.hexdigest(), 'module_roots': classes,
                              'model_file': mods['model'].__file__}))
if mode == 'replacement':
    # a checkpoint written by the reference model loads strictly, and vice versa
    blob = sys.argv[3]
    if os.path.exists(blob):
        model.load_state_dict(torch.load(blob), strict=True)
        print('STRICT-LOAD-OK')
else:
    torch.save(sd, sys.argv[3])
'''


def _run(exp_dir, mode, blob):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env['PYTHONPATH'] = os.path.join(ROOT, 'compat') if mode == 'replacement' else ROOT
    res = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}, exp_dir, mode, blob], env=env, cwd='/tmp', capture_output=True,
                         text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith('RESULT ')]
    assert res.returncode == 0 and lines, res.stdout[-2000:] + res.stderr[-4000:]
    return json.loads(lines[0][7:]), res.stdout


@pytest.mark.parametrize('exp', ['geotransformer.3dmatch.stage4.gse.k3.max.oacl.stage2.sinkhorn',
                                 'geotransformer.kitti.stage5.gse.k3.max.oacl.stage2.sinkhorn',
                                 'geotransformer.modelnet.rpmnet.stage4.gse.k3.max.oacl.stage2.sinkhorn'])
def test_reference_model_py_constructs_on_the_replacement(exp, tmp_path):
    exp_dir = os.path.join(REF, 'experiments', exp)
    if not os.path.isdir(exp_dir):
        pytest.skip('/root/reference not present (GPU box): the build container runs this')
    blob = str(tmp_path / 'reference_state_dict.pt')
    ref, _ = _run(exp_dir, 'reference', blob)
    got, out = _run(exp_dir, 'replacement', blob)
    assert got['model_file'].startswith(REF)                       # the reference's own model.py ...
    assert got['module_roots'] == ['geotransformer_amd', 'model', 'torch'] or got['module_roots'] == ['backbone', 'geotransformer_amd', 'model', 'torch'], got['module_roots']
    assert 'geotransformer' not in got['module_roots']             # ... built only from replacement modules
    assert list(got['keys']) == list(ref['keys'])                  # same keys in the same order
    assert got['keys'] == ref['keys']                              # same shapes
    assert got['sha256'] == ref['sha256']                          # same seeded weights, byte for byte
    assert 'STRICT-LOAD-OK' in out
    if '3dmatch' in exp:  # and it is the state_dict the demo golden was produced with
        from util import load_demo_golden
        assert got['sha256'] == str(load_demo_golden()['sd/sha256'])


def test_reference_scripts_fixture_is_the_live_tree():
    """tests/golden/reference_scripts.npz (what tests/test_reference_forward_gpu.py executes on the GPU box) holds the reference's
    scripts byte for byte."""
    import hashlib

    import numpy as np
    if not os.path.isdir(os.path.join(REF, 'experiments')):
        pytest.skip('/root/reference not present (GPU box): the build container runs this')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_scripts.npz'))
    for short in ('3dmatch', 'kitti', 'modelnet'):
        exp_dir = os.path.join(REF, 'experiments', str(g[f'{short}/dirname']))
        for name in ('config.py', 'backbone.py', 'model.py'):
            with open(os.path.join(exp_dir, name), 'rb') as f:
                blob = f.read()
            assert g[f'{short}/{name}'].tobytes() == blob, (short, name)
            assert str(g[f'{short}/{name}/sha256']) == hashlib.sha256(blob).hexdigest()


def test_alias_package_keeps_module_identity_and_probes_return_none():
    """aliasing must not overwrite geotransformer_amd.X.__spec__, and find_spec on a name the replacement lacks is None."""
    code = r'''
import importlib.util
import geotransformer, geotransformer.utils.common as a, geotransformer_amd.utils.common as b
assert a is b and b.__spec__.name == 'geotransformer_amd.utils.common' and b.__package__ == 'geotransformer_amd.utils'
assert importlib.util.find_spec('geotransformer.engine') is None
try:
    import geotransformer.engine
    raise SystemExit('imported a module that does not exist')
except ModuleNotFoundError as exc:
    assert 'geotransformer.engine' in str(exc)
from geotransformer.modules.kpconv import ConvBlock
assert ConvBlock.__module__ == 'geotransformer_amd.modules.kpconv.modules'
print('ALIAS-OK')
'''
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=os.path.join(ROOT, 'compat'))
    res = subprocess.run([sys.executable, '-c', code], env=env, cwd='/tmp', capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'ALIAS-OK' in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
