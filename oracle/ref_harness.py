"""TEST INFRASTRUCTURE ONLY -- imports the REAL reference Python (/root/reference) on CPU.

Used only in the build container (where /root/reference exists) to (a) generate the golden vectors
under tests/golden/ and (b) validate oracle/model_oracle.py.  Nothing is written under /root/reference:
bytecode writing is disabled and config.py's ensure_dir is neutralised.

Shims (the reference cannot run on CPU unmodified):
  * stub modules for absent, import-time-only dependencies: IPython, ipdb, coloredlogs, easydict, open3d
    (open3d.io.read_point_cloud -> minimal binary-LE PLY reader for the 15-point kernel disposition);
  * geotransformer.ext  -> the real reference C++ cores through oracle/_ref/libgeoref.so;
  * Tensor.cuda()/Module.cuda() -> contiguous()/identity so `.cuda()` calls inside modules are no-ops.
"""
import logging
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = '/root/reference'
EXPERIMENTS = {
    '3dmatch': 'geotransformer.3dmatch.stage4.gse.k3.max.oacl.stage2.sinkhorn',
    'kitti': 'geotransformer.kitti.stage5.gse.k3.max.oacl.stage2.sinkhorn',
    'modelnet': 'geotransformer.modelnet.rpmnet.stage4.gse.k3.max.oacl.stage2.sinkhorn',
}
_state = {}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'geotransformer'))


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def _read_ply(path):
    with open(path, 'rb') as f:
        header = b''
        while not header.endswith(b'end_header\n'):
            header += f.readline()
        n = int([l for l in header.decode().split('\n') if l.startswith('element vertex')][0].split()[-1])
        h = header.decode()
        dtype = '<f8' if ('property double x' in h or 'property float64 x' in h) else '<f4'
        pts = np.frombuffer(f.read(), dtype=dtype, count=3 * n).reshape(n, 3).astype(np.float64)
    return types.SimpleNamespace(points=pts)


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod('IPython', embed=lambda *a, **k: None)
    mod('ipdb', set_trace=lambda *a, **k: None)
    mod('coloredlogs', ColoredFormatter=logging.Formatter)
    mod('easydict', EasyDict=_AttrDict)
    o3d = mod('open3d')
    o3d.io = mod('open3d.io', read_point_cloud=_read_ply)
    o3d.geometry = mod('open3d.geometry')
    o3d.utility = mod('open3d.utility')


class _RefExt:
    """geotransformer.ext backed by the real reference cores (oracle/_ref/libgeoref.so)."""

    def __init__(self):
        from oracle import neighbors
        self.lib = neighbors.reference()
        assert self.lib is not None, 'oracle/_ref/libgeoref.so missing: make -C oracle ref'

    def radius_neighbors(self, q_points, s_points, q_lengths, s_lengths, radius):
        out = self.lib.radius_neighbors(q_points.numpy(), s_points.numpy(), q_lengths.numpy(), s_lengths.numpy(), radius)
        return torch.from_numpy(out)

    def grid_subsampling(self, points, lengths, voxel_size):
        pts, lens = self.lib.grid_subsampling(points.numpy(), lengths.numpy(), voxel_size)
        return [torch.from_numpy(pts), torch.from_numpy(lens)]


def setup():
    """Make `import geotransformer` resolve to the reference, CPU-runnable.  Idempotent."""
    if _state.get('ready'):
        return
    assert available(), '/root/reference is not present'
    sys.dont_write_bytecode = True
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ext = _RefExt()
    ext_mod = types.ModuleType('geotransformer.ext')
    ext_mod.radius_neighbors = ext.radius_neighbors
    ext_mod.grid_subsampling = ext.grid_subsampling
    sys.modules['geotransformer.ext'] = ext_mod
    import geotransformer  # noqa: F401
    geotransformer.ext = ext_mod
    import geotransformer.utils.common as common
    common.ensure_dir = lambda p: None
    torch.Tensor.cuda = lambda t, *a, **k: t.contiguous()
    torch.nn.Module.cuda = lambda m, *a, **k: m
    _state['ready'] = True


def load_experiment(name):
    """Returns (config module, model module) of an experiment directory, imported under unique names."""
    setup()
    import importlib.util
    exp_dir = os.path.join(REF_ROOT, 'experiments', EXPERIMENTS[name])
    mods = {}
    # model.py does `from backbone import KPConvFPN`; config.py is imported by name too
    for short in ('config', 'backbone', 'model'):
        spec = importlib.util.spec_from_file_location(short, os.path.join(exp_dir, short + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[short] = m
        spec.loader.exec_module(m)
        mods[short] = m
    for short in ('config', 'backbone', 'model'):
        sys.modules.pop(short, None)
    return mods['config'], mods['model']


def build_model(name='3dmatch', overrides=None, seed=7351):
    """create_model(make_cfg()) with the reference's seed convention (config.py:13; kernel points use np.random)."""
    config, model_mod = load_experiment(name)
    cfg = config.make_cfg()
    for path, value in (overrides or {}).items():
        node = cfg
        keys = path.split('.')
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = model_mod.create_model(cfg).eval()
    return cfg, model


def collate(item, cfg, neighbor_limits):
    setup()
    from geotransformer.utils.data import registration_collate_fn_stack_mode
    from geotransformer.utils.torch import to_cuda
    data = registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                              cfg.backbone.init_radius, neighbor_limits)
    return to_cuda(data)  # with the .cuda shim this only makes the sliced neighbour tensors contiguous


def load_evaluator(name):
    """The experiment's own `Evaluator` class (experiments/<exp>/loss.py) and its config, CPU-runnable."""
    setup()
    import importlib.util
    exp_dir = os.path.join(REF_ROOT, 'experiments', EXPERIMENTS[name])
    out = {}
    for short in ('config', 'loss'):
        spec = importlib.util.spec_from_file_location(f'_ref_{name}_{short}', os.path.join(exp_dir, short + '.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        out[short] = m
    return out['config'].make_cfg(), out['loss'].Evaluator
