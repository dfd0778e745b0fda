"""Seeded synthetic registration pairs in the shapes BASELINE.json names.

A scene is a union of random planar / box / sphere *surfaces* inside a cube, with continuous jitter
(no quantisation => neighbour lists are tie-free) and voxel-thinned to one point per ``voxel`` cell so
the local density matches real 3DMatch/KITTI data (~30 neighbours at r = 2.5 voxel).  ``ref`` and
``src`` are overlapping subsets; ``src`` is moved by a random rigid transform, and
``transform`` maps src -> ref (the convention of the reference datasets,
geotransformer/datasets/registration/threedmatch/dataset.py:131-135).

numpy only: this file is shared by bench.py, the tests and the golden-vector generator.
"""
import numpy as np

CONFIGS = {
    # name: scene extent (m), voxel, points per cloud, stages, init radius, neighbour limits
    'modelnet': dict(extent=(1.0, 1.0, 1.0), voxel=0.05, n_points=1024, num_stages=3, radius=0.125,
                     limits=[24, 24, 24]),
    '3dmatch': dict(extent=(3.0, 3.0, 3.0), voxel=0.025, n_points=20000, num_stages=4, radius=0.0625,
                    limits=[38, 36, 36, 38]),
    'kitti': dict(extent=(120.0, 120.0, 6.0), voxel=0.3, n_points=120000, num_stages=5, radius=1.275,
                  limits=[40, 40, 40, 40, 40]),
}


def _random_rotation(rng, max_angle=np.pi):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    angle = rng.uniform(-max_angle, max_angle)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)


def _layout(rng):
    """Random scene layout in unit-cube coordinates (scaled later): walls, boxes, spheres."""
    boxes = []
    for _ in range(10):
        size = rng.uniform(0.08, 0.35, size=3)
        lo = rng.uniform(0, 1, size=3) * (1.0 - size)
        boxes.append((lo, size))
    spheres = []
    for _ in range(8):
        spheres.append((rng.uniform(0.1, 0.9, size=3), rng.uniform(0.05, 0.2)))
    return boxes, spheres


def _surface_samples(rng, layout, ext, n):
    """~n points on the layout's surfaces inside the box [0, ext], area-weighted."""
    boxes, spheres = layout
    ext = np.asarray(ext, dtype=np.float64)
    prims = []  # (area, sampler)
    for axis in range(3):
        a, b = [c for c in range(3) if c != axis]
        area = ext[a] * ext[b]
        for side in (0.02, 0.98):
            def wall(k, axis=axis, side=side):
                pts = rng.uniform(0, 1, size=(k, 3)) * ext
                pts[:, axis] = side * ext[axis]
                return pts
            # only the floor and two walls: a scan never sees the whole room shell
            if (axis, side) in ((2, 0.02), (0, 0.02), (1, 0.98)):
                prims.append((area, wall))
    for lo, size in boxes:
        lo_s, size_s = lo * ext, size * ext
        area = 2 * (size_s[0] * size_s[1] + size_s[1] * size_s[2] + size_s[0] * size_s[2])

        def box(k, lo_s=lo_s, size_s=size_s):
            face_area = np.array([size_s[1] * size_s[2], size_s[0] * size_s[2], size_s[0] * size_s[1]] * 2)
            face = rng.choice(6, size=k, p=face_area / face_area.sum())
            pts = lo_s + rng.uniform(0, 1, size=(k, 3)) * size_s
            ax = face % 3
            pts[np.arange(k), ax] = lo_s[ax] + (face // 3) * size_s[ax]
            return pts
        prims.append((area, box))
    for c, r in spheres:
        rad = r * ext.min()

        def sph(k, c=c, rad=rad):
            d = rng.normal(size=(k, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            return np.clip(c * ext + rad * d, 0, ext)
        prims.append((4 * np.pi * rad * rad, sph))
    areas = np.array([p[0] for p in prims])
    counts = np.maximum((areas / areas.sum() * n).astype(np.int64), 16)
    return np.concatenate([f(int(k)) for (_, f), k in zip(prims, counts)], axis=0)


def _voxel_thin(rng, pts, voxel):
    """Keep one (jittered) point per voxel cell: continuous coordinates, real-data-like density."""
    keys = np.floor(pts / voxel).astype(np.int64)
    _, first = np.unique(keys, axis=0, return_index=True)
    pts = pts[np.sort(first)]
    centres = (np.floor(pts / voxel) + 0.5) * voxel
    return centres + rng.uniform(-0.4, 0.4, size=pts.shape) * voxel


def make_scene(seed, extent, voxel, n_target):
    """A jittered, voxel-thinned, fully dense surface cloud with ``n_target`` points (float64).

    The scene is rescaled (surface area ~ scale^2) until the number of occupied voxels matches
    ``n_target``; ``extent`` only fixes the aspect ratio.  Returns (points, rng, extent_used).
    """
    rng = np.random.default_rng(seed)
    layout = _layout(rng)
    shape = np.asarray(extent, dtype=np.float64)
    shape = shape / shape.max()
    scale = np.sqrt(n_target * voxel * voxel / 4.0)  # first guess: ~4 unit areas of surface
    pts = None
    for it in range(6):
        ext = shape * scale
        srng = np.random.default_rng([seed, 1000 + it])
        raw = _surface_samples(srng, layout, ext, int(n_target * 10))
        pts = _voxel_thin(srng, raw, voxel)
        ratio = pts.shape[0] / float(n_target)
        if 1.0 <= ratio <= 1.04:
            break
        scale *= np.sqrt(1.02 / ratio)
    if pts.shape[0] > n_target:
        sel = rng.permutation(pts.shape[0])[:n_target]
        pts = pts[np.sort(sel)]
    return pts, rng, shape * scale


def make_pair(seed=0, config='3dmatch', n_points=None, overlap=0.6, extent=None, voxel=None):
    """Return the item dict the reference datasets produce (threedmatch/dataset.py:131-135).

    ref = subset A of the scene, src = R*subset B + t, |A ∩ B| / |A| ~= overlap.
    ``transform`` (4,4) maps src onto ref.
    """
    cfg = CONFIGS[config]
    extent = cfg['extent'] if extent is None else extent
    voxel = cfg['voxel'] if voxel is None else voxel
    n_points = cfg['n_points'] if n_points is None else n_points
    # scene needs n*(2-overlap) points so that both subsets have n points with the requested overlap
    n_scene = int(np.ceil(n_points * (2.0 - overlap)))
    scene, rng, extent = make_scene(seed, extent, voxel, n_scene)
    n_scene = scene.shape[0]
    n_points = min(n_points, int(n_scene / (2.0 - overlap)))
    # split along a random direction so that overlap is spatially coherent (like two scans of one room)
    direction = rng.normal(size=3)
    direction /= np.linalg.norm(direction)
    order = np.argsort(scene @ direction, kind='stable')
    ref_idx = np.sort(order[:n_points])
    src_idx = np.sort(order[n_scene - n_points:])
    ref = scene[ref_idx]
    src_in_ref_frame = scene[src_idx]
    R = _random_rotation(rng)
    centre = src_in_ref_frame.mean(axis=0)
    t = rng.uniform(-0.5, 0.5, size=3) * np.asarray(extent).min()
    # src = R (p - c) + c + t  => p = R^T (src - c - t) + c
    src = (src_in_ref_frame - centre) @ R.T + centre + t
    Rinv = R.T
    tinv = centre - Rinv @ (centre + t)
    transform = np.eye(4)
    transform[:3, :3] = Rinv
    transform[:3, 3] = tinv
    # independent sensor noise on src so that the two clouds do not share identical samples
    src = src + rng.normal(scale=0.05 * voxel, size=src.shape)
    return {
        'ref_points': ref.astype(np.float32),
        'src_points': src.astype(np.float32),
        'ref_feats': np.ones((ref.shape[0], 1), dtype=np.float32),
        'src_feats': np.ones((src.shape[0], 1), dtype=np.float32),
        'transform': transform.astype(np.float32),
    }


def quantise(points, step=0.001):
    """Millimetre-quantised copy (tie-heavy, like the reference's data/demo clouds)."""
    return (np.round(points / step) * step).astype(np.float32)
