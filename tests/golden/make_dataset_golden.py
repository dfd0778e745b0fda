"""Generates tests/golden/datasets.npz by running the REAL reference pair-dataset loaders (imported from /root/reference through
oracle/ref_harness.py) over the synthetic benchmark trees of tests/util.py, with fixed np.random / random seeds.

Build-container only (needs /root/reference).  The one shim beyond ref_harness: the reference calls `torch.load(path)` on pickled
numpy fragments, which torch >= 2.6 refuses by default (weights_only=True); the call is given weights_only=False here.
usage: python tests/golden/make_dataset_golden.py
"""
import functools
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ref_harness  # noqa: E402
from util import DATASET_CASES, make_dataset_trees, run_dataset_case  # noqa: E402


def main():
    ref_harness.setup()
    torch.load = functools.partial(torch.load, weights_only=False)
    from geotransformer.datasets.registration.kitti.dataset import OdometryKittiPairDataset
    from geotransformer.datasets.registration.modelnet.dataset import ModelNetPairDataset
    from geotransformer.datasets.registration.threedmatch.dataset import ThreeDMatchPairDataset
    classes = {c.__name__: c for c in (ThreeDMatchPairDataset, OdometryKittiPairDataset, ModelNetPairDataset)}
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        roots = make_dataset_trees(tmp)
        for name, cls_name, tree, kwargs, seeds, indices in DATASET_CASES:
            flat = run_dataset_case(classes[cls_name], roots[tree], kwargs, seeds, indices)
            for key, value in flat.items():
                out[f'{name}/{key}'] = value
            print(f'{name}: {len(flat)} entries, len(dataset) = {int(flat["len"])}')
    path = os.path.join(ROOT, 'tests', 'golden', 'datasets.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
