"""Generates tests/golden/reference_scripts.npz: a TEST FIXTURE (like the .npz golden vectors next to it), not product code.

SURVEY.md section 2 #14 / section 8b boundary 2 says the reference's own experiments/*/{config,backbone,model}.py "must run
unchanged" on the replacement modules.  Their forward needs a GPU, and the GPU box has no /root/reference -- so the three scripts
of each experiment travel as DATA: byte-for-byte snapshots (with the reference's MIT LICENSE) packed into one compressed archive.
tests/test_reference_forward_gpu.py unpacks them into a temporary directory, imports them with `geotransformer` resolving to
geotransformer_amd (compat/), and runs `model(data_dict)`.  Nothing in geotransformer_amd/ reads this archive; it is regenerated
from /root/reference by this script (build container only) and its SHA-256s are checked against the live tree by
tests/test_reference_scripts.py whenever /root/reference is present.

Run from the repo root in the build container:   python tests/golden/make_reference_scripts_fixture.py
"""
import hashlib
import os

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
EXPERIMENTS = {
    '3dmatch': 'geotransformer.3dmatch.stage4.gse.k3.max.oacl.stage2.sinkhorn',
    'kitti': 'geotransformer.kitti.stage5.gse.k3.max.oacl.stage2.sinkhorn',
    'modelnet': 'geotransformer.modelnet.rpmnet.stage4.gse.k3.max.oacl.stage2.sinkhorn',
}
FILES = ('config.py', 'backbone.py', 'model.py')


def main():
    store = {}
    for short, exp in EXPERIMENTS.items():
        store[f'{short}/dirname'] = np.array(exp)
        for name in FILES:
            with open(os.path.join(REF, 'experiments', exp, name), 'rb') as f:
                blob = f.read()
            store[f'{short}/{name}'] = np.frombuffer(blob, dtype=np.uint8)
            store[f'{short}/{name}/sha256'] = np.array(hashlib.sha256(blob).hexdigest())
    with open(os.path.join(REF, 'LICENSE'), 'rb') as f:
        store['LICENSE'] = np.frombuffer(f.read(), dtype=np.uint8)
    path = os.path.join(HERE, 'reference_scripts.npz')
    np.savez_compressed(path, **store)
    print('reference_scripts', os.path.getsize(path), 'bytes;', len(EXPERIMENTS) * len(FILES), 'scripts')


if __name__ == '__main__':
    main()
