"""Generates tests/golden/configs.json: the three reference experiments' `make_cfg()` trees, flattened to dotted keys.
Build-container only (imports /root/reference through oracle/ref_harness.py).  Directory entries depend on where the reference is
installed and are dropped; `data.dataset_root` is kept relative to the reference root.
usage: python tests/golden/make_config_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness  # noqa: E402

DIRECTORIES = {'root_dir', 'working_dir', 'output_dir', 'snapshot_dir', 'log_dir', 'event_dir', 'feature_dir', 'registration_dir', 'exp_name'}


def flatten(tree, prefix=''):
    out = {}
    for key, value in tree.items():
        if isinstance(value, dict):
            out.update(flatten(value, prefix + key + '.'))
        else:
            out[prefix + key] = value
    return out


def main():
    golden = {}
    for name in ('3dmatch', 'kitti', 'modelnet'):
        config, _ = ref_harness.load_experiment(name)
        flat = {k: v for k, v in flatten(dict(config.make_cfg())).items() if k not in DIRECTORIES}
        flat['data.dataset_root'] = os.path.relpath(flat['data.dataset_root'], ref_harness.REF_ROOT)
        golden[name] = flat
        print(name, len(flat), 'keys')
    path = os.path.join(ROOT, 'tests', 'golden', 'configs.json')
    with open(path, 'w') as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print('wrote', path)


if __name__ == '__main__':
    main()
