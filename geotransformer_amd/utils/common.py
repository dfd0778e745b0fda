"""The two helpers of geotransformer/utils/common.py the hot path's callers import (experiments/*/config.py:7, test.py:9)."""
import os
import pickle


def ensure_dir(path):
    """geotransformer/utils/common.py:6-8."""
    os.makedirs(path, exist_ok=True)


def load_pickle(filename):
    with open(filename, 'rb') as f:
        return pickle.load(f)


def dump_pickle(data, filename):
    with open(filename, 'wb') as f:
        pickle.dump(data, f)
