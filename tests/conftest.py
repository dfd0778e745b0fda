import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device skips the gpu-marked tests instead of failing inside them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='needs a HIP GPU (run with -m gpu on the GPU box)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def oracle_lib():
    """The CPU restatement (oracle/libneighbors_oracle.so), built on demand."""
    from oracle import neighbors
    return neighbors.restated()


@pytest.fixture(scope='session')
def reference_lib():
    """The real reference cores (oracle/_ref/libgeoref.so) or skip when never built."""
    from oracle import neighbors
    lib = neighbors.reference()
    if lib is None:
        pytest.skip('oracle/_ref/libgeoref.so not built (needs /root/reference)')
    return lib


@pytest.fixture(params=['bf16x3', 'fp32'])
def matrix_precision(request):
    """Both fp32-grade arithmetic modes of the matrix-pipe kernels: split-bf16 products ('bf16x3') and exact fp32 MFMA products
    ('fp32', the reference's own arithmetic and the mode the headline is measured in).  Yields the mode's name."""
    from geotransformer_amd import kernels
    prev = kernels.set_precision(request.param)
    try:
        yield request.param
    finally:
        kernels.set_precision(prev)
