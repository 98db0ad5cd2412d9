import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must not silently pass on a box without a GPU.  (Round 3 ran the feeder's tests first because forking
    worker processes from a long-lived test process took seconds; the workers now come from a fork server --
    training_data._start_context -- and the suite runs in its natural order.)"""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def unfused_pools(monkeypatch):
    """Handles created inside the test keep pool1-3 as separate kernels (SSD_POOL_FUSE=0, read per handle), so every
    activation and gradient is materialised for the layer-local oracle checks; tests/test_gpu_pool_fusion.py shows the fused
    step bit-identical to this one."""
    monkeypatch.setenv('SSD_POOL_FUSE', '0')


@pytest.fixture
def direct_convs(monkeypatch):
    """Handles created inside the test run the 3x3 / stride 1 layers on the direct fp32 kernels (SSD_WINOGRAD=0, read per
    handle): the form every layer had before round 6 stays under the same oracle tests as the Winograd form that replaced it."""
    monkeypatch.setenv('SSD_WINOGRAD', '0')
