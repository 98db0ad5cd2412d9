import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """-m gpu tests must not silently pass on a box without a GPU.  The feeder's GPU tests fork worker processes: in a
    fresh process a fork takes 6-8 ms (profiles/r03_b_fork_probe.txt), at the end of this suite -- a hundred handles created
    and destroyed, tens of GB mapped and unmapped -- it takes seconds (the same five tests: 17 s first, 230 s last), so
    they run first."""
    items.sort(key=lambda it: 0 if 'test_gpu_feeder' in it.nodeid else (1 if 'test_gpu_learning' in it.nodeid else 2))      # stable: everything else keeps its order
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
