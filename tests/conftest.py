import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def built_lib():
    """hipcc cross-compiles gfx950 without a GPU; the prebuilt .so travels to the GPU box."""
    from masr_amd import build
    return build.build()


def needs_experiments():
    """skip marker for the tests of the measured-and-rejected kernels: they are compiled only with MASR_BUILD_EXPERIMENTS=1
    (masr_amd/build.py); the default library holds the product kernels and refuses their masr_debug_set keys"""
    from masr_amd import build
    return pytest.mark.skipif(not build.has_experiments(), reason='experimental kernels are not in this build (MASR_BUILD_EXPERIMENTS=1)')
