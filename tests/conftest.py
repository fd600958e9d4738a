import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def amlike_device():
    """Binds libamlike to GPU 0; fails loudly when the library or the GPU is missing."""
    from kaldi_amd import _lib
    _lib.check(_lib.lib().amlike_init(0))
    return 0
