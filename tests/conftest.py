import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def bxd():
    import numpy as np
    d = np.load(os.path.join(ROOT, "tests", "golden", "bxd.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def gpu_api():
    """The product path. Fails loudly (no skip, no fallback) when the HIP library or GPU is missing."""
    from gemma_amd import api
    api.init(0, verbose=0)
    return api
