import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    return binding.load()


@pytest.fixture(scope="session")
def gpu():
    import gaussian_process_liouville_equation_amd as pkg
    api = pkg.open_api(0)
    yield api
    api.close()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
