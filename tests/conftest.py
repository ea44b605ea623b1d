import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ckpt_ra1e4():
    import numpy as np
    d = np.load(os.path.join(GOLDEN, "ckpt2d_ra10000.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def ckpt_ra1e5():
    import numpy as np
    d = np.load(os.path.join(GOLDEN, "ckpt2d_ra100000.npz"))
    return {k: d[k] for k in d.files}
