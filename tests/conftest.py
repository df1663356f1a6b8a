import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def losses_golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "losses_golden.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def model_golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "model_golden.npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def extra_golden():
    import numpy as np
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "extra_golden.npz"), allow_pickle=False))
