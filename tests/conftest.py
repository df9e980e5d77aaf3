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
def golden():
    def load(name):
        import torch
        with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
            return {k: (torch.from_numpy(z[k]) if z[k].ndim > 0 else z[k].item()) for k in z.files}
    return load
