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


# GPU files in the order they should run under ``pytest -x``: the stage-by-stage parity tests of SURVEY.md 8(a) first, the
# derived / driver / multi-rank tests after them, and the statistical training-trajectory tests (minutes of training runs) LAST --
# a failing trajectory test must not leave the parity rows unrun.  Files not named keep their
# alphabetical place between the two groups.
GPU_ORDER_FIRST = ("test_gpu_parity", "test_gpu_edges", "test_gpu_rng", "test_gpu_train", "test_gpu_losses", "test_gpu_criterion",
                   "test_gpu_optim", "test_gpu_configs", "test_gpu_driver", "test_gpu_generic", "test_gpu_manipulator",
                   "test_gpu_manipulator_frame", "test_gpu_bench_line", "test_gpu_sharded", "test_gpu_rccl")
GPU_ORDER_LAST = ("test_gpu_convergence",)


def _order_key(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in GPU_ORDER_FIRST:
        return (0, GPU_ORDER_FIRST.index(name))
    if name in GPU_ORDER_LAST:
        return (2, GPU_ORDER_LAST.index(name))
    return (1, 0)


def pytest_collection_modifyitems(session, config, items):
    items.sort(key=_order_key)                                      # stable: the order inside a file is untouched
