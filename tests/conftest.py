import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

torch.set_grad_enabled(False)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU oracle runs, excluded from the default CPU suite")


def pytest_collection_modifyitems(config, items):
    have_gpu = torch.cuda.is_available()
    skip_gpu = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(skip_gpu)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            path = os.path.join(GOLDEN, name + ".npz")
            if not os.path.exists(path):
                pytest.skip("fixture %s.npz not generated" % name)
            with np.load(path) as z:
                cache[name] = {k: torch.from_numpy(z[k]) for k in z.files}
        return cache[name]

    return load


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (built on demand; hipcc cross-compiles without a GPU)."""
    from neo360_amd import build
    return build.build()


def max_abs(a, b):
    a = torch.as_tensor(a).detach().cpu() if not isinstance(a, torch.Tensor) else a.detach().cpu()
    b = torch.as_tensor(b).detach().cpu() if not isinstance(b, torch.Tensor) else b.detach().cpu()
    return float((a.double() - b.double()).abs().max())
