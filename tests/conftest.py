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


def _torch_runtime_first():
    """On a GPU box bring torch's bundled HIP runtime up before libfic_hip.so's (see capi._torch_first): tests that hand
    torch CUDA tensors to the library must not depend on which test ran first.  Done when conftest is imported, i.e. before
    any test module is collected: some of them ask the library at import time how it was built (capi.has_xcheck())."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


_torch_runtime_first()


@pytest.fixture(scope="session")
def oracle():
    from oracle import fic_oracle
    fic_oracle.build()
    fic_oracle.lib()
    return fic_oracle


@pytest.fixture(scope="session")
def lena_grey():
    return np.load(os.path.join(GOLDEN, "lena_grey_256.npy"))


@pytest.fixture(scope="session")
def lena64():
    return np.load(os.path.join(GOLDEN, "lena64.npy"))


@pytest.fixture(scope="session")
def lena_colored():
    return np.load(os.path.join(GOLDEN, "lena_colored_256.npy"))


def same_f32(x, y):
    """Bit equality of float32 arrays, treating every NaN as equal (Java has one NaN value)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    nx, ny = np.isnan(x), np.isnan(y)
    return bool((nx == ny).all() and (x.view(np.uint32)[~nx] == y.view(np.uint32)[~ny]).all())
