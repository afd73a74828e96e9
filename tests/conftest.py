import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    d = os.path.join(REPO, "tests", "golden")
    return {n: np.load(os.path.join(d, n + ".npz")) for n in ("kat", "segment", "spmm", "layers", "sampler", "convert")}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o
