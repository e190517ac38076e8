import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        cases = {}
        for key in z.files:
            if "__" in key:
                case, field = key.split("__", 1)
                cases.setdefault(case, {})[field] = z[key]
            else:
                cases[key] = z[key]
        return cases

    return load


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
