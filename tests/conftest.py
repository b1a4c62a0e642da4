import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libsacref.so (genuine reference build)")


@pytest.fixture(scope="session")
def orc():
    from oracle_api import Checker

    return Checker("orc")


@pytest.fixture(scope="session")
def ref():
    from oracle_api import Checker, ref_available

    if not ref_available():
        pytest.skip("oracle/_ref/libsacref.so not built (needs /root/reference)")
    return Checker("ref")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "ref_golden.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden_r2():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "ref_golden_r2.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden_r4():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "ref_golden_r4.npz")
    return np.load(path, allow_pickle=False)


@pytest.fixture(scope="session")
def golden_r3():
    import numpy as np

    path = os.path.join(ROOT, "tests", "golden", "ref_golden_r3.npz")
    return np.load(path, allow_pickle=False)
