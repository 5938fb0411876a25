import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "cal_22-mpc_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pkg(sub=None):
    """Import the package (its directory name has a dash, so go through importlib)."""
    return importlib.import_module(PKG if sub is None else f"{PKG}.{sub}")


@pytest.fixture(scope="session")
def configs():
    return pkg("configs")


@pytest.fixture(scope="session")
def traces():
    return pkg("traces")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build(ref=True)
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
