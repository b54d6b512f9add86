"""pytest configuration: markers, repo path, shared fixtures."""
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import importlib

    return importlib.import_module("beamforming-lk_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.oracle()  # builds liboracle_das.so if needed
    return oracle_py
