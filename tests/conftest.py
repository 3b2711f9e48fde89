import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:                      # `from conftest import ...` in the test modules, also when one file is run alone
    sys.path.insert(0, HERE)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix=""):
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
