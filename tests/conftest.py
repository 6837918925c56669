import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_small():
    return load_golden("ac20_ed25519_small.json")


@pytest.fixture(scope="session")
def golden_demo():
    return load_golden("demo_zkp_ac20_elliptic.json")


@pytest.fixture(scope="session")
def golden_n1023():
    return load_golden("ac20_ed25519_n1023.json")
