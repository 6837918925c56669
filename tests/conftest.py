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


@pytest.fixture()
def refshape():
    """tests/refshape (the builder-written stand-in for an importable reference, see its __init__) imported over
    tests/golden/mpyc_shim; both leave sys.modules again afterwards, and whatever install() did is undone."""
    import types
    shim = os.path.join(GOLDEN, "mpyc_shim")

    def ours(name):
        return name == "mpyc" or name.startswith("mpyc.") or name.startswith("tests.refshape")
    saved = {k: v for k, v in sys.modules.items() if ours(k)}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, shim)
    try:
        from tests.refshape import demo, frontends
        from tests.refshape.ac20 import circuit_sat_cb, circuit_sat_r1cs, compressed_pivot, pivot
        yield types.SimpleNamespace(package="tests.refshape.ac20", demo=demo, frontends=frontends, cs=circuit_sat_cb,
                                    r1cs=circuit_sat_r1cs, compressed_pivot=compressed_pivot, pivot=pivot)
    finally:
        if "verifiable_mpc_amd" in sys.modules:
            sys.modules["verifiable_mpc_amd"].uninstall("tests.refshape.ac20")
            sys.modules["verifiable_mpc_amd"].Ed25519Point.is_additive = True
            sys.modules["verifiable_mpc_amd"].Ed25519Point.is_multiplicative = False
        sys.path.remove(shim)
        for k in [k for k in sys.modules if ours(k)]:
            del sys.modules[k]
        sys.modules.update(saved)
