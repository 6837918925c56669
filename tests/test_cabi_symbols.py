"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/vmpc.h declares (no compute calls here)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    from verifiable_mpc_amd import _native, build
    build.build(verbose=False)
    return _native


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vmpc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vmpc_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_table(native):
    assert header_symbols() == sorted(native.SYMBOLS)


def test_library_exports_every_symbol(native):
    lib = native.load_library()
    out = subprocess.check_output(["nm", "-D", "--defined-only", native.LIB_PATH], text=True)
    exported = set(re.findall(r" T (vmpc_[a-z0-9_]+)", out))
    for name in header_symbols():
        assert name in exported, name
        assert getattr(lib, name) is not None


def test_no_gpu_is_reported_not_hidden(native):
    """without a GPU the library says so (VMPC_E_NODEV) instead of falling back to anything"""
    n, info = native.backend_info()
    if n >= 1:
        pytest.skip("GPU present")
    assert n == native.E_NODEV
    with pytest.raises(native.VmpcError) as ei:
        native.Context(0)
    assert ei.value.code == native.E_NODEV
    import numpy as np
    with pytest.raises(native.VmpcError):
        native.ed25519_msm(np.zeros((1, 32), np.uint8), np.zeros((1, 64), np.uint8))


def test_header_cites_reference_call_sites():
    text = open(os.path.join(ROOT, "include", "vmpc.h")).read()
    for cite in ("pivot.py:143-144", "compressed_pivot.py:64", "circuit_sat_r1cs.py:64-70",
                 "compressed_pivot.py:70-76", "pivot.py:84-92"):
        assert cite in text
