"""Build-container only: the REAL reference modules (imported from /root/reference over tests/golden/mpyc_shim, as
tests/golden/make_fixtures.py does) after `verifiable_mpc_amd.install()`, driven by the reference's own
demos/demo_zkp_ac20.main.  The reference never travels to the GPU box and this container has no GPU, so the
accelerated branches are followed up to their first native call (a sentinel raised from get_context); the groups the
GPU path leaves alone run to the end.  Skipped wherever /root/reference is absent."""
import os
import subprocess
import sys

import pytest

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "verifiable_mpc", "ac20")),
                                reason="the reference checkout exists in the build container only")

DRIVER = r'''
import sys, types
sys.dont_write_bytecode = True
root, reference, group, choice = sys.argv[1:5]
sys.path[:0] = [root + "/tests/golden/mpyc_shim", reference, reference + "/demos", root]
stub = types.ModuleType("verifiable_mpc.ac20.pairing")      # BN256 pairings: outside the shim (make_fixtures.py)
def _no_pairing(*a, **k):
    raise NotImplementedError("pairing")
stub.optimal_ate = _no_pairing
sys.modules["verifiable_mpc.ac20.pairing"] = stub

import verifiable_mpc_amd as vm
from verifiable_mpc_amd import device

class Reached(Exception):
    pass

def no_gpu_here():
    raise Reached("native call")
device.get_context = no_gpu_here

patched = vm.install()
import demo_zkp_ac20 as demo
import verifiable_mpc.ac20.circuit_sat_cb as cs
import verifiable_mpc.ac20.knowledge_of_exponent as koe
from mpyc.fingroups import QuadraticResidues
demo.QuadraticResidues = lambda l=None: QuadraticResidues(l=64)   # l=1024: minutes of safe-prime search in the shim
demo.GROUP = group
def setup_reached(*a, **k):
    raise Reached("koe.trusted_setup")
koe.trusted_setup = setup_reached
try:
    result = demo.main(getattr(cs.PivotChoice, choice), 3)
    print("RESULT", sorted(result.items()))
except Reached as e:
    import traceback
    frames = [f.name for f in traceback.extract_tb(e.__traceback__)]
    print("REACHED", e, "|".join(frames))
'''


def drive(group, choice):
    out = subprocess.run([sys.executable, "-B", "-c", DRIVER, ROOT, REFERENCE, group, choice],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith(("RESULT", "REACHED"))][-1]


@pytest.mark.parametrize("choice", ["compressed", "pivot"])
def test_reference_demo_reaches_the_native_path_on_ed25519(choice):
    """demos/demo_zkp_ac20.py:78 `cs.create_generators(g_length, cs.PivotChoice.<choice>, group, progress_bar=True)`
    with the reference's enum and the (shim) MPyC group: accepted, and the first thing that stops it is the GPU"""
    line = drive("Elliptic", choice)
    assert line.startswith("REACHED native call"), line
    assert "create_generators" in line and "fixed_base" in line


@pytest.mark.parametrize("choice", ["compressed", "pivot"])
def test_reference_demo_default_group_runs_on_the_reference_cpu_path(choice):
    """the demo's default QuadraticResidues group (demo_zkp_ac20.py:50-52) and --basic after install()"""
    line = drive("QR", choice)
    assert line == "RESULT [('L_wellformed_from_Cfgh_forms', True), ('pivot_verification', True), ('y1*y2=y3', True)]"


def test_reference_demo_koe_reaches_the_reference_trusted_setup():
    line = drive("QR", "koe")
    assert line.startswith("REACHED koe.trusted_setup"), line
