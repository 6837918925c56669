"""A short run of scripts/soak.py: random sizes (1 .. 2^15), uniform / witness-like / all-equal scalars through the
variable-base, tabulated (1-16 rows) and batched Ed25519 commitments and the BN-256 G1 / G2 sums, every result checked
by the exponent identity  sum_i s_i (e_i B) == (sum_i s_i e_i mod order) B  (pivot.py:143-144; pynocchio.py:228-246)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_parity_soak(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak.py"), "8", str(seed)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "soak ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_many_proofs_back_to_back():
    """scripts/prove_stress.py for 10 s: ~2000 compact proofs over random sizes 2^3 .. 2^13, each verified - the
    prover's rounds are queued ahead of their challenges (vmpc_p4_run_compact, compressed_pivot.py:29-86): no hang,
    no stale challenge, the pinned mailbox's sequence numbers never reused"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "prove_stress.py"), "10", "3"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0 and "prove stress ok" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
