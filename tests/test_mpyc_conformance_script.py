"""scripts/check_against_mpyc.py is what pins parity with REAL MPyC's byte formats on a machine that has MPyC.
It cannot do that here (no MPyC), but its comparisons can be exercised: run over the build's stand-in
(VMPC_CHECK_AGAINST_SHIM=1) every check must come out equal, and without MPyC it must say so and exit 2."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "check_against_mpyc.py")


def test_without_mpyc_the_script_reports_and_exits_2():
    try:
        import mpyc  # noqa: F401
        pytest.skip("real MPyC is installed here: run the script itself")
    except ImportError:
        pass
    out = subprocess.run([sys.executable, "-B", SCRIPT], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 2 and "not importable" in out.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/verifiable_mpc"), reason="needs the reference checkout")
def test_every_comparison_passes_over_the_stand_in():
    env = dict(os.environ, VMPC_CHECK_AGAINST_SHIM="1")
    out = subprocess.run([sys.executable, "-B", SCRIPT, "--reference", "/root/reference"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "DIFF" not in out.stdout and out.stdout.count("  ok    ") > 60
    assert "pre-image 2: challenge" in out.stdout           # c0, c1 and the one round of the N = 4 case compared
    assert "self-test" in out.stdout


@pytest.mark.skipif(not os.path.isdir("/root/reference/verifiable_mpc"), reason="needs the reference checkout")
@pytest.mark.parametrize("shim_format,call", [("()|1|0", "point_brackets='()', coord_signed=True, scalar_signed=False"),
                                              ("[]|1|1", "coord_signed=True")])
def test_a_different_format_is_detected_and_named(shim_format, call):
    """the stand-in made to print the OTHER way (round brackets, signed coordinates, unsigned scalars): the script must
    name the ONE call that fixes it, and with that call applied every comparison - including the whole N = 4
    Protocol-5 case against the oracle under the same switches - must come out equal (exit status 1: defaults differ)."""
    env = dict(os.environ, VMPC_CHECK_AGAINST_SHIM="1", VMPC_SHIM_FORMAT=shim_format)
    out = subprocess.run([sys.executable, "-B", SCRIPT, "--reference", "/root/reference"], capture_output=True,
                         text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 1, out.stdout[-3000:] + out.stderr[-2000:]
    assert f"verifiable_mpc_amd.set_reference_format({call})" in out.stdout
    assert "DIFF" not in out.stdout, out.stdout[-3000:]
    assert "challenges c0, c1, c (oracle, switched format)" in out.stdout and "all equal ONCE" in out.stdout
