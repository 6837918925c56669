"""MI355X-native AC20 compressed-Sigma-protocol hot path (Pedersen vector-commitment MSM +
log-round A_i/B_i folding over Ed25519) behind the call signatures of
toonsegers/verifiable_mpc's verifiable_mpc/ac20/{pivot,compressed_pivot}.py.

Importing the package does not touch the GPU; the first kernel call loads
libvmpc_hip.so and raises if it (or a GPU) is missing - there is no CPU fallback.
"""
import os as _os

# the HIP runtime's hardware-queue count, before anything initialises HIP (csrc/api.hip explains; the library sets the
# same default when it is loaded, this covers a host that imports the package and THEN initialises HIP through torch)
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from . import circuit_sat, compressed_pivot, pivot  # noqa: E402,F401
from .circuit_sat import PivotChoice, circuit_sat_prover, circuit_sat_verifier, create_generators  # noqa: E402,F401
from .device import PointVector, ScalarVector, get_context  # noqa: E402,F401
from .fields import GF  # noqa: E402,F401
from .formats import get_reference_format, reset_reference_format, set_reference_format  # noqa: E402,F401
from .groups import Ed25519Point, EllipticCurve, EllipticCurvePoint, is_ed25519_element  # noqa: E402,F401

__version__ = "0.1.0"


from .dropin import install, uninstall  # noqa: E402,F401  (rebinds the reference's hot-path names: dropin.py)
