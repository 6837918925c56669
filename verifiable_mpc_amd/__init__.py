"""MI355X-native AC20 compressed-Sigma-protocol hot path (Pedersen vector-commitment MSM +
log-round A_i/B_i folding over Ed25519) behind the call signatures of
toonsegers/verifiable_mpc's verifiable_mpc/ac20/{pivot,compressed_pivot}.py.

Importing the package does not touch the GPU; the first kernel call loads
libvmpc_hip.so and raises if it (or a GPU) is missing - there is no CPU fallback.
"""
from . import circuit_sat, compressed_pivot, pivot  # noqa: F401
from .circuit_sat import PivotChoice, circuit_sat_prover, circuit_sat_verifier, create_generators  # noqa: F401
from .device import PointVector, ScalarVector, get_context  # noqa: F401
from .fields import GF  # noqa: F401
from .formats import get_reference_format, reset_reference_format, set_reference_format  # noqa: F401
from .groups import Ed25519Point, EllipticCurve, EllipticCurvePoint  # noqa: F401

__version__ = "0.1.0"


def install(reference_package="verifiable_mpc.ac20"):
    """Point an importable copy of the reference at this package:
    replaces the hot-path functions in the reference's modules so that its circuit front
    end and demos run unchanged on top of the GPU path.  Returns the patched names."""
    import importlib
    ref_pivot = importlib.import_module(reference_package + ".pivot")
    ref_cp = importlib.import_module(reference_package + ".compressed_pivot")
    ref_r1cs = importlib.import_module(reference_package + ".circuit_sat_r1cs")
    patched = []
    for mod, name, fn in [
        (ref_pivot, "vector_commitment", pivot.vector_commitment),
        (ref_pivot, "fiat_shamir_hash", pivot.fiat_shamir_hash),
        (ref_cp, "protocol_5_prover", compressed_pivot.protocol_5_prover),
        (ref_cp, "protocol_5_verifier", compressed_pivot.protocol_5_verifier),
        (ref_cp, "protocol_4_prover", compressed_pivot.protocol_4_prover),
        (ref_cp, "protocol_4_verifier", compressed_pivot.protocol_4_verifier),
        (ref_r1cs, "create_generators", circuit_sat.create_generators),
    ]:
        setattr(mod, name, fn)
        patched.append(f"{mod.__name__}.{name}")
    try:
        ref_cb = importlib.import_module(reference_package + ".circuit_sat_cb")
        ref_cb.create_generators = circuit_sat.create_generators
        patched.append(f"{ref_cb.__name__}.create_generators")
    except ImportError:
        pass
    return patched
