"""The pieces of verifiable_mpc/ac20/circuit_sat_r1cs.py / circuit_sat_cb.py that sit on the
AC20 hot path: generator setup and the pivot dispatch enum.

    PivotChoice                     circuit_sat_r1cs.py:39-44
    create_generators               circuit_sat_r1cs.py:47-93
    next_power_of_2                 circuit_sat_r1cs.py:391-392
    check_input_length_power_of_2   circuit_sat_cb.py:46-56

The circuit front end (circuit_builder, Protocol 8's form construction) is out of scope
and keeps running in the reference's Python; `verifiable_mpc_amd.install()` points the
reference's modules at the functions of this package.
"""
from enum import Enum
from random import SystemRandom

from .device import PointVector
from .groups import Ed25519Point, adopt_notation, as_point, is_ed25519_group

prng = SystemRandom()


class PivotChoice(Enum):
    """Select pivot proof system."""
    pivot = 1
    compressed = 2
    koe = 3


def choice_name(pivot_choice):
    """'pivot' / 'compressed' / 'koe' for a member of ANY PivotChoice enum - the reference's callers pass members of
    circuit_sat_r1cs.PivotChoice (demos/demo_zkp_ac20.py:27,78), not of the class above - or for the bare name"""
    return getattr(pivot_choice, "name", pivot_choice)


# Generators are a CRS: up to this many get a fixed-base table at creation (at most 2 KiB each, capped
# at 256 MiB per table).  Besides shortening every commitment over g, the table lets the compact-transcript
# prover skip the generator folds altogether (compressed_pivot._tabulated).
PRECOMPUTE_MAX = 1 << 20


WIDE_TABLE_MIN = (1 << 19) - 1


def create_generators(g_length, pivot_choice, group=None, progress_bar=False):
    """Create generators g, h, k with g_i = h ** r_i on the GPU (one lane per generator,
    csrc/exact.hip k_repeat).  Exponents are drawn from `prng` in the reference's order:
    r_0 .. r_{g_length-1}, then k's exponent (circuit_sat_r1cs.py:64,81)."""
    choice = choice_name(pivot_choice)
    if choice not in ("pivot", "compressed"):
        # the KoE pivot lives on BN256 with pairings: not part of the accelerated path (an installed reference
        # keeps its own create_generators for it, dropin.py)
        raise NotImplementedError
    assert group is not None
    if not is_ed25519_group(group):
        raise NotImplementedError(f"{getattr(group, '__name__', group)}: only Ed25519 (projective) is accelerated; "
                                  "install() leaves other groups with the reference's own create_generators")
    adopt_notation(group)
    h = as_point(group.generator)
    random_exponents = list(prng.randrange(1, group.order) for i in range(g_length))
    if progress_bar:
        print("Generating keys: on device", end="\r")
    g = PointVector.fixed_base(h, random_exponents)
    if choice == "pivot":
        if g_length <= PRECOMPUTE_MAX:
            g.precompute([h], wide=g_length >= WIDE_TABLE_MIN)
        return {"g": g, "h": h}
    k = Ed25519Point.repeat(h, prng.randrange(1, group.order))
    if g_length <= PRECOMPUTE_MAX:
        # from 2^19 generators the wide-window table beside the prover's (commitments and the rounds before the fold
        # jump read it: 13 mixed additions per term; + 13 x 128 bytes per generator of HBM)
        g.precompute([h, k], wide=g_length >= WIDE_TABLE_MIN)
    return {"g": g, "h": h, "k": k}


def next_power_of_2(x):
    return 1 << (x).bit_length()


def check_input_length_power_of_2(x, circuit, padding_value=0):
    """Padding needed so that len(z) + 1 is a power of two (circuit_sat_cb.py:46-56)."""
    assert circuit.input_ct == len(x)
    z_len = circuit.input_ct + 3 + 2 * circuit.mul_ct
    if not bin(z_len + 1).count("1") == 1:
        padding = next_power_of_2(z_len) - z_len - 1
    else:
        padding = 0
    check = padding == 0
    return check, padding, z_len + padding


# ---- the harness names (circuit_sat_cb.py:255-318) ----------------------------------------------------------------
# The circuit front end - circuit_builder, the Protocol-8 form construction, the dispatch on PivotChoice - is
# quadratic-time Python that stays with the reference (out of scope).  These two names exist so
# that a caller who switches to this package finds them: they install() the hot path into the reference's modules
# and hand the call to the reference's own function, unchanged.
REFERENCE_PACKAGE = "verifiable_mpc.ac20"


def _reference_circuit_sat():
    import importlib
    from . import install
    try:
        ref = importlib.import_module(REFERENCE_PACKAGE + ".circuit_sat_cb")
    except ImportError as e:
        raise ImportError(
            f"circuit_sat_prover / circuit_sat_verifier delegate to the reference's {REFERENCE_PACKAGE}.circuit_sat_cb "
            f"(circuit front end, out of scope of this package); it is not importable here: {e}") from e
    install(REFERENCE_PACKAGE)
    return ref


def circuit_sat_prover(generators, circuit, x, gf, pivot_choice=PivotChoice.compressed):
    """circuit_sat_cb.py:255-282, same arguments and return value: the reference's function over this package's
    vector_commitment / protocol_5_prover / create_generators (install())."""
    ref = _reference_circuit_sat()
    return ref.circuit_sat_prover(generators, circuit, x, gf, _ref_choice(ref, pivot_choice))


def circuit_sat_verifier(proof, generators, circuit, gf, pivot_choice=PivotChoice.compressed):
    """circuit_sat_cb.py:285-318, same arguments and return value (the verification dict)."""
    ref = _reference_circuit_sat()
    return ref.circuit_sat_verifier(proof, generators, circuit, gf, _ref_choice(ref, pivot_choice))


def _ref_choice(ref, choice):
    """this package's PivotChoice member -> the reference's member of the same name (the reference compares
    members of ITS enum)"""
    ref_enum = getattr(ref, "PivotChoice", None)
    name = getattr(choice, "name", None)
    if ref_enum is not None and name is not None and not isinstance(choice, ref_enum):
        return ref_enum[name]
    return choice
