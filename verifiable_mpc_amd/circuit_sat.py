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
from .groups import Ed25519Point

prng = SystemRandom()


class PivotChoice(Enum):
    """Select pivot proof system."""
    pivot = 1
    compressed = 2
    koe = 3


def _our_point(obj):
    if isinstance(obj, Ed25519Point):
        return obj
    return Ed25519Point((int(obj[0]), int(obj[1]), int(obj[2])))


# Generators are a CRS: up to this many get a fixed-base table at creation (at most 2 KiB each, capped
# at 256 MiB per table).  Besides shortening every commitment over g, the table lets the compact-transcript
# prover skip the generator folds altogether (compressed_pivot._tabulated).
PRECOMPUTE_MAX = 1 << 20


def create_generators(g_length, pivot_choice, group=None, progress_bar=False):
    """Create generators g, h, k with g_i = h ** r_i on the GPU (one lane per generator,
    csrc/exact.hip k_repeat).  Exponents are drawn from `prng` in the reference's order:
    r_0 .. r_{g_length-1}, then k's exponent (circuit_sat_r1cs.py:64,81)."""
    if pivot_choice not in (PivotChoice.pivot, PivotChoice.compressed):
        # the KoE pivot lives on BN256 with pairings: not part of the accelerated path
        raise NotImplementedError
    assert group is not None
    h = _our_point(group.generator)
    random_exponents = list(prng.randrange(1, group.order) for i in range(g_length))
    if progress_bar:
        print("Generating keys: on device", end="\r")
    g = PointVector.fixed_base(h, random_exponents)
    if pivot_choice == PivotChoice.pivot:
        if g_length <= PRECOMPUTE_MAX:
            g.precompute([h])
        return {"g": g, "h": h}
    k = Ed25519Point.repeat(h, prng.randrange(1, group.order))
    if g_length <= PRECOMPUTE_MAX:
        g.precompute([h, k])
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
