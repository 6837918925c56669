"""Stand-in for the two names of verifiable_mpc/ac20/circuit_sat_r1cs.py that sit on the path: the caller's OWN
PivotChoice enum (circuit_sat_r1cs.py:39-44 - a different class from verifiable_mpc_amd.PivotChoice, which is the
point) and create_generators (circuit_sat_r1cs.py:47-93) over any group, exponents drawn in the reference's order."""
import enum
from random import SystemRandom

prng = SystemRandom()

PivotChoice = enum.Enum("PivotChoice", [("pivot", 1), ("compressed", 2), ("koe", 3)])

KOE_SETUP_CALLS = []        # the KoE branch only records that it was reached (pairings are out of scope everywhere)


def create_generators(g_length, pivot_choice, group=None, progress_bar=False):
    if pivot_choice is PivotChoice.koe and isinstance(group, list):
        KOE_SETUP_CALLS.append((g_length, group))
        return {"pp_lhs": [], "pp_rhs": []}
    if pivot_choice not in (PivotChoice.pivot, PivotChoice.compressed):
        raise NotImplementedError
    assert group is not None
    h = group.generator
    exponents = [prng.randrange(1, group.order) for _ in range(g_length)]
    generators = {"g": [h ** e for e in exponents], "h": h}
    if pivot_choice is PivotChoice.compressed:
        generators["k"] = h ** prng.randrange(1, group.order)
    return generators
