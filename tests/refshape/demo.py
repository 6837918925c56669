"""Stand-in for demos/demo_zkp_ac20.py:31-90: pick the group, flip its notation flags, build the field, run
create_generators -> circuit_sat_prover -> pprint -> circuit_sat_verifier through the `cs` module object."""
import pprint

from mpyc.finfields import GF
from mpyc.fingroups import EllipticCurve, QuadraticResidues

from .ac20 import circuit_sat_cb as cs


def group_and_field(name):
    if name == "Elliptic":
        group = EllipticCurve("Ed25519", "projective")
        group.is_additive = False
        group.is_multiplicative = True
    elif name == "QR":
        group = QuadraticResidues(l=64)       # the demo's l=1024 takes the shim minutes to find a safe prime for
    else:
        raise NotImplementedError(name)
    return group, GF(modulus=group.order)


def main(pivot_choice, group, gf, circuit, x, out=None):
    say = (lambda *a: None) if out is None else (lambda *a: print(*a, file=out))
    say("Pivot selected: ", pivot_choice)
    g_length = circuit.g_length
    generators = cs.create_generators(g_length, pivot_choice, group, progress_bar=False)
    say("Generators created/trusted setup done.")
    proof = cs.circuit_sat_prover(generators, circuit, x, gf, pivot_choice)
    say("Proof:")
    say(pprint.pformat(proof, indent=4))
    verification = cs.circuit_sat_verifier(proof, generators, circuit, gf, pivot_choice)
    say("Verification checks: ")
    say(pprint.pformat(verification, indent=4))
    return proof, generators, verification
