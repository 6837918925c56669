"""Stand-in for the harness half of verifiable_mpc/ac20/circuit_sat_cb.py: Protocol 8 around an arbitrary pivot
(circuit_sat_cb.py:59-166 prover, :169-252 verifier, :255-318 the dispatch on PivotChoice).

The circuit front end (circuit_builder, calculate_fgh_polys, the f/g/h forms) is out of scope of the build and absent
from the GPU box, so `circuit` here is a data object (tests/refshape/frontends.py) that answers what the front end
would compute: str(circuit), the witness vector z for the inputs x, and - as a function of the first challenge - the
responses and forms that enter the second hash.  Everything that touches the hot path is called the way the reference
calls it: `pivot.vector_commitment`, `pivot.fiat_shamir_hash`, `compressed_pivot.protocol_5_prover`,
`pivot.prove_linear_form_eval` through the MODULE objects, `create_generators` / `PivotChoice` imported by NAME."""
from random import SystemRandom

from . import compressed_pivot, pivot
from .circuit_sat_r1cs import PivotChoice, create_generators  # noqa: F401  (cs.create_generators: demo_zkp_ac20.py:78)

prng = SystemRandom()
FIRST, SECOND = "First hash circuit satisfiability protocol", "Second hash circuit satisfiability protocol"


def _first_challenge(circuit, z_commitment, gf):
    return pivot.fiat_shamir_hash([z_commitment, str(circuit), FIRST], gf.order)


def _nullity_form(circuit, z_commitment, c, y, gf):
    """both sides of Protocol 8 derive the same L from the same two hashes (circuit_sat_cb.py:107-164, :205-244)"""
    outputs, circuit_forms, lin_forms = circuit.forms(c, y)
    rho = pivot.fiat_shamir_hash(list(y) + [z_commitment, outputs, circuit_forms, lin_forms, SECOND], gf.order)
    return outputs, sum(form * (rho ** i) for i, form in enumerate(lin_forms))


def protocol_8_excl_pivot_prover(generators, circuit, x, gf):
    z = circuit.witness(x)
    gamma = prng.randrange(1, gf.order)
    z_commitment = pivot.vector_commitment(z, gamma, generators["g"], generators["h"])
    proof = {"z_commitment": z_commitment}
    c = _first_challenge(circuit, z_commitment, gf)
    y = circuit.responses(c, z)
    assert y[0] * y[1] == y[2]
    proof["y1"], proof["y2"], proof["y3"] = y
    outputs, L = _nullity_form(circuit, z_commitment, c, y, gf)
    proof["outputs"] = outputs
    proof["L"] = L
    return proof, z_commitment, L, z, gamma


def protocol_8_excl_pivot_verifier(proof, circuit, gf):
    y = [proof["y1"], proof["y2"], proof["y3"]]
    if not y[0] * y[1] == y[2]:
        return {"y1*y2=y3": False}, None
    verification = {"y1*y2=y3": True}
    c = _first_challenge(circuit, proof["z_commitment"], gf)
    _, L = _nullity_form(circuit, proof["z_commitment"], c, y, gf)
    verification["L_wellformed_from_Cfgh_forms"] = bool(L == proof["L"])
    return verification, L


def circuit_sat_prover(generators, circuit, x, gf, pivot_choice=PivotChoice.compressed):
    proof, z_commitment, L, z, gamma = protocol_8_excl_pivot_prover(generators, circuit, x, gf)
    if pivot_choice == PivotChoice.compressed:
        pivot_proof = compressed_pivot.protocol_5_prover(generators, z_commitment, L, L(z), z, gamma, gf)
    elif pivot_choice == PivotChoice.pivot:
        pivot_proof = pivot.prove_linear_form_eval(generators["g"], generators["h"], z_commitment, L, L(z), z, gamma, gf)
    else:
        raise NotImplementedError
    proof["pivot_proof"] = pivot_proof
    return proof


def circuit_sat_verifier(proof, generators, circuit, gf, pivot_choice=PivotChoice.compressed):
    verification, L = protocol_8_excl_pivot_verifier(proof, circuit, gf)
    if not verification.get("L_wellformed_from_Cfgh_forms"):
        return verification
    if pivot_choice == PivotChoice.compressed:
        ok = compressed_pivot.protocol_5_verifier(generators, proof["z_commitment"], L, 0, proof["pivot_proof"], gf)
    elif pivot_choice == PivotChoice.pivot:
        z, phi, c = proof["pivot_proof"]
        ok = pivot.verify_linear_form_proof(generators["g"], generators["h"], proof["z_commitment"], L, 0, z, phi, c)
    else:
        raise NotImplementedError
    verification["pivot_verification"] = ok
    return verification
