"""The stand-in harness by itself, on the CPU: tests/refshape follows the reference's call pattern closely enough to
reproduce, hash for hash and representative for representative, the fixture that the reference's OWN modules produced
for the seeded `demo_zkp_ac20.py --elliptic` run (tests/golden/make_fixtures.py demo_case).  That is what entitles the
GPU test (tests/test_gpu_dropin.py) to use it as "the reference calling the installed functions"."""
import io
import random

SEED = 20200152 + 600
hx = lambda v: format(int(v), "x")


def proj_hex(pt):
    return [hx(c.value) for c in pt.value]


def typed_of(v, order):
    return "i:" + str(v) if isinstance(v, int) else "f:" + hx(int(v) % order)


def seed_like_the_fixture(rs):
    # make_fixtures.demo_case: prng of (circuit_sat_r1cs, circuit_sat_cb, compressed_pivot, pivot) = Random(seed+10+i)
    rs.r1cs.prng = random.Random(SEED + 10)
    rs.compressed_pivot.prng = random.Random(SEED + 12)
    rs.pivot.prng = random.Random(SEED + 13)


class Replay:
    """circuit_sat_cb's prng: the front end's draws are not replayed, only gamma is needed (circuit_sat_cb.py:92)"""

    def __init__(self, values):
        self.values = list(values)

    def randrange(self, *a):
        return self.values.pop(0)


def record_hashes(rs, calls):
    inner = rs.pivot.fiat_shamir_hash

    def wrapped(input_list, order):
        c = inner(input_list, order)
        calls.append(c)
        return c
    rs.pivot.fiat_shamir_hash = wrapped


def check_against_fixture(case, proof, verification, calls, order):
    ret = case["returned_proof"]
    assert verification == case["verification"]
    assert [hx(c) for c in calls] == [h["c"] for h in case["all_hashes"]]
    assert list(proof.keys()) == ret["keys"]
    assert proj_hex(proof["z_commitment"]) == case["protocol8"]["z_commitment_proj"]
    assert [typed_of(v, order) for v in proof["L"].coeffs] == case["protocol8"]["L"]["coeffs"]
    pp = proof["pivot_proof"]
    assert list(pp.keys()) == ret["pivot_proof_keys"]
    assert typed_of(pp["t"], order) == ret["t_typed"]
    assert proj_hex(pp["A"]) == ret["A_proj"]
    for i in range(case["rounds"]):
        assert proj_hex(pp[f"A{i}"]) == ret["A_i_proj"][i]
        assert proj_hex(pp[f"B{i}"]) == ret["B_i_proj"][i]
    assert [typed_of(v, order) for v in pp["z_prime"]] == ret["z_prime_typed"]


def test_stand_in_harness_reproduces_the_reference_fixture_on_cpu(refshape, golden_demo):
    rs, case = refshape, golden_demo
    group, gf = rs.demo.group_and_field("Elliptic")
    seed_like_the_fixture(rs)
    rs.cs.prng = Replay([int(case["protocol8"]["gamma"], 16)])
    calls = []
    record_hashes(rs, calls)
    circuit = rs.frontends.FixtureCircuit(case, gf)
    out = io.StringIO()
    proof, generators, verification = rs.demo.main(rs.cs.PivotChoice.compressed, group, gf, circuit,
                                                   circuit.inputs(), out)
    check_against_fixture(case, proof, verification, calls, group.order)
    assert "'pivot_verification': True" in out.getvalue()


def test_stand_in_harness_quadratic_residues_and_basic_pivot_on_cpu(refshape):
    rs = refshape
    group, gf = rs.demo.group_and_field("QR")
    for choice, n in ((rs.cs.PivotChoice.compressed, 15), (rs.cs.PivotChoice.pivot, 6)):
        circuit = rs.frontends.SyntheticCircuit(gf, n, 3, 5)
        proof, generators, verification = rs.demo.main(choice, group, gf, circuit, circuit.inputs())
        assert verification == {"y1*y2=y3": True, "L_wellformed_from_Cfgh_forms": True, "pivot_verification": True}
        # and a wrong response does not verify
        bad = dict(proof)
        if choice is rs.cs.PivotChoice.compressed:
            bad["pivot_proof"] = dict(proof["pivot_proof"], t=proof["pivot_proof"]["t"] + 1)
        else:
            z, phi, c = proof["pivot_proof"]
            bad["pivot_proof"] = (z, phi + 1, c)
        assert rs.cs.circuit_sat_verifier(bad, generators, circuit, gf, choice)["pivot_verification"] is False
