"""The drop-in, driven the way the reference drives it.

`verifiable_mpc_amd.install()` is applied to tests/refshape - a builder-written package with the reference's module and
function names that follows the call pattern of circuit_sat_cb.py:255-318 and demos/demo_zkp_ac20.py:69-90 (its own
PivotChoice enum, the mpyc shim's GF elements and EllipticCurve generator, its own form classes; on the CPU it
reproduces the reference-made fixture: tests/test_refshape_harness.py).  Nothing in these tests constructs an input
with this package's types: whatever reaches the installed functions is foreign, as it is for a maintainer who adds the
two lines of INTEGRATION.md section 3 to the reference's demo."""
import io
import random

import pytest

from tests.test_refshape_harness import (Replay, check_against_fixture, proj_hex, record_hashes,
                                         seed_like_the_fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture()
def vm():
    import verifiable_mpc_amd as v
    v.get_context()
    return v


def record_internal_hashes(vm, monkeypatch, calls):
    """Protocol 5 / 4 inside the package call pivot.fiat_shamir_hash(_variants) of THIS package"""
    one, many = vm.pivot.fiat_shamir_hash, vm.pivot.fiat_shamir_hash_variants

    def wrapped(input_list, order):
        c = one(input_list, order)
        calls.append(c)
        return c

    def wrapped_many(common, tails, order):
        cs = many(common, tails, order)
        calls.extend(cs)
        return cs
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash", wrapped)
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash_variants", wrapped_many)


def test_installed_functions_reproduce_the_demo_fixture_from_foreign_types(vm, refshape, golden_demo, monkeypatch):
    """(i) every one of the 20 Fiat-Shamir hashes of the seeded demo run and the returned proof, representatives
    included, with the harness's enum / shim field elements / shim generator / harness forms as inputs"""
    rs, case = refshape, golden_demo
    patched = vm.install(rs.package)
    assert f"{rs.package}.pivot.prove_linear_form_eval" in patched
    assert rs.cs.create_generators.__vmpc_accelerated__ is vm.circuit_sat.create_generators
    group, gf = rs.demo.group_and_field("Elliptic")
    assert not isinstance(group.generator, vm.Ed25519Point) and rs.cs.PivotChoice is not vm.PivotChoice
    seed_like_the_fixture(rs)
    rs.cs.prng = Replay([int(case["protocol8"]["gamma"], 16)])
    calls = []
    record_internal_hashes(vm, monkeypatch, calls)
    record_hashes(rs, calls)                 # the two Protocol-8 hashes: through the harness's module attribute
    circuit = rs.frontends.FixtureCircuit(case, gf)
    x = circuit.inputs()
    assert not any(isinstance(v, vm.fields.FiniteFieldElement) for v in x)
    out = io.StringIO()
    proof, generators, verification = rs.demo.main(rs.cs.PivotChoice.compressed, group, gf, circuit, x, out)
    # the accelerated path ran: generators live on the device, returned points are this package's
    assert isinstance(generators["g"], vm.PointVector) and isinstance(proof["pivot_proof"]["A0"], vm.Ed25519Point)
    check_against_fixture(case, proof, verification, calls, group.order)
    assert "'pivot_verification': True" in out.getvalue()
    # the caller goes on in ITS notation with what it got back (demo_zkp_ac20.py:47-48 flipped the flags)
    A0 = proof["pivot_proof"]["A0"]
    assert (A0 * A0) == A0 ** 2
    # a tampered proof is rejected through the installed verifier, not raised on
    bad = dict(proof, pivot_proof=dict(proof["pivot_proof"], A1=proof["pivot_proof"]["B1"]))
    assert rs.cs.circuit_sat_verifier(bad, generators, circuit, gf, rs.cs.PivotChoice.compressed)[
        "pivot_verification"] is False


def _run(rs, choice, group_name, circuit_of, seed):
    group, gf = rs.demo.group_and_field(group_name)
    for i, mod in enumerate((rs.r1cs, rs.cs, rs.compressed_pivot, rs.pivot)):
        mod.prng = random.Random(seed + i)
    circuit = circuit_of(gf)
    calls = []
    record_hashes(rs, calls)
    proof, generators, verification = rs.demo.main(choice, group, gf, circuit, circuit.inputs())
    return proof, generators, verification, calls


@pytest.mark.parametrize("choice_name,n", [("compressed", 31), ("pivot", 12)])
def test_installed_equals_uninstalled_on_ed25519(vm, refshape, choice_name, n):
    """Both pivots over Ed25519 through the installed functions = the harness's own CPU code on the same seeds:
    generators, [z], every hash the harness makes, and the pivot proof (for Pi_s: z, phi and the challenge, whose
    pre-image holds the NORMALISED announcement, pivot.py:169-174)."""
    rs = refshape
    choice = rs.cs.PivotChoice[choice_name]
    circuit_of = lambda gf: rs.frontends.SyntheticCircuit(gf, n, 2, 77)
    want_proof, want_gens, want_ver, want_calls = _run(rs, choice, "Elliptic", circuit_of, 4100)
    assert isinstance(want_gens["g"], list)
    vm.install(rs.package)
    proof, gens, ver, calls = _run(rs, choice, "Elliptic", circuit_of, 4100)
    assert isinstance(gens["g"], vm.PointVector)
    assert ver == want_ver == {"y1*y2=y3": True, "L_wellformed_from_Cfgh_forms": True, "pivot_verification": True}
    assert [p.coords for p in gens["g"].to_points()] == [tuple(c.value for c in p.value) for p in want_gens["g"]]
    assert proof["z_commitment"].coords == tuple(c.value for c in want_proof["z_commitment"].value)
    assert calls[:2] == want_calls[:2]                       # the harness-level hashes (Protocol 8)
    if choice_name == "pivot":
        z, phi, c = proof["pivot_proof"]
        wz, wphi, wc = want_proof["pivot_proof"]
        assert (c, phi) == (wc, wphi) and [int(v) for v in z] == [int(v) for v in wz]
        assert [type(v) for v in z] == [type(v) for v in wz]
    else:
        pp, wp = proof["pivot_proof"], want_proof["pivot_proof"]
        assert list(pp) == list(wp)
        for key in wp:
            if key == "z_prime":
                assert [int(v) for v in pp[key]] == [int(v) for v in wp[key]]
            elif key == "t":
                assert int(pp[key]) == int(wp[key])
            else:
                assert list(pp[key].coords) == [c.value for c in wp[key].value], key


@pytest.mark.parametrize("choice_name,n", [("compressed", 15), ("pivot", 6)])
def test_other_groups_fall_through_to_the_original(vm, refshape, choice_name, n):
    """(ii) QuadraticResidues after install(): the harness's own functions run (plain lists, QR elements) and the
    proof verifies - SURVEY.md section 8b 'must fall through to the CPU path'"""
    rs = refshape
    vm.install(rs.package)
    choice = rs.cs.PivotChoice[choice_name]
    proof, gens, ver, calls = _run(rs, choice, "QR", lambda gf: rs.frontends.SyntheticCircuit(gf, n, 3, 5), 9)
    assert isinstance(gens["g"], list) and not isinstance(gens["h"], vm.Ed25519Point)
    assert ver == {"y1*y2=y3": True, "L_wellformed_from_Cfgh_forms": True, "pivot_verification": True}


def test_koe_goes_to_the_original_create_generators(vm, refshape):
    rs = refshape
    vm.install(rs.package)
    from mpyc.fingroups import EllipticCurve
    groups = [EllipticCurve("BN256", "jacobian"), EllipticCurve("BN256_twist", "jacobian")]
    assert rs.cs.create_generators(7, rs.cs.PivotChoice.koe, groups) == {"pp_lhs": [], "pp_rhs": []}
    assert rs.r1cs.KOE_SETUP_CALLS[-1][0] == 7
    # and a BN256 element handed to an installed function is not mistaken for an Ed25519 one (three coordinates too)
    g1 = groups[0].generator
    assert not vm.is_ed25519_element(g1)


def test_a_list_of_foreign_points_is_a_generator_vector(vm, refshape):
    """(c) `g` as the list of MPyC(-shim) points a reference caller holds; exponents shim field elements and ints,
    negative and oversized included: the commitment's (X:Y:Z) is the one the caller's own arithmetic produces"""
    rs = refshape
    group, gf = rs.demo.group_and_field("Elliptic")
    rng = random.Random(31)
    g = [group.generator ** rng.randrange(1, group.order) for _ in range(9)]
    h = group.generator ** rng.randrange(1, group.order)
    x = [gf(rng.randrange(group.order)) for _ in range(5)] + [0, 1, -3, 2**300 + 5]
    gamma = rng.randrange(1, group.order)
    want = rs.pivot.vector_commitment(x, gamma, g, h)
    vm.install(rs.package)
    got = rs.pivot.vector_commitment(x, gamma, g, h)
    assert isinstance(got, vm.Ed25519Point) and list(got.coords) == [c.value for c in want.value]
    assert repr(got) == repr(want)
    pv = vm.PointVector.from_points(g + [h])
    assert [list(p.coords) for p in pv.to_points()] == [[c.value for c in p.value] for p in g + [h]]
