"""Pins the oracle: RFC 8032 constants / known answers for the curve layer, and the golden
fixtures produced by the reference's own modules (tests/golden/make_fixtures.py) for the
protocol layer.  CPU only."""
import hashlib

import pytest

from oracle import ac20_ref as ac
from oracle import ed25519_ref as ed

h2i = lambda s: int(s, 16)
hx = lambda v: format(v, "x")


def test_rfc8032_constants():
    assert ed.BASE_X == 15112221349535400772501151409588531511454012693041857206046113283949847762202
    assert ed.BASE_Y == 46316835694926478169428394003475163141307993866256225615783033603165251855960
    assert ed.BASE_X % 2 == 0 and ed.on_curve(ed.BASE)
    assert ed.pt_eq(ed.pt_repeat(ed.BASE, ed.ELL), ed.IDENTITY)
    assert ed.encode_rfc8032(ed.BASE).hex() == "5866666666666666666666666666666666666666666666666666666666666666"


@pytest.mark.parametrize("seed,pk", [
    ("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
     "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"),
    ("4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
     "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c"),
    ("c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
     "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025"),
])
def test_rfc8032_section_7_1_public_keys(seed, pk):
    assert ed.rfc8032_public_key(bytes.fromhex(seed)).hex() == pk


def test_group_laws_and_negative_exponents():
    p, q = ed.pt_repeat(ed.BASE, 12345), ed.pt_repeat(ed.BASE, 99991)
    assert ed.pt_eq(ed.pt_add(p, q), ed.pt_repeat(ed.BASE, 12345 + 99991))
    assert ed.pt_eq(ed.pt_add(p, ed.pt_repeat(p, -1)), ed.IDENTITY)
    assert ed.pt_eq(ed.pt_repeat(p, ed.ELL + 5), ed.pt_repeat(p, 5))
    assert ed.pt_eq(ed.pt_dbl(p), ed.pt_add(p, p))
    xs = [ed.pt_repeat(ed.BASE, i + 1) for i in range(7)]
    assert ed.pt_eq(ed.tree_reduce(ed.pt_add, xs, ed.IDENTITY), ed.pt_repeat(ed.BASE, 28))
    assert ed.decode_rfc8032(ed.encode_rfc8032(p))[:2] == ed.pt_affine(p)


def test_forms_known_answers(golden_small):
    # ac20/test/test_pivot.py:84-90
    assert golden_small["forms"] == {"expr_27": 27, "expr_8": 8}
    assert ac.form_eval([0, 1, 2], 0, [1, 2, 3]) == 8


def run_case(case, raw=None):
    gens = ac.create_generators([h2i(e) for e in case["gen_exponents"]], h2i(case["gen_exponent_k"]))
    x = [h2i(v) for v in case["x"]]
    L = [h2i(v) for v in case["L"]]
    r = [h2i(v) for v in case["r"]]
    Lc = h2i(case.get("L_constant", "0"))
    P = ac.vector_commitment(x, h2i(case["gamma"]), gens["g"], gens["h"])
    assert [hx(c) for c in ed.pt_affine(P)] == case["P"]
    tr = {}
    proof = ac.protocol_5_prover(gens, P, L, Lc, h2i(case["y"]), x, h2i(case["gamma"]), r,
                                 h2i(case["rho"]), "reference", tr, raw)
    pr = case["proof"]
    assert hx(proof["t"]) == pr["t"]
    assert [hx(c) for c in ed.pt_affine(proof["A"])] == pr["A"]
    for i in range(case["rounds"]):
        assert [hx(c) for c in ed.pt_affine(proof[f"A{i}"])] == pr["A_i"][i]
        assert [hx(c) for c in ed.pt_affine(proof[f"B{i}"])] == pr["B_i"][i]
    assert [hx(v) for v in proof["z_prime"]] == pr["z_prime"]
    assert [hx(c) for c in [tr["c0"], tr["c1"]] + tr["c"]] == [h["c"] for h in case["hashes"]]
    assert ac.protocol_5_verifier(gens, P, L, Lc, h2i(case["y"]), proof, "reference", raw) is True
    return gens, proof


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
def test_protocol5_fixture(golden_small, idx):
    case = golden_small["p5"][idx]
    gens, _ = run_case(case)
    raw = b"".join(ed.proj_to_bytes(p) for p in gens["g"] + [gens["h"], gens["k"]])
    assert hashlib.sha256(raw).hexdigest() == case["gens_proj_sha256"]


def test_first_preimage_text(golden_small):
    case = golden_small["p5"][0]
    gens = ac.create_generators([h2i(e) for e in case["gen_exponents"]], h2i(case["gen_exponent_k"]))
    P = (h2i(case["P"][0]), h2i(case["P"][1]), 1)
    A = (h2i(case["proof"]["A"][0]), h2i(case["proof"]["A"][1]), 1)
    t0, t1 = ac.p5_hash_texts(h2i(case["proof"]["t"]), A, gens, P, [h2i(v) for v in case["L"]], 0,
                              h2i(case["y"]))
    assert [t0, t1] == [h["text"] for h in case["hashes"][:2]]


def test_demo_fixture(golden_demo):
    """BASELINE config 1 (demo_zkp_ac20.py --elliptic, N = 128) with the demo's int-typed form."""
    raw = [int(v[2:]) if v[0] == "i" else None for v in golden_demo["L_typed"]]
    run_case(golden_demo, raw)
    assert golden_demo["verification"] == {"y1*y2=y3": True, "L_wellformed_from_Cfgh_forms": True,
                                           "pivot_verification": True}
    assert golden_demo["n"] == 127 and golden_demo["rounds"] == 6


def test_basic_pivot_fixture(golden_small):
    case = golden_small["pis"][0]
    g = ac.create_generators([h2i(e) for e in case["gen_exponents"]])["g"]
    x = [h2i(v) for v in case["x"]]
    L = [h2i(v) for v in case["L"]]
    P = ac.vector_commitment(x, h2i(case["gamma"]), g, ed.BASE)
    z, phi, c = ac.prove_linear_form_eval(g, ed.BASE, P, L, 0, h2i(case["y"]), x, h2i(case["gamma"]),
                                          [h2i(v) for v in case["r"]], h2i(case["rho"]))
    assert [hx(v) for v in z] == case["z"] and hx(phi) == case["phi"] and hx(c) == case["c"]
    assert ac.verify_linear_form_proof(g, ed.BASE, P, L, 0, h2i(case["y"]), z, phi, c)


def test_compact_mode_roundtrip_and_soundness():
    import random
    rng = random.Random(3)
    n = 7
    gens = ac.create_generators([rng.randrange(1, ed.ELL) for _ in range(n)], rng.randrange(1, ed.ELL))
    x = [rng.randrange(ed.ELL) for _ in range(n)]
    L = [rng.randrange(ed.ELL) for _ in range(n)]
    gamma = rng.randrange(ed.ELL)
    P = ac.vector_commitment(x, gamma, gens["g"], gens["h"])
    y = ac.form_eval(L, 0, x)
    proof = ac.protocol_5_prover(gens, P, L, 0, y, x, gamma, [rng.randrange(ed.ELL) for _ in range(n)],
                                 rng.randrange(ed.ELL), "compact")
    assert ac.protocol_5_verifier(gens, P, L, 0, y, proof, "compact")
    assert not ac.protocol_5_verifier(gens, P, L, 0, (y + 1) % ed.ELL, proof, "compact")
    assert not ac.protocol_5_verifier(gens, P, L, 0, y, proof, "reference")
