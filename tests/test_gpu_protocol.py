"""GPU parity of the drop-in Protocol 4/5 (verifiable_mpc_amd.compressed_pivot) against
 (1) the golden fixtures captured from the reference's own modules
     (tests/golden/make_fixtures.py), and
 (2) the oracle restatement run on the same seeded inputs.
Bit-exact: commitments, every A_i/B_i (affine), t, z', and every Fiat-Shamir challenge
(i.e. the pre-image text itself, including the un-normalised folded generators).
"""
import hashlib
import random

import pytest

from oracle import ac20_ref as ac
from oracle import ed25519_ref as ed

pytestmark = pytest.mark.gpu

ELL = ed.ELL
hx = lambda v: format(int(v), "x")
h2i = lambda s: int(s, 16)


@pytest.fixture(scope="module")
def vm():
    import verifiable_mpc_amd as v
    v.get_context()
    return v


@pytest.fixture()
def record_hashes(vm, monkeypatch):
    calls = []
    orig = vm.pivot.fiat_shamir_hash

    def wrapped(input_list, order):
        c = orig(input_list, order)
        calls.append(c)
        return c
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash", wrapped)
    orig_v = vm.pivot.fiat_shamir_hash_variants

    def wrapped_v(common, tails, order):
        cs = orig_v(common, tails, order)
        calls.extend(cs)
        return cs
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash_variants", wrapped_v)
    return calls


def aff_hex(pt):
    n = pt.normalize()
    return [hx(n.coords[0]), hx(n.coords[1])]


def build_generators(vm, case, monkeypatch, seed):
    """create_generators with the reference's draw order replayed from the fixture seed."""
    group = vm.EllipticCurve("Ed25519", "projective")
    monkeypatch.setattr(vm.circuit_sat, "prng", random.Random(seed))
    gens = vm.create_generators(case["n"], vm.PivotChoice.compressed, group)
    return group, gens


def check_proof(case, proof, hashes):
    pr = case["proof"]
    assert hx(int(proof["t"]) % ELL) == pr["t"]
    assert aff_hex(proof["A"]) == pr["A"]
    for i in range(case["rounds"]):
        assert aff_hex(proof[f"A{i}"]) == pr["A_i"][i], f"A{i}"
        assert aff_hex(proof[f"B{i}"]) == pr["B_i"][i], f"B{i}"
    assert [hx(int(v) % ELL) for v in proof["z_prime"]] == pr["z_prime"]
    assert [hx(c) for c in hashes] == [h["c"] for h in case["hashes"]]


@pytest.mark.parametrize("idx", [0, 1, 2, 3])
@pytest.mark.parametrize("device_mode", [False, True])
def test_protocol5_matches_reference_fixture(vm, golden_small, monkeypatch, record_hashes, idx,
                                             device_mode):
    case = golden_small["p5"][idx]
    n = case["n"]
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    gf = vm.GF(group.order)
    # generator representatives are the reference's (projective, un-normalised)
    pts = gens["g"].to_points() + [gens["h"], gens["k"]]
    raw = b"".join(p.to_proj_bytes() for p in pts)
    assert hashlib.sha256(raw).hexdigest() == case["gens_proj_sha256"]
    if "gens_proj" in case:
        assert [[hx(c) for c in p.coords] for p in pts] == case["gens_proj"]

    x = [gf(h2i(v)) for v in case["x"]]
    coeffs = [gf(h2i(v)) for v in case["L"]]
    gamma, y = h2i(case["gamma"]), gf(h2i(case["y"]))
    if device_mode:
        x = vm.ScalarVector.from_ints(x)
        L = vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
    else:
        L = vm.pivot.LinearForm(coeffs)
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    assert aff_hex(P) == case["P"]

    monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(case["seed"] + 2))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf)
    prover_hashes = list(record_hashes)
    check_proof(case, proof, prover_hashes)

    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf) is True
    assert record_hashes[len(prover_hashes):] == prover_hashes
    bad = dict(proof)
    bad["z_prime"] = [proof["z_prime"][0] + 1, proof["z_prime"][1]]
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, bad, gf) is False
    bad = dict(proof)
    bad["A0"] = proof["B0"]
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, bad, gf) is False


def test_protocol5_first_preimage_text(vm, golden_small, monkeypatch):
    """the streamed pre-image equals the reference's str(input_list) byte for byte (N = 4)."""
    case = golden_small["p5"][0]
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    gf = vm.GF(group.order)
    texts = []

    class Spy:
        def __init__(self, h=None, buf=b""):
            self.h = h or hashlib.sha256()
            self.buf = bytearray(buf)

        def copy(self):
            return Spy(self.h.copy(), self.buf)

        def update(self, b):
            self.buf += bytes(b)
            self.h.update(b)

        def digest(self):
            texts.append(bytes(self.buf).decode())
            return self.h.digest()
    import types
    monkeypatch.setattr(vm.pivot, "hashlib", types.SimpleNamespace(sha256=lambda: Spy()))
    x = [gf(h2i(v)) for v in case["x"]]
    L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
    P = vm.Ed25519Point((h2i(case["P"][0]), h2i(case["P"][1]), 1))
    monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(case["seed"] + 2))
    vm.compressed_pivot.protocol_5_prover(gens, P, L, gf(h2i(case["y"])), x, h2i(case["gamma"]), gf)
    assert texts == [h["text"] for h in case["hashes"]]


def test_hash_input_dump_is_what_gets_hashed(vm, golden_small, monkeypatch):
    """the DEBUG dump of logger "compressed_pivot_hash_inputs" (compressed_pivot.py:56-58,122-124) holds the very
    pre-images of the reference fixture: Protocol 5's list once (the reference logs it before appending the 0 / 1
    tails), then one list per round, for prover and verifier"""
    import logging
    case = golden_small["p5"][0]
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    gf = vm.GF(group.order)
    lg = logging.getLogger("compressed_pivot_hash_inputs")
    records = []

    class Keep(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())
    handler = Keep()
    lg.addHandler(handler)
    lg.setLevel(logging.DEBUG)
    try:
        x = [gf(h2i(v)) for v in case["x"]]
        L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
        P = vm.Ed25519Point((h2i(case["P"][0]), h2i(case["P"][1]), 1))
        monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(case["seed"] + 2))
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, gf(h2i(case["y"])), x, h2i(case["gamma"]), gf)
        n_prover = len(records)
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, gf(h2i(case["y"])), proof, gf) is True
    finally:
        lg.setLevel(logging.INFO)
        lg.removeHandler(handler)
    texts = [h["text"] for h in case["hashes"]]
    tail = ", 0, 'First hash of compressed pivot']"
    assert texts[0].endswith(tail)
    want = [texts[0][:-len(tail)] + "]"] + texts[2:]
    got = [r.split("input_list=\n", 1)[1] for r in records]
    assert got[:n_prover] == want and got[n_prover:] == want
    assert records[0].startswith("Method protocol_5_prover: Before fiat_shamir_hash, input_list=")
    assert records[n_prover].startswith("Method protocol_5_verifier:")
    assert records[n_prover + 1].startswith("Method protocol_4_verifier:")


@pytest.mark.parametrize("early_pair", [False, True, "table_rounds", "table_rounds_wide", "table_rounds_main_stream"])
def test_protocol5_n1023_device_mode(vm, golden_n1023, monkeypatch, record_hashes, early_pair):
    """N = 1024 in device mode against the reference-made fixture; also with the next round's pair committed over the
    unfolded vector beside the fold (compressed_pivot._early_pair_*, an experiment that is off by default), and with
    the big rounds' pairs taken from the round context over the tabulated CRS ahead of the exact folds (round 6:
    REF_TABLE_PAIR_MIN, what a 2^20-generator CRS from create_generators does) - every challenge of the reference's
    own run either way"""
    case = golden_n1023
    monkeypatch.setattr(vm.compressed_pivot, "EARLY_PAIR_MIN", 64 if early_pair is True else 0)
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    if isinstance(early_pair, str):
        monkeypatch.setattr(vm.compressed_pivot, "REF_TABLE_PAIR_MIN", 16)
        monkeypatch.setenv("VMPC_P4_COMMIT_TABLE_MIN_LOG2", "0")
        # (the context on a stream of its own - round_begin / round_end beside the exact fold - or on the main one)
        monkeypatch.setattr(vm.compressed_pivot, "REF_TABLE_PAIR_SIDE_STREAM", not early_pair.endswith("main_stream"))
        gens["g"].precompute([gens["h"], gens["k"]], wide=early_pair.endswith("wide"))
        made = []
        for name in ("round", "round_begin"):
            def spy(self, c=None, orig=getattr(vm._native.P4Rounds, name)):
                made.append(c)
                return orig(self, c)
            monkeypatch.setattr(vm._native.P4Rounds, name, spy)
    gf = vm.GF(group.order)
    x = vm.ScalarVector.from_ints([h2i(v) for v in case["x"]])
    L = vm.pivot.LinearForm(vm.ScalarVector.from_ints([h2i(v) for v in case["L"]]))
    gamma, y = h2i(case["gamma"]), gf(h2i(case["y"]))
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    assert aff_hex(P) == case["P"]
    monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(case["seed"] + 2))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf)
    hashes = list(record_hashes)
    check_proof(case, proof, hashes)
    if isinstance(early_pair, str):
        assert len(made) == 6 and made[0] is None       # rounds 0 .. 5 (vectors of 1024 .. 32 elements) from the context
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf) is True


def typed_value(gf, s):
    kind, v = s.split(":")
    return int(v) if kind == "i" else gf(int(v, 16))


def test_demo_zkp_ac20_elliptic_protocol5_call(vm, golden_demo, monkeypatch, record_hashes):
    """BASELINE config 1: the Protocol-5 call made by demos/demo_zkp_ac20.py --elliptic
    (N = 128), with the demo's exact Python typing of x and L (plain ints stay unreduced in
    the pre-image, pivot.py:62-70)."""
    case = golden_demo
    group, gens = build_generators(vm, case, monkeypatch, 20200152 + 600 + 10)
    gf = vm.GF(group.order)
    x = [typed_value(gf, s) for s in case["x_typed"]]
    L = vm.pivot.AffineForm([typed_value(gf, s) for s in case["L_typed"]],
                            typed_value(gf, case["L_constant_typed"]))
    y = typed_value(gf, case["y_typed"])
    gamma = h2i(case["gamma"])
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    assert aff_hex(P) == case["P"]
    monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(20200152 + 600 + 12))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf)
    check_proof(case, proof, list(record_hashes))
    # the demo's verifier call passes y = 0 and the un-shifted form (circuit_sat_cb.py:297-299)
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, 0, proof, gf) is True


def _typed_of(v):
    return "i:" + str(v) if isinstance(v, int) else "f:" + hx(int(v) % ELL)


def proj_hex(pt):
    return [hx(c) for c in pt.coords]


def test_demo_zkp_ac20_whole_transcript(vm, golden_demo, monkeypatch, record_hashes):
    """BASELINE config 1 at the Protocol-8 level: every Fiat-Shamir hash of the seeded
    `demo_zkp_ac20.py --elliptic` run - the two of circuit_sat_cb.py:107-111 / :149-162, which contain
    the UN-normalised commitment [z], then c0, c1 and the six round challenges - and the proof dict the
    demo prints, points with their (X:Y:Z) representatives and scalars with their Python types.

    The circuit front end is out of scope and absent from the GPU box, so the objects it feeds into
    the hashes (str(circuit), the circuit / f / g / h forms) come from the fixture as typed
    coefficient lists and are rebuilt with THIS package's classes; everything on the hot path -
    [z] (pivot.py:139-145 through the exact list-mode path), the hashes, L = sum lin_form_i * rho^i
    (circuit_sat_cb.py:164), Protocol 5 - is computed here."""
    case, p8, ret = golden_demo, golden_demo["protocol8"], golden_demo["returned_proof"]
    group, gens = build_generators(vm, case, monkeypatch, 20200152 + 600 + 10)
    gf = vm.GF(group.order)
    tv = lambda s_: typed_value(gf, s_)

    def form(rec):
        cls = vm.pivot.LinearForm if rec["linear"] else vm.pivot.AffineForm
        return cls([tv(c) for c in rec["coeffs"]], tv(rec["constant"]))

    # circuit_sat_cb.py:91-104: z and its commitment (the caller passes Python lists)
    z = [tv(v) for v in p8["z_typed"]]
    gamma = h2i(p8["gamma"])
    z_commitment = vm.pivot.vector_commitment(z, gamma, list(gens["g"]), gens["h"])
    assert proj_hex(z_commitment) == p8["z_commitment_proj"]              # representative, not just the element
    assert aff_hex(z_commitment) == case["P"]
    # first hash (:107-111)
    c = vm.pivot.fiat_shamir_hash(
        [z_commitment, p8["circuit_str"], "First hash circuit satisfiability protocol"], gf.order)
    assert hx(c) == p8["hashes"][0]["c"]
    # second hash (:149-162)
    y1, y2, y3 = (tv(v) for v in p8["y_typed"])
    outputs = [tv(v) for v in p8["outputs_typed"]]
    circuit_forms = [form(f) for f in p8["circuit_forms"]]
    lin_forms = [form(f) for f in p8["lin_forms"]]
    rho = vm.pivot.fiat_shamir_hash([y1, y2, y3, z_commitment, outputs, circuit_forms, lin_forms,
                                     "Second hash circuit satisfiability protocol"], gf.order)
    assert hx(rho) == p8["hashes"][1]["c"]
    # L (:164): the same expression over this package's form classes
    L = sum((linform_i) * (rho ** i) for i, linform_i in enumerate(lin_forms))
    assert [_typed_of(v) for v in L.coeffs] == p8["L"]["coeffs"] == case["L_typed"]
    assert _typed_of(L.constant) == p8["L"]["constant"]
    # circuit_sat_cb.py:263-266
    monkeypatch.setattr(vm.compressed_pivot, "prng", random.Random(20200152 + 600 + 12))
    proof = vm.compressed_pivot.protocol_5_prover(gens, z_commitment, L, L(z), z, gamma, gf)
    assert [hx(v) for v in record_hashes] == [h["c"] for h in case["all_hashes"][:10]]
    assert list(proof.keys()) == ret["pivot_proof_keys"]
    assert _typed_of(proof["t"]) == ret["t_typed"]
    assert proj_hex(proof["A"]) == ret["A_proj"]
    for i in range(case["rounds"]):
        assert proj_hex(proof[f"A{i}"]) == ret["A_i_proj"][i], f"A{i}"
        assert proj_hex(proof[f"B{i}"]) == ret["B_i_proj"][i], f"B{i}"
    assert [_typed_of(v) for v in proof["z_prime"]] == ret["z_prime_typed"]
    # verifier (:285-318): recomputes the same ten hashes; y = 0 and the un-shifted form
    del record_hashes[:]
    cv = vm.pivot.fiat_shamir_hash(
        [z_commitment, p8["circuit_str"], "First hash circuit satisfiability protocol"], gf.order)
    rv = vm.pivot.fiat_shamir_hash([y1, y2, y3, z_commitment, outputs, circuit_forms, lin_forms,
                                    "Second hash circuit satisfiability protocol"], gf.order)
    assert (cv, rv) == (c, rho)
    assert vm.compressed_pivot.protocol_5_verifier(gens, z_commitment, L, 0, proof, gf) is True
    assert [hx(v) for v in record_hashes] == [h["c"] for h in case["all_hashes"][10:]]


@pytest.mark.parametrize("n", [3, 31])
def test_compact_transcript_matches_oracle(vm, monkeypatch, n):
    rng = random.Random(500 + n)
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    ek = rng.randrange(1, ELL)
    ogens = ac.create_generators(exps, ek)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    gens = {"g": vm.PointVector.fixed_base(group.generator, exps, keep_proj=False),
            "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, ek)}
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    gamma = rng.randrange(1, ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    rho = rng.randrange(ELL)
    oP = ac.vector_commitment(x, gamma, ogens["g"], ogens["h"])
    oy = ac.form_eval(coeffs, 0, x)
    want = ac.protocol_5_prover(ogens, oP, coeffs, 0, oy, x, gamma, r, rho, "compact")
    xs, Lf = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
    P = vm.pivot.vector_commitment(xs, gamma, gens["g"], gens["h"])
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, gf(oy), xs, gamma, gf,
                                                  transcript="compact", r=r, rho=rho)
    rounds = (n + 1).bit_length() - 2
    assert int(proof["t"]) % ELL == want["t"]
    for key in ["A"] + [f"A{i}" for i in range(rounds)] + [f"B{i}" for i in range(rounds)]:
        assert tuple(proof[key].normalize().coords[:2]) == ed.pt_affine(want[key]), key
    assert [int(v) % ELL for v in proof["z_prime"]] == want["z_prime"]
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, gf(oy), proof, gf,
                                                   transcript="compact") is True
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, gf(oy + 1), proof, gf,
                                                   transcript="compact") is False


def test_precomputed_generators_give_identical_proofs(vm):
    """PointVector.precompute (fixed-base tables) changes how commitments are computed, not what
    they are: commitments and whole Protocol-5 proofs are identical with and without it."""
    rng = random.Random(808)
    n = 63
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    results = []
    for pre in (False, 16, 2, 1):
        g = vm.PointVector.fixed_base(h, exps, keep_proj=True)
        if pre:
            g.precompute([h, k], rows=pre)
            assert g[:10]._table is g._table and g[1:]._table is None
        gens = {"g": g, "h": h, "k": k}
        xs, Lf = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
        P = vm.pivot.vector_commitment(xs, gamma, g, h)
        Pk = vm.pivot.vector_commitment(xs[:20], gamma, g, k)          # prefix, other base point
        y = gf(Lf(xs))
        proofs = {}
        for mode in ("compact", "reference"):
            proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript=mode,
                                                          r=list(r), rho=rho)
            assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript=mode) is True
            proofs[mode] = {key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                                  [int(e) for e in v] if isinstance(v, list) else int(v))
                            for key, v in proof.items()}
        results.append((tuple(P.normalize().coords), tuple(Pk.normalize().coords), proofs))
    assert results[0] == results[1] == results[2] == results[3]


@pytest.mark.parametrize("log_n", [2, 5, 12])
def test_native_round_context_equals_python_driven_rounds(vm, monkeypatch, log_n):
    """vmpc_p4_* (csrc/prover.hip: rounds resident on the device) against the same compact prover with the
    rounds driven from Python: identical proofs, and the used-up context refuses further rounds."""
    rng = random.Random(900 + log_n)
    n = (1 << log_n) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n)])
    g.precompute([h, k])
    gens = {"g": g, "h": h, "k": k}
    xs = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)])
    Lf = vm.pivot.LinearForm(vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)]))
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    P = vm.pivot.vector_commitment(xs, gamma, g, h)
    y = gf(Lf(xs))
    proofs = []
    # the chain of challenges inside the C call / one C call per round / every round driven from Python
    for native, chain in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(vm.compressed_pivot, "NATIVE_ROUNDS", native)
        monkeypatch.setattr(vm.compressed_pivot, "NATIVE_CHAIN", chain)
        calls = []
        real, real_run = vm._native.P4Rounds.round, vm._native.P4Rounds.run_compact
        monkeypatch.setattr(vm._native.P4Rounds, "round", lambda self, c=None: calls.append(1) or real(self, c))
        monkeypatch.setattr(vm._native.P4Rounds, "run_compact",
                            lambda self, *a: calls.append(100) or real_run(self, *a))
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                      r=list(r), rho=rho)
        monkeypatch.setattr(vm._native.P4Rounds, "round", real)
        monkeypatch.setattr(vm._native.P4Rounds, "run_compact", real_run)
        assert sum(calls) == (100 if chain else (log_n - 1) if native else 0)
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript="compact") is True
        proofs.append({key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                             [int(e) for e in v] if isinstance(v, list) else int(v)) for key, v in proof.items()})
    assert proofs[0] == proofs[1] == proofs[2]
    assert proofs[0] == proofs[1]


@pytest.mark.parametrize("log_n,jump_k,min_log2", [(8, 5, 6), (8, 2, 4), (9, 6, 7), (6, 1, 3), (12, 5, 5), (7, 3, 7),
                                                   (5, 3, 5), (12, 4, 12)])
def test_native_round_context_with_fold_jumps(vm, monkeypatch, log_n, jump_k, min_log2):
    """The round context applies the pending challenges to the generators after `jump_k` rounds
    (vmpc_msm_table_fold_dev + a table for the folded vector) - once or repeatedly, down to a 4-element base:
    the proof does not change."""
    rng = random.Random(1900 + log_n + jump_k)
    n = (1 << log_n) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n)])
    g.precompute([h, k], rows=rng.choice([1, 4, 16]))
    gens = {"g": g, "h": h, "k": k}
    xs = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)])
    Lf = vm.pivot.LinearForm(vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)]))
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    P = vm.pivot.vector_commitment(xs, gamma, g, h)
    y = gf(Lf(xs))
    proofs = []
    for setting in ((str(jump_k), str(min_log2)), ("0", "30")):
        monkeypatch.setenv("VMPC_P4_JUMP", setting[0])
        monkeypatch.setenv("VMPC_P4_JUMP_MIN_LOG2", setting[1])
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                      r=list(r), rho=rho)
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript="compact") is True
        proofs.append({key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                             [int(e) for e in v] if isinstance(v, list) else int(v)) for key, v in proof.items()})
    assert proofs[0] == proofs[1]


@pytest.mark.parametrize("log_n,jump_k,min_log2", [(3, 0, 30), (9, 3, 5), (13, 5, 8)])
def test_rounds_queued_ahead_of_their_challenge(vm, monkeypatch, log_n, jump_k, min_log2):
    """vmpc_p4_run_compact queues round i + 1 behind a stream wait while round i runs and hands it the challenge
    through pinned memory (csrc/prover.hip); with VMPC_P4_NO_QUEUE_AHEAD every round is launched after its
    challenge.  Same proofs, proof after proof on one context (the mailbox's sequence numbers keep counting),
    through fold jumps and the bucket-free short rounds."""
    rng = random.Random(7700 + log_n)
    n = (1 << log_n) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n)], keep_proj=False)
    g.precompute([h, k])
    gens = {"g": g, "h": h, "k": k}
    Lf = vm.pivot.LinearForm(vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)]))
    monkeypatch.setenv("VMPC_P4_JUMP", str(jump_k))
    monkeypatch.setenv("VMPC_P4_JUMP_MIN_LOG2", str(min_log2))
    monkeypatch.setenv("VMPC_EXPERIMENTAL", "1")
    for rep in range(3):
        xs = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)])
        gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
        r = [rng.randrange(ELL) for _ in range(n)]
        P = vm.pivot.vector_commitment(xs, gamma, g, h)
        y = gf(Lf(xs))
        proofs = []
        for plain in (False, True, False):
            if plain:
                monkeypatch.setenv("VMPC_P4_NO_QUEUE_AHEAD", "1")
            else:
                monkeypatch.delenv("VMPC_P4_NO_QUEUE_AHEAD", raising=False)
            proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                          r=list(r), rho=rho)
            proofs.append({key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                                 [int(e) for e in v] if isinstance(v, list) else int(v)) for key, v in proof.items()})
        assert proofs[0] == proofs[1] == proofs[2]
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript="compact") is True


def test_nothing_grows_while_a_wait_is_queued(vm):
    """The invariant the queued-ahead rounds rest on (csrc/prover.hip): with a stream wait pending that only the calling
    thread can release, a call that would have to grow the context's workspace or its pinned block - both synchronise
    the stream to do so - returns an error instead of hanging.  The state is set through the library's test hook."""
    import numpy as np
    from verifiable_mpc_amd import _native
    ctx = _native.Context(0)
    try:
        lib = ctx.lib
        n = 1 << 12
        pts = ctx.alloc(64 * n)
        base = ctx.upload(np.frombuffer(vm.Ed25519Point.generator.to_affine_bytes(), np.uint8))
        sc = ctx.upload(_native.ints_to_array(list(range(1, n + 1)), 32))
        ctx.fixed_base(base.ptr, sc.ptr, n, pts.ptr)
        out = ctx.alloc(128)
        ctx.msm(sc.ptr, pts.ptr, 64, None, None, 0, out.ptr, None)          # a small workspace exists now
        ctx.sync()
        assert lib.vmpc_ctx_debug_hold_wait(ctx.handle, 1) == 0
        with pytest.raises(_native.VmpcError) as ei:
            ctx.msm(sc.ptr, pts.ptr, n, None, None, 0, out.ptr, None)       # 64x the terms: the arena has to grow
        assert ei.value.code == _native.E_INVAL
        assert b"would grow while the stream waits" in lib.vmpc_last_error()
        assert lib.vmpc_ctx_debug_hold_wait(ctx.handle, 0) == 0
        ctx.msm(sc.ptr, pts.ptr, n, None, None, 0, out.ptr, None)           # and afterwards it simply grows
        ctx.sync()
    finally:
        ctx.lib.vmpc_ctx_debug_hold_wait(ctx.handle, 0)
        ctx.close()


@pytest.mark.parametrize("n_table,with_h", [(128, False), (127, True)])
def test_protocol4_over_a_prefix_of_a_tabulated_crs(vm, n_table, with_h):
    """protocol_4_prover(g[:m], ...) with g[:m] a STRICT prefix of a tabulated vector (PointVector slices keep
    the table): the round context derives N from the table, so a prefix must take the round-by-round path.
    Same proof as over an untabulated copy of the same 64 points."""
    rng = random.Random(4711 + n_table)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n_table)], keep_proj=False)
    g.precompute([h, k] if with_h else [k])
    m = 64
    prefix = g[:m]
    assert prefix._table is g._table and len(prefix) == m
    plain = vm.PointVector.from_affine_array(prefix.affine_array())
    assert plain._table is None
    z = [rng.randrange(ELL) for _ in range(m)]
    lc = [rng.randrange(ELL) for _ in range(m)]
    proofs = []
    for gv in (prefix, plain):
        z_hat = vm.ScalarVector.from_ints(z)
        L_tilde = vm.pivot.LinearForm(vm.ScalarVector.from_ints(lc))
        Q = vm.pivot.vector_commitment(z_hat, int(L_tilde(z_hat)), gv, k)
        tr = vm.compressed_pivot._Transcript("compact", group.order, hashlib.sha256(b"prefix test").digest())
        proof = vm.compressed_pivot.protocol_4_prover(gv, k, Q, L_tilde, z_hat, gf, {}, transcript=tr)
        tr = vm.compressed_pivot._Transcript("compact", group.order, hashlib.sha256(b"prefix test").digest())
        assert vm.compressed_pivot.protocol_4_verifier(plain, k, Q, L_tilde, gf, proof, transcript=tr) is True
        proofs.append({key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else [int(e) for e in v])
                       for key, v in proof.items()})
    assert proofs[0] == proofs[1]
    # the binding itself refuses a table that is longer than the vectors it is handed
    with pytest.raises(AssertionError):
        vm._native.P4Rounds(g.ctx, g._table, 0, g._table.extra_index(k), z_hat.ptr, z_hat.ptr, n_total=m)


@pytest.mark.parametrize("log_n,direct_log2,jump_k,min_log2,rows", [
    (10, 5, 5, 30, 16),     # small vector: one fold straight down to 2^5 generators, then bucket-free commitments
    (12, 6, 3, 9, 16),      # two folds by the large-block rule (2^12 -> 2^9 -> 2^6), bucket-free from there
    (13, 11, 5, 30, 8),     # the default threshold: 2^13 -> 2^11 after two rounds
    (7, 11, 0, 30, 16),     # no folds at all: bucket-free from round 0 on the CRS's own table (h among the extras)
    (7, 11, 0, 30, 8),      # ... with 32-bit digits (8-row table)
    (6, 11, 5, 30, 4),      # a 4-row table keeps the bucket method (64-bit digits would be a 64-step ladder)
])
def test_bucket_free_short_rounds(vm, monkeypatch, log_n, direct_log2, jump_k, min_log2, rows):
    """csrc/prover.hip k_p4_direct: once the vector is short, A_i / B_i are computed without buckets (one lane per
    generator and table row) and the context folds down to that size as soon as it can.  Same proofs as with
    every commitment by the bucket method and no fold."""
    rng = random.Random(7700 + 13 * log_n + direct_log2)
    n = (1 << log_n) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n)], keep_proj=False)
    g.precompute([h, k], rows=rows)
    gens = {"g": g, "h": h, "k": k}
    # the witness distribution of the demo circuit: mostly zeros and small values, masks likewise,
    # so that whole digits - and whole scalars - of the round's commitment scalars are zero
    small = lambda: rng.choice([0, 0, 0, 1, 2, ELL - 1, rng.randrange(ELL)])
    xs = vm.ScalarVector.from_ints([small() for _ in range(n)])
    Lf = vm.pivot.LinearForm(vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)]))
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [0 if rng.random() < 0.6 else rng.randrange(ELL) for _ in range(n)]
    P = vm.pivot.vector_commitment(xs, gamma, g, h)
    y = gf(Lf(xs))
    proofs, stages = [], []
    ctx = vm.get_context()
    for setting in ((str(direct_log2), str(jump_k), str(min_log2)), ("0", "0", "30")):
        monkeypatch.setenv("VMPC_P4_DIRECT_LOG2", setting[0])
        monkeypatch.setenv("VMPC_EXPERIMENTAL", "1")            # measured losers are only read behind this switch
        monkeypatch.setenv("VMPC_P4_FOLD_TO_DIRECT", "1")       # the fold-down-to-the-short-form rule (off by default)
        monkeypatch.setenv("VMPC_P4_JUMP", setting[1])
        monkeypatch.setenv("VMPC_P4_JUMP_MIN_LOG2", setting[2])
        ctx.profile(True)
        ctx.profile_read(reset=True)
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                      r=list(r), rho=rho)
        stages.append({name: cnt for name, (ms, cnt) in ctx.profile_read(reset=True).items() if cnt})
        ctx.profile(False)
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript="compact") is True
        proofs.append({key: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                             [int(e) for e in v] if isinstance(v, list) else int(v)) for key, v in proof.items()})
    assert proofs[0] == proofs[1]
    assert "p4_direct" not in stages[1] and "table_fold" not in stages[1]
    if rows >= 8:
        assert stages[0].get("p4_direct", 0) >= 1            # the bucket-free kernel really ran
    else:
        assert "p4_direct" not in stages[0]


def test_two_round_contexts_on_one_vmpc_ctx(vm):
    """The first context takes the arena pooled in the vmpc_ctx, a second one alive at the same time gets a
    private one (csrc/prover.hip); interleaved rounds give what each gives alone, and the pool is free again
    afterwards."""
    ctx = vm.get_context()
    group = vm.EllipticCurve("Ed25519", "projective")
    h = group.generator
    k = vm.Ed25519Point.repeat(h, 4242)
    g = vm.PointVector.fixed_base(h, list(range(3, 34)))           # 31 generators + h = 32
    g.precompute([h, k])
    z1 = vm.ScalarVector.from_ints(list(range(1, 33)))
    z2 = vm.ScalarVector.from_ints(list(range(101, 133)))
    L = vm.ScalarVector.from_ints(list(range(7, 39)))

    def alone(z):
        r = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, L.ptr)
        out = [r.round(None), r.round(11), r.round(12), r.round(13)]
        out.append(r.finish(14))
        r.close()
        return out

    want1, want2 = alone(z1), alone(z2)
    a = vm._native.P4Rounds(ctx, g._table, 1, 1, z1.ptr, L.ptr)
    b = vm._native.P4Rounds(ctx, g._table, 1, 1, z2.ptr, L.ptr)
    got1, got2 = [], []
    for c in (None, 11, 12, 13):
        got1.append(a.round(c))
        got2.append(b.round(c))
    got2.append(b.finish(14))
    got1.append(a.finish(14))
    b.close()
    a.close()
    assert got1 == want1 and got2 == want2 and want1 != want2
    assert alone(z1) == want1


def test_native_round_in_two_halves(vm):
    """vmpc_p4_round_begin + vmpc_p4_round_end = vmpc_p4_round; between the halves the context refuses every other
    call (and a fold of its generators that is due: vmpc_p4_prefold)"""
    ctx = vm.get_context()
    group = vm.EllipticCurve("Ed25519", "projective")
    h = group.generator
    k = vm.Ed25519Point.repeat(h, 4242)
    g = vm.PointVector.fixed_base(h, list(range(3, 34)))           # 31 generators + h = 32
    g.precompute([h, k])
    z = vm.ScalarVector.from_ints(list(range(1, 33)))
    L = vm.ScalarVector.from_ints(list(range(7, 39)))
    whole = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, L.ptr)
    want = [whole.round(None), whole.round(11), whole.round(12), whole.round(13), whole.finish(14)]
    whole.close()
    r = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, L.ptr)
    with pytest.raises(vm._native.VmpcError):
        r.round_end()                                  # nothing in flight
    got = []
    for i, c in enumerate((None, 11, 12, 13)):
        r.round_begin(c)
        for bad in (lambda: r.round_begin(5), lambda: r.round(5), r.prefold, lambda: r.finish(14)):
            with pytest.raises(vm._native.VmpcError):
                bad()
        got.append(r.round_end())
        r.prefold()                                    # (nothing due on a 32-element vector: a no-op)
    got.append(r.finish(14))
    r.close()
    assert got == want
    # a context destroyed with a round in flight
    r = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, L.ptr)
    r.round_begin(None)
    r.close()
    ctx.sync()


@pytest.mark.parametrize("log_n", [6, 9])
def test_lazy_generator_fold_gives_the_same_rounds(vm, monkeypatch, log_n):
    """vmpc_p4_create_opts(lazy_fold): the round that is given the jump_k-th challenge commits over the UNFOLDED table
    and the fold is made by vmpc_p4_prefold (or at the start of the round after) - the same A_i, B_i as the eager fold
    and as a context that never folds"""
    monkeypatch.setenv("VMPC_P4_JUMP_MIN_LOG2", "3")
    ctx = vm.get_context()
    group = vm.EllipticCurve("Ed25519", "projective")
    h = group.generator
    k = vm.Ed25519Point.repeat(h, 4242)
    n = (1 << log_n) - 1
    rng = random.Random(log_n)
    g = vm.PointVector.fixed_base(h, [rng.randrange(1, 2**252) for _ in range(n)])
    g.precompute([h, k])
    z = vm.ScalarVector.from_ints([rng.randrange(2**252) for _ in range(n + 1)])
    L = vm.ScalarVector.from_ints([rng.randrange(2**252) for _ in range(n + 1)])
    cs = [None] + [rng.randrange(1, 2**252) for _ in range(log_n - 2)]

    def rounds(prefold_after=None, **kw):
        r = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, L.ptr, **kw)
        out = []
        for i, c in enumerate(cs):
            out.append(r.round(c))
            if prefold_after is not None and i == prefold_after:
                r.prefold()
        out.append(r.finish(12345))
        r.close()
        return out

    want = rounds(jump_k=0)                                   # never folds
    assert rounds(jump_k=2) == want                           # folds at the start of the round given c_1
    assert rounds(jump_k=2, lazy_fold=True) == want           # ... never asked: at the start of the round after
    assert rounds(jump_k=2, lazy_fold=True, prefold_after=2) == want      # asked right after that round
    assert rounds(jump_k=2, lazy_fold=True, prefold_after=1) == want      # asked too early: nothing is due, no-op
    assert rounds(jump_k=3, lazy_fold=True, prefold_after=3) == want


def test_native_round_context_argument_checks(vm):
    ctx = vm.get_context()
    group = vm.EllipticCurve("Ed25519", "projective")
    h = group.generator
    k = vm.Ed25519Point.repeat(h, 99)
    g = vm.PointVector.fixed_base(h, list(range(2, 9)))           # 7 generators + h = 8
    g.precompute([h, k])
    z = vm.ScalarVector.from_ints(list(range(1, 9)))
    with pytest.raises(vm._native.VmpcError):                      # k's slot inside the tail of g_hat
        vm._native.P4Rounds(ctx, g._table, 1, 0, z.ptr, z.ptr)
    with pytest.raises(vm._native.VmpcError):                      # N = 7 + 0 is not a power of two
        vm._native.P4Rounds(ctx, g._table, 0, 1, z.ptr, z.ptr)
    rounds = vm._native.P4Rounds(ctx, g._table, 1, 1, z.ptr, z.ptr)
    rounds.round(None)
    rounds.round(5)
    with pytest.raises(vm._native.VmpcError):                      # log2(8) - 1 = 2 rounds only
        rounds.round(7)
    with pytest.raises(vm._native.VmpcError):                      # a later round needs the challenge
        rounds.round(None)
    z0, z1 = rounds.finish(7)                                      # z' = z_l + c z_r twice over
    zs = [(a + 5 * b) % ELL for a, b in zip(range(1, 5), range(5, 9))]
    assert (z0, z1) == ((zs[0] + 7 * zs[2]) % ELL, (zs[1] + 7 * zs[3]) % ELL)
    with pytest.raises(vm._native.VmpcError):
        rounds.finish(7)
    rounds.close()


def test_basic_pivot_fixture(vm, golden_small, monkeypatch, record_hashes):
    """Pi_s (pivot.py:156-205)."""
    case = golden_small["pis"][0]
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h = group.generator
    g = vm.PointVector.fixed_base(h, [h2i(e) for e in case["gen_exponents"]])
    x = [gf(h2i(v)) for v in case["x"]]
    L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
    gamma, y = h2i(case["gamma"]), gf(h2i(case["y"]))
    P = vm.pivot.vector_commitment(x, gamma, g, h)
    assert aff_hex(P) == case["P"]
    monkeypatch.setattr(vm.pivot, "prng", random.Random(case["seed"] + 2))
    monkeypatch.setattr(vm.Ed25519Point, "is_multiplicative", True)
    monkeypatch.setattr(vm.Ed25519Point, "is_additive", False)
    z, phi, c = vm.pivot.prove_linear_form_eval(g, h, P, L, y, x, gamma, gf)
    assert [hx(v.value) for v in z] == case["z"] and hx(phi) == case["phi"] and hx(c) == case["c"]
    assert vm.pivot.verify_linear_form_proof(g, h, P, L, y, z, phi, c) is True
    assert [hx(v) for v in record_hashes] == [hh["c"] for hh in case["hashes"]]
    assert vm.pivot.verify_linear_form_proof(g, h, P, L, y, z, phi + 1, c) is False


def test_vector_commitment_exact_representative(vm):
    """exact_representative=True replays pivot.py:143-144 + list_mul: same (X:Y:Z)."""
    rng = random.Random(77)
    n = 21
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    ogens = ac.create_generators(exps)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    g = vm.PointVector.fixed_base(group.generator, exps)
    x = [rng.randrange(ELL) for _ in range(n)]
    gamma = rng.randrange(ELL)
    want = ac.vector_commitment(x, gamma, ogens["g"], ogens["h"])       # signed exponents
    got = vm.pivot.vector_commitment([gf(v) for v in x], gamma, g, group.generator,
                                     exact_representative=True)
    assert got.coords == want
    want_u = ac.vector_commitment(x, gamma, ogens["g"], ogens["h"], signed_exponents=False)
    got_u = vm.pivot.vector_commitment(x, gamma, g, group.generator, exact_representative=True)
    assert got_u.coords == want_u
    assert vm.pivot.list_mul(g).coords == ed.tree_reduce(ed.pt_add, ogens["g"], ed.IDENTITY)


def test_reference_error_conventions(vm):
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    g = vm.PointVector.fixed_base(group.generator, [3, 5])
    with pytest.raises(AssertionError, match="Not enough generators."):
        vm.pivot.vector_commitment([1, 2, 3], 1, g, group.generator)
    gens = {"g": g, "h": group.generator, "k": group.generator}
    with pytest.raises(AssertionError, match="power of 2"):
        vm.compressed_pivot.protocol_5_prover(gens, group.generator, vm.pivot.LinearForm([1, 2]),
                                              0, [1, 2], 1, gf)
    with pytest.raises(NotImplementedError):
        vm.create_generators(3, vm.PivotChoice.koe, group)
    with pytest.raises(NotImplementedError):
        vm.EllipticCurve("BN256")


def test_mpc_local_commitment_shares(vm):
    """mpc_ac20.vector_commitment's local step (mpc_ac20.py:35-42): three parties, threshold 1;
    the product of the per-party elements equals the plain commitment of the secret vector."""
    from verifiable_mpc_amd import mpc_ac20
    rng = random.Random(31)
    n, parties, threshold = 40, 3, 1
    group = vm.EllipticCurve("Ed25519", "projective")
    g = vm.PointVector.fixed_base(group.generator, [rng.randrange(1, ELL) for _ in range(n)])
    h = group.generator
    x = [rng.randrange(ELL) for _ in range(n)]
    gamma = rng.randrange(ELL)
    shares = mpc_ac20.shamir_shares(x + [gamma], threshold, parties, rng)
    lam = mpc_ac20.recombination_vector([1, 2, 3])
    assert [sum(l * s[i] for l, s in zip(lam, shares)) % ELL for i in range(n + 1)] == x + [gamma]
    parts = [mpc_ac20.local_commitment_share(shares[p][:n], shares[p][n], g, h, lam[p])
             for p in range(parties)]
    want = vm.pivot.vector_commitment(x, gamma, g, h)
    assert mpc_ac20.combine_commitment_shares(parts) == want
    assert want.normalize().coords[:2] == ed.pt_affine(ac.vector_commitment(
        x, gamma, [p.coords for p in g.to_points()], h.coords))


def test_proof_survives_the_wire(vm, golden_small, monkeypatch):
    from verifiable_mpc_amd import wire
    case = golden_small["p5"][2]
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    gf = vm.GF(group.order)
    x = [gf(h2i(v)) for v in case["x"]]
    L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
    gamma, y = h2i(case["gamma"]), gf(h2i(case["y"]))
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf)
    back, mode = wire.deserialize_proof(wire.serialize_proof(proof), gf)
    assert mode == "reference"
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, back, gf) is True


@pytest.mark.parametrize("mode", ["reference", "compact"])
def test_verifier_rejects_points_outside_the_group(vm, golden_small, monkeypatch, mode):
    """Prover-supplied points that are off the curve, or on it but outside the order-l subgroup, make
    the verifier return False before any kernel does arithmetic on them (the niels mixed addition
    absorbs (0, 0); exponents are reduced mod l, compressed_pivot.py:66 does not reduce c**2)."""
    from verifiable_mpc_amd import compressed_pivot as cp
    case = golden_small["p5"][2]
    group, gens = build_generators(vm, case, monkeypatch, case["seed"] + 1)
    gf = vm.GF(group.order)
    x = [gf(h2i(v)) for v in case["x"]]
    L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
    gamma, y = h2i(case["gamma"]), gf(h2i(case["y"]))
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    proof = cp.protocol_5_prover(gens, P, L, y, x, gamma, gf, transcript=mode)
    assert cp.protocol_5_verifier(gens, P, L, y, proof, gf, transcript=mode) is True
    t4 = vm.Ed25519Point((pow(2, (ed.P - 1) // 4, ed.P), 0, 1), check=True)       # order 4
    t2 = vm.Ed25519Point((0, ed.P - 1, 1), check=True)                            # order 2
    off = vm.Ed25519Point((0, 0, 1))                                             # not on the curve
    assert cp._valid_group_elements([P, proof["A"], vm.Ed25519Point.identity]) is True
    for bad_pt in (off, t4, t2, vm.Ed25519Point.operation(proof["A0"], t4),
                   vm.Ed25519Point((5, 7, 0))):                                   # Z = 0
        assert cp._valid_group_elements([proof["A"], bad_pt]) is False
        for key in ("A0", "B1", "A"):
            bad = dict(proof)
            bad[key] = bad_pt
            assert cp.protocol_5_verifier(gens, P, L, y, bad, gf, transcript=mode) is False
        assert cp.protocol_5_verifier(gens, bad_pt, L, y, proof, gf, transcript=mode) is False
    # Protocol 4 called on its own validates Q and the round points too
    g_hat = gens["g"] + [gens["h"]]
    Lt = vm.pivot.LinearForm(list(L.coeffs) + [0])
    bad = {k_: v for k_, v in proof.items() if k_ not in ("t", "A")}
    bad["B0"] = t4
    assert cp.protocol_4_verifier(g_hat, gens["k"], P, Lt, gf, bad, transcript=mode) is False
