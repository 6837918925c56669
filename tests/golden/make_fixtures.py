#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference, read-only):
    python3 -B tests/golden/make_fixtures.py
The reference's verifiable_mpc.ac20.{pivot,compressed_pivot,circuit_sat_r1cs,
circuit_sat_cb,circuit_builder} are imported unmodified from /root/reference on top
of the build-written mpyc shim (tests/golden/mpyc_shim), every module-level `prng`
is replaced by a seeded random.Random, and the inputs/outputs of the hot-path calls
are recorded.  The fixtures are DATA (inputs, expected outputs, pre-image digests);
no reference source text is stored.

What these fixtures pin: the reference's protocol logic (hash layout, round structure,
index conventions, scalar algebra).  What they cannot pin: real MPyC's byte formats,
because the shim is ours (oracle/ed25519_ref.py header).
"""
import hashlib
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
# scripts/check_against_mpyc.py re-runs the cases below over REAL MPyC on a machine that has it
# (VMPC_FIXTURES_REAL_MPYC=1: no shim on the path; VMPC_REFERENCE: the reference checkout)
if os.environ.get("VMPC_FIXTURES_REAL_MPYC") != "1":
    sys.path.insert(0, os.path.join(HERE, "mpyc_shim"))
sys.path.insert(0, os.environ.get("VMPC_REFERENCE", "/root/reference"))
sys.path.insert(0, REPO)

# the KoE pivot's BN256 pairing module is out of scope and needs extension fields the
# shim does not have; a placeholder keeps `import knowledge_of_exponent` working.
_pairing_stub = types.ModuleType("verifiable_mpc.ac20.pairing")


def _no_pairing(*a, **k):
    raise NotImplementedError("pairings are outside the shim's scope")


_pairing_stub.optimal_ate = _no_pairing
sys.modules["verifiable_mpc.ac20.pairing"] = _pairing_stub

from mpyc.finfields import GF                                   # noqa: E402 (shim)
from mpyc.fingroups import EllipticCurve                        # noqa: E402 (shim)
import verifiable_mpc.ac20.pivot as pivot                       # noqa: E402 (reference)
import verifiable_mpc.ac20.compressed_pivot as compressed_pivot  # noqa: E402
import verifiable_mpc.ac20.circuit_sat_r1cs as cs_r1cs          # noqa: E402
from oracle import ed25519_ref as ed                            # noqa: E402

SEED = 20200152


def hx(v):
    return format(int(v), "x")


def pt_affine_hex(pt):
    n = pt.normalize()
    return [hx(n[0].value), hx(n[1].value)]


def pt_proj_hex(pt):
    return [hx(c.value) for c in pt.value]


def typed(v, order):
    if isinstance(v, int):
        return "i:" + str(v)
    return "f:" + hx(int(v) % order)


class Recorder:
    """Wraps pivot.fiat_shamir_hash to record each pre-image's digest and result."""

    def __init__(self, keep_text=False):
        self.calls = []
        self.keep_text = keep_text
        self._orig = pivot.fiat_shamir_hash

    def __enter__(self):
        def wrapped(input_list, order):
            text = str(input_list)
            c = self._orig(input_list, order)
            rec = {"len": len(text), "sha256": hashlib.sha256(text.encode()).hexdigest(),
                   "c": hx(c)}
            if self.keep_text:
                rec["text"] = text
            self.calls.append(rec)
            return c
        pivot.fiat_shamir_hash = wrapped
        return self

    def __exit__(self, *a):
        pivot.fiat_shamir_hash = self._orig


def group_and_field():
    group = EllipticCurve("Ed25519", "projective")   # demos/demo_zkp_ac20.py:46-49
    group.is_additive = False
    group.is_multiplicative = True
    return group, GF(modulus=group.order)


def p5_case(n, seed, keep_text=False, keep_proj=False):
    """Protocol 5 prove+verify on synthetic (L, x) as SURVEY.md section 8d cfg 3."""
    group, gf = group_and_field()
    rng = random.Random(seed)
    # generator exponents: reference draw order r_0..r_{n-1}, then k's (circuit_sat_r1cs.py:64,81)
    cs_r1cs.prng = random.Random(seed + 1)
    st = cs_r1cs.prng.getstate()
    generators = cs_r1cs.create_generators(n, cs_r1cs.PivotChoice.compressed, group)
    replay = random.Random()
    replay.setstate(st)
    exps = [replay.randrange(1, group.order) for _ in range(n)]
    exp_k = replay.randrange(1, group.order)

    x = [gf(rng.randrange(group.order)) for _ in range(n)]
    if n >= 7:                     # exercise zero / one / minus-one scalars
        x[1] = gf(0)
        x[2] = gf(1)
        x[3] = gf(-1)
    gamma = rng.randrange(1, group.order)
    L = pivot.LinearForm([gf(rng.randrange(group.order)) for _ in range(n)])
    P = pivot.vector_commitment(x, gamma, generators["g"], generators["h"])
    y = L(x)

    compressed_pivot.prng = random.Random(seed + 2)
    st = compressed_pivot.prng.getstate()
    with Recorder(keep_text) as rec:
        proof = compressed_pivot.protocol_5_prover(generators, P, L, y, x, gamma, gf)
        n_prover_hashes = len(rec.calls)
        ok = compressed_pivot.protocol_5_verifier(generators, P, L, y, proof, gf)
    replay.setstate(st)
    r = [replay.randrange(group.order) for _ in range(n)]
    rho = replay.randrange(group.order)
    assert ok is True
    rounds = (n + 1).bit_length() - 2
    prover_calls = rec.calls[:n_prover_hashes]
    assert rec.calls[n_prover_hashes:] == prover_calls  # verifier recomputes the same hashes
    out = {
        "n": n, "seed": seed, "rounds": rounds,
        "gen_exponents": [hx(e) for e in exps], "gen_exponent_k": hx(exp_k),
        "gens_proj_sha256": hashlib.sha256(b"".join(
            ed.proj_to_bytes(tuple(c.value for c in p.value))
            for p in generators["g"] + [generators["h"], generators["k"]])).hexdigest(),
        "x": [hx(v.value) for v in x], "gamma": hx(gamma),
        "L": [hx(v.value) for v in L.coeffs], "y": hx(y.value),
        "r": [hx(v) for v in r], "rho": hx(rho),
        "P": pt_affine_hex(P),
        "proof": {"t": hx(proof["t"].value), "A": pt_affine_hex(proof["A"]),
                  "A_i": [pt_affine_hex(proof[f"A{i}"]) for i in range(rounds)],
                  "B_i": [pt_affine_hex(proof[f"B{i}"]) for i in range(rounds)],
                  "z_prime": [hx(int(v) % group.order) for v in proof["z_prime"]]},
        "hashes": prover_calls,      # c0, c1, then one per round
        "verified": ok,
    }
    if keep_proj:
        out["gens_proj"] = [pt_proj_hex(p) for p in generators["g"]] + \
            [pt_proj_hex(generators["h"]), pt_proj_hex(generators["k"])]
    return out


def pis_case(n, seed):
    """Basic pivot Pi_s (pivot.py:156-205)."""
    group, gf = group_and_field()
    rng = random.Random(seed)
    exps = [rng.randrange(1, group.order) for _ in range(n)]
    g = [group.generator ** e for e in exps]
    h = group.generator
    x = [gf(rng.randrange(group.order)) for _ in range(n)]
    gamma = rng.randrange(1, group.order)
    L = pivot.LinearForm([gf(rng.randrange(group.order)) for _ in range(n)])
    P = pivot.vector_commitment(x, gamma, g, h)
    y = L(x)
    pivot.prng = random.Random(seed + 2)
    st = pivot.prng.getstate()
    with Recorder() as rec:
        z, phi, c = pivot.prove_linear_form_eval(g, h, P, L, y, x, gamma, gf)
        ok = pivot.verify_linear_form_proof(g, h, P, L, y, z, phi, c)
    replay = random.Random()
    replay.setstate(st)
    r = [replay.randrange(gf.order) for _ in range(n)]
    rho = replay.randrange(gf.order)
    assert ok is True
    return {"n": n, "seed": seed, "gen_exponents": [hx(e) for e in exps],
            "x": [hx(v.value) for v in x], "gamma": hx(gamma),
            "L": [hx(v.value) for v in L.coeffs], "y": hx(y.value),
            "r": [hx(v) for v in r], "rho": hx(rho), "P": pt_affine_hex(P),
            "z": [hx(v.value) for v in z], "phi": hx(phi), "c": hx(c),
            "hashes": rec.calls, "verified": ok}


def forms_case():
    """Known answers of the reference's own LinearForm test (ac20/test/test_pivot.py:84-90)."""
    lf = pivot.LinearForm([0, 1, 2])
    return {"expr_27": (lf + lf + 2 * lf + lf.eval([1, 1, 1]) - lf).eval([1, 2, 3]),
            "expr_8": lf([1, 2, 3])}


def demo_case(seed):
    """BASELINE config 1: demos/demo_zkp_ac20.py --elliptic (n=3), seeded, with the
    Protocol-5 call (compressed_pivot.py:89) intercepted."""
    import contextlib
    import io
    import verifiable_mpc.ac20.circuit_sat_cb as cs_cb
    import verifiable_mpc.ac20.circuit_builder as cb
    sys.path.insert(0, "/root/reference/demos")
    import demo_zkp_ac20 as demo
    demo.GROUP = "Elliptic"                    # what the --elliptic flag sets (:107-108)
    for i, mod in enumerate((cs_r1cs, cs_cb, compressed_pivot, pivot, cb)):
        mod.prng = random.Random(seed + 10 + i)
    gen_state = cs_r1cs.prng.getstate()
    captured = {}
    orig_p5 = compressed_pivot.protocol_5_prover

    def spy(generators, P, L, y, x, gamma, gf):
        st = compressed_pivot.prng.getstate()
        with Recorder() as rec:
            proof = orig_p5(generators, P, L, y, x, gamma, gf)
        replay = random.Random()
        replay.setstate(st)
        n = len(x)
        order = gf.order
        r = [replay.randrange(order) for _ in range(n)]
        rho = replay.randrange(order)
        rounds = (n + 1).bit_length() - 2
        replay.setstate(gen_state)
        exps = [replay.randrange(1, order) for _ in range(n)]
        exp_k = replay.randrange(1, order)
        captured.update({
            "n": n, "rounds": rounds,
            "gen_exponents": [hx(e) for e in exps], "gen_exponent_k": hx(exp_k),
            # Python typing of the scalar inputs matters to the transcript: plain ints stay
            # unreduced through pivot.py:62-70; "i:<decimal>" = int, "f:<hex residue>" = GF(l)
            "x_typed": [typed(v, order) for v in x], "gamma": hx(gamma),
            "L_typed": [typed(v, order) for v in L.coeffs],
            "L_constant_typed": typed(L.constant, order), "y_typed": typed(y, order),
            "x": [hx(int(v) % order) for v in x],
            "L": [hx(int(v) % order) for v in L.coeffs],
            "L_constant": hx(int(L.constant) % order), "y": hx(int(y) % order),
            "r": [hx(v) for v in r], "rho": hx(rho), "P": pt_affine_hex(P),
            "proof": {"t": hx(int(proof["t"]) % order), "A": pt_affine_hex(proof["A"]),
                      "A_i": [pt_affine_hex(proof[f"A{i}"]) for i in range(rounds)],
                      "B_i": [pt_affine_hex(proof[f"B{i}"]) for i in range(rounds)],
                      "z_prime": [hx(int(v) % order) for v in proof["z_prime"]]},
            "hashes": rec.calls,
        })
        return proof

    # Protocol 8 level (circuit_sat_cb.py:59-166, :255-318): everything the two Protocol-8 hashes are made
    # of, the un-normalised commitment [z] they contain (:103-111), and the proof as returned
    p8 = {}
    orig_p8 = cs_cb.protocol_8_excl_pivot_prover
    orig_forms = cb.calculate_circuit_forms

    def form_rec(f, order):
        return {"coeffs": [typed(v, order) for v in f.coeffs], "constant": typed(f.constant, order),
                "linear": isinstance(f, pivot.LinearForm)}

    def spy_p8(generators, circuit, x, gf, use_koe=False):
        order = gf.order
        with Recorder() as rec:
            proof, z_commitment, L, z, gamma = orig_p8(generators, circuit, x, gf, use_koe)
        # the objects of the second hash are rebuilt exactly as :127-147 builds them (deterministic
        # functions of the circuit, the first challenge and y1..y3)
        c1 = int(rec.calls[0]["c"], 16)
        lf = cb.calculate_fg_form(circuit, wire=0, challenge=c1, gf=gf)
        lg = cb.calculate_fg_form(circuit, wire=1, challenge=c1, gf=gf)
        lh = cb.calculate_h_form(circuit, c1, gf)
        circuit_forms = [cb.convert_to_ac20(f, circuit) for f in orig_forms(circuit)]
        outputs = proof["outputs"]
        lin_forms = [form - y for form, y in zip(circuit_forms, outputs)] + \
            [lf - proof["y1"], lg - proof["y2"], lh - proof["y3"]]
        check = [proof["y1"], proof["y2"], proof["y3"], z_commitment, outputs, circuit_forms, lin_forms,
                 "Second hash circuit satisfiability protocol"]
        assert hashlib.sha256(str(check).encode()).hexdigest() == rec.calls[1]["sha256"]
        p8.update({
            "z_typed": [typed(v, order) for v in z], "gamma": hx(gamma),
            "z_commitment_proj": pt_proj_hex(z_commitment),
            "circuit_str": str(circuit),
            "hashes": rec.calls,                                  # first, second hash (:107-111, :149-162)
            "y_typed": [typed(proof[k], order) for k in ("y1", "y2", "y3")],
            "outputs_typed": [typed(v, order) for v in outputs],
            "circuit_forms": [form_rec(f, order) for f in circuit_forms],
            "lin_forms": [form_rec(f, order) for f in lin_forms],
            "L": form_rec(L, order),
        })
        return proof, z_commitment, L, z, gamma

    all_hashes = Recorder()
    captured_proof = {}
    orig_prover = cs_cb.circuit_sat_prover

    def spy_prover(generators, circuit, x, gf, pivot_choice=cs_cb.PivotChoice.compressed):
        proof = orig_prover(generators, circuit, x, gf, pivot_choice)
        captured_proof["proof"] = proof
        return proof

    compressed_pivot.protocol_5_prover = spy
    cs_cb.protocol_8_excl_pivot_prover = spy_p8
    cs_cb.circuit_sat_prover = spy_prover
    try:
        with contextlib.redirect_stdout(io.StringIO()) as buf, all_hashes:
            verification = demo.main(cs_cb.PivotChoice.compressed, 3)
    finally:
        compressed_pivot.protocol_5_prover = orig_p5
        cs_cb.protocol_8_excl_pivot_prover = orig_p8
        cs_cb.circuit_sat_prover = orig_prover
    captured["verification"] = verification
    captured["stdout_head"] = [l for l in buf.getvalue().replace("\r", "\n").split("\n")
                               if l.startswith("Length of")]
    # the full returned proof (demos/demo_zkp_ac20.py:82-84 prints it): points with their
    # representatives, scalars with their Python type
    order = 2**252 + 27742317777372353535851937790883648493
    pp = captured_proof["proof"]["pivot_proof"]
    rounds = captured["rounds"]
    captured["protocol8"] = p8
    captured["returned_proof"] = {
        "keys": list(captured_proof["proof"].keys()),
        "pivot_proof_keys": list(pp.keys()),
        "t_typed": typed(pp["t"], order), "A_proj": pt_proj_hex(pp["A"]),
        "A_i_proj": [pt_proj_hex(pp[f"A{i}"]) for i in range(rounds)],
        "B_i_proj": [pt_proj_hex(pp[f"B{i}"]) for i in range(rounds)],
        "z_prime_typed": [typed(v, order) for v in pp["z_prime"]],
    }
    # every Fiat-Shamir hash of the run in call order: prover (2 Protocol-8, c0, c1, one per round), then
    # the verifier's recomputation of the same
    captured["all_hashes"] = all_hashes.calls
    half_n = len(all_hashes.calls) // 2
    assert all_hashes.calls[:half_n] == all_hashes.calls[half_n:]
    return captured


def mpc_case(n, seed):
    """the reference's MPyC driver (verifiable_mpc/ac20/mpc_ac20.py:35-51,141-269) run
    with ONE party (m = 1: shares are the values; how test/test_demo_zkp_mpc_ac20.py:17-23 runs it) on
    the shim's single-party `mpc`.  Pins what is opened and hashed when, the draw order of the jointly
    random exponents / masks, and the proof; group elements affine (the representative real MPyC's
    secure_repeat would return is unknown)."""
    import verifiable_mpc.ac20.mpc_ac20 as mpc_ac20
    from mpyc.runtime import mpc
    group, _ = group_and_field()
    secfld = mpc.SecFld(modulus=group.order)
    gf = secfld.field
    rng = random.Random(seed)
    mpc._rng = random.Random(seed + 1)
    st = mpc._rng.getstate()
    generators = mpc.run(mpc_ac20.create_generators(group, secfld, n))
    # secure_repeat OPENS a group element; which representative real MPyC hands back is unknown, and the
    # generators enter the first pre-image as they are (mpc_ac20.py:238).  The harness therefore passes
    # them on normalised - on both sides - so that the transcript is a function of group elements only.
    generators = {"g": [p.normalize() for p in generators["g"]], "h": generators["h"],
                  "k": generators["k"].normalize()}
    replay = random.Random()
    replay.setstate(st)
    exps = [replay.randrange(group.order) for _ in range(n + 1)]          # [0] is k's (mpc_ac20.py:50)
    x_plain = [gf(rng.randrange(group.order)) for _ in range(n)]
    x_plain[1], x_plain[2] = gf(0), gf(-1)
    gamma_plain = gf(rng.randrange(1, group.order))
    x = [secfld(v) for v in x_plain]
    gamma = secfld(gamma_plain)
    L = pivot.LinearForm([gf(rng.randrange(group.order)) for _ in range(n)])
    P = mpc.run(mpc_ac20.vector_commitment(x, gamma, generators["g"], generators["h"]))
    y = L(x)                                                               # secret
    mpc._rng = random.Random(seed + 2)
    st = mpc._rng.getstate()
    with Recorder() as rec:
        proof = mpc.run(mpc_ac20.protocol_5_prover(generators, P, L, y, x, gamma, gf))
        n_prover = len(rec.calls)
        ok = compressed_pivot.protocol_5_verifier(generators, P, L, y.share, proof, gf)
    assert ok is True and rec.calls[n_prover:] == rec.calls[:n_prover]
    replay.setstate(st)
    r = [replay.randrange(group.order) for _ in range(n)]
    rho = replay.randrange(group.order)
    rounds = (n + 1).bit_length() - 2
    return {"n": n, "seed": seed, "rounds": rounds, "parties": 1,
            "gen_exponents": [hx(e) for e in exps],
            "generators": {"g": [pt_affine_hex(p) for p in generators["g"]],
                           "h": pt_affine_hex(generators["h"]), "k": pt_affine_hex(generators["k"])},
            "x": [hx(v.value) for v in x_plain], "gamma": hx(gamma_plain.value),
            "L": [hx(v.value) for v in L.coeffs], "y": hx(y.share.value),
            "r": [hx(v) for v in r], "rho": hx(rho), "P": pt_affine_hex(P),
            "proof": {"t": hx(proof["t"].value), "A": pt_affine_hex(proof["A"]),
                      "A_i": [pt_affine_hex(proof[f"A{i}"]) for i in range(rounds)],
                      "B_i": [pt_affine_hex(proof[f"B{i}"]) for i in range(rounds)],
                      "z_prime": [hx(int(v) % group.order) for v in proof["z_prime"]]},
            "proof_keys": list(proof.keys()),
            "hashes": rec.calls[:n_prover], "verified": ok}


def pynocchio_case(seed):
    """BASELINE config 5 shape: the reference's Pinocchio key generation and compute_proof
    (trinocchio/pynocchio.py:101-273) on the demo's program (demos/demo_zkp_pynocchio.py:45-50),
    BN-256 groups from the shim.  Stores the evaluation-key points the prover reads, the witness,
    h and the eight proof elements, all affine."""
    import verifiable_mpc.trinocchio.pynocchio as pynocchio
    import verifiable_mpc.tools.code_to_qap as c2q
    import verifiable_mpc.tools.qap_creator as qc
    bn_curve = EllipticCurve("BN256", "jacobian")
    bn_twist = EllipticCurve("BN256_twist", "jacobian")
    g1, g2 = bn_curve.generator, bn_twist.generator
    modulus = bn_curve.order
    gf = GF(modulus=modulus)
    gf.is_signed = False
    pynocchio.prng = random.Random(seed)
    inputs = [gf(3)]
    code = """
def qeval(x):
    y = x**3 + x**2 + x
    return y + x + 5
"""
    qap = c2q.QAP(code, gf)
    td = pynocchio.Trapdoor(modulus)
    gen = pynocchio.Generators(td, g1, g2)
    evalkey = pynocchio.generate_evalkey(td, qap, gen)
    c = qap.calculate_witness(inputs)
    p_poly = pynocchio.compute_p_poly(qap, c)
    h, r = p_poly / qap.t
    assert r == qc.Poly([0] * qap.d)
    deltas = pynocchio.SampleDeltas(modulus)
    h = h + pynocchio.compute_h_zk_terms(qap, c, deltas)
    proof = pynocchio.compute_proof(qap, c, h, evalkey, deltas)

    def enc(pt):
        v = pt.value
        if v is None:
            return None
        flat = []
        for cpt in v:
            flat += list(cpt) if isinstance(cpt, tuple) else [cpt]
        return [hx(x) for x in flat]
    try:
        gf.is_signed = True
    except Exception:
        pass
    return {"seed": seed, "m": qap.m, "d": qap.d, "indices_mid": list(qap.indices_mid),
            "c": [hx(int(v) % modulus) for v in c], "h": [hx(int(v) % modulus) for v in h.coeffs],
            "deltas": [hx(deltas.v), hx(deltas.w), hx(deltas.y)],
            "evalkey": {k: enc(v) for k, v in evalkey.items()},
            "proof": {k: enc(v) for k, v in proof.items()}}


def main():
    out = {
        "forms": forms_case(),
        "p5": [p5_case(3, SEED, keep_text=True, keep_proj=True),
               p5_case(7, SEED + 100, keep_proj=True),
               p5_case(15, SEED + 200),
               p5_case(127, SEED + 300)],
        "pis": [pis_case(4, SEED + 400)],
    }
    with open(os.path.join(HERE, "ac20_ed25519_small.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, "ac20_ed25519_n1023.json"), "w") as f:
        json.dump(p5_case(1023, SEED + 500), f, indent=0, sort_keys=True)
    with open(os.path.join(HERE, "mpc_ac20_m1.json"), "w") as f:
        json.dump({"cases": [mpc_case(7, SEED + 800), mpc_case(31, SEED + 900)]}, f, indent=0, sort_keys=True)
    try:
        with open(os.path.join(HERE, "pynocchio_bn256.json"), "w") as f:
            json.dump(pynocchio_case(SEED + 700), f, indent=0, sort_keys=True)
        print("pynocchio fixture written")
    except Exception as e:
        import traceback
        traceback.print_exc()
        print("pynocchio fixture not generated:", type(e).__name__, e)
    try:
        demo = demo_case(SEED + 600)
        with open(os.path.join(HERE, "demo_zkp_ac20_elliptic.json"), "w") as f:
            json.dump(demo, f, indent=0, sort_keys=True)
        print("demo:", demo["n"], demo["verification"], demo["stdout_head"])
    except Exception as e:      # the demo needs more of mpyc than the P5 path
        print("demo fixture not generated:", type(e).__name__, e)
    print("fixtures written")


if __name__ == "__main__":
    main()
