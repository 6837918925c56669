"""Pins the C oracle (oracle/ed25519_oracle.c) to the Python oracle: exact projective
representatives for vector_commitment / fold / fixed base, plus RFC 8032 via the Python side."""
import random
import time

import numpy as np

from oracle import ac20_ref as ac
from oracle import c_oracle
from oracle import ed25519_ref as ed

ELL = ed.ELL


def sc(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), np.uint8).reshape(-1, 32)


def proj(pts):
    return np.frombuffer(b"".join(ed.proj_to_bytes(p) for p in pts), np.uint8).reshape(-1, 96)


def aff(pts):
    return np.frombuffer(b"".join(ed.affine_to_bytes(p) for p in pts), np.uint8).reshape(-1, 64)


def test_fixed_base_and_fold_exact():
    rng = random.Random(1)
    exps = [rng.randrange(1, ELL) for _ in range(12)] + [1, 2, ELL - 1]
    op, oa = c_oracle.fixed_base(proj([ed.BASE])[0], sc(exps))
    want = ac.create_generators(exps)["g"]
    assert [ed.proj_from_bytes(op[i].tobytes()) for i in range(len(exps))] == want
    assert [ed.affine_from_bytes(oa[i].tobytes())[:2] for i in range(len(exps))] == [ed.pt_affine(p) for p in want]
    gl, gr = want[:6], want[6:12]
    for c in (rng.randrange(ELL), 0, 1):
        fp, fa = c_oracle.fold(proj(gl), proj(gr), sc([c])[0], proj_in=True)
        assert [ed.proj_from_bytes(fp[i].tobytes()) for i in range(6)] == ac.fold_generators(gl, gr, c)


def test_vector_commitment_exact():
    rng = random.Random(2)
    for n in (1, 2, 5, 16, 33):
        gens = ac.create_generators([rng.randrange(1, ELL) for _ in range(n)])
        x = [rng.randrange(ELL) for _ in range(n)]
        x[0] = ELL - 1
        gamma = rng.randrange(ELL)
        for signed in (False, True):
            op, oa = c_oracle.vector_commitment(sc(x), sc([gamma])[0], proj(gens["g"]), proj([gens["h"]])[0],
                                                proj_in=True, signed_exp=signed)
            want = ac.vector_commitment(x, gamma, gens["g"], gens["h"], signed_exponents=signed)
            assert ed.proj_from_bytes(op.tobytes()) == want
            assert ed.affine_from_bytes(oa.tobytes())[:2] == ed.pt_affine(want)
        # affine inputs
        op, oa = c_oracle.vector_commitment(sc(x), sc([gamma])[0], aff(gens["g"]), aff([gens["h"]])[0])
        assert ed.affine_from_bytes(oa.tobytes())[:2] == ed.pt_affine(want)


def test_rfc8032_through_c():
    # TEST 1 secret scalar a (clamped SHA-512 half) times B gives the published public key
    import hashlib
    seed = bytes.fromhex("9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60")
    a = int.from_bytes(hashlib.sha512(seed).digest()[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    _, oa = c_oracle.fixed_base(proj([ed.BASE])[0], sc([a % ELL]))
    pt = ed.affine_from_bytes(oa[0].tobytes())
    assert ed.encode_rfc8032(pt).hex() == "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a"


def test_threads_change_nothing():
    """the threaded run (ladders, fold elements, tree levels spread over cores) gives the single-threaded run's
    exact (X:Y:Z) - every tree shape from 1 to a few thousand leaves, odd levels included"""
    rng = random.Random(3)
    base = proj([ed.BASE])[0]
    n = 3000
    exps = sc([rng.randrange(1, ELL) for _ in range(n)])
    x = sc([rng.randrange(ELL) for _ in range(n)])
    gamma = sc([rng.randrange(ELL)])[0]
    prev = c_oracle.set_threads(1)
    try:
        g1, a1 = c_oracle.fixed_base(base, exps)
        f1 = c_oracle.fold(g1[:n // 2], g1[n // 2:], gamma, proj_in=True)[0]
        want = {m: c_oracle.vector_commitment(x[:m], gamma, g1[:m], base, proj_in=True, signed_exp=True)[0].tobytes()
                for m in (1, 2, 3, 1023, 1024, 1025, 2047, 2999, 3000)}
        c_oracle.set_threads(5)
        g5, a5 = c_oracle.fixed_base(base, exps)
        assert (g1 == g5).all() and (a1 == a5).all()
        assert (c_oracle.fold(g1[:n // 2], g1[n // 2:], gamma, proj_in=True)[0] == f1).all()
        for m, w in want.items():
            assert c_oracle.vector_commitment(x[:m], gamma, g1[:m], base, proj_in=True,
                                              signed_exp=True)[0].tobytes() == w, m
    finally:
        c_oracle.set_threads(prev)


def test_ac20_ref_over_point_arrays():
    """oracle/ac20_ref.py run with the generator vector held as a c_oracle.PointArray (group work in C, what makes
    N = 2^20 reachable) returns what it returns over lists of int tuples: both transcripts, every challenge,
    prover and verifier"""
    rng = random.Random(4)
    n = 31
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    ek = rng.randrange(1, ELL)
    gens = ac.create_generators(exps, ek)
    agens = dict(gens, g=c_oracle.PointArray.from_points(gens["g"]))
    assert list(agens["g"]) == gens["g"] and agens["g"][5] == gens["g"][5] and len(agens["g"][3:9]) == 6
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    P = ac.vector_commitment(x, gamma, gens["g"], gens["h"])
    assert ac.vector_commitment(x, gamma, agens["g"], gens["h"]) == P
    assert ac.vector_commitment(x, -5, agens["g"], gens["h"]) == ac.vector_commitment(x, -5, gens["g"], gens["h"])
    y = ac.form_eval(coeffs, 0, x)
    prev = c_oracle.set_threads(3)
    try:
        for mode in ("reference", "compact"):
            t1, t2 = {}, {}
            want = ac.protocol_5_prover(gens, P, coeffs, 0, y, x, gamma, r, rho, mode, trace=t1)
            got = ac.protocol_5_prover(agens, P, coeffs, 0, y, x, gamma, r, rho, mode, trace=t2)
            assert got == want
            assert t1["c"] == t2["c"] and (t1["c0"], t1["c1"]) == (t2["c0"], t2["c1"])
            assert [list(v) for v in t2["g_hat"]] == t1["g_hat"]
            assert ac.protocol_5_verifier(agens, P, coeffs, 0, y, got, mode) is True
            assert ac.protocol_5_verifier(agens, P, coeffs, 0, (y + 1) % ELL, got, mode) is False
    finally:
        c_oracle.set_threads(prev)
