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
