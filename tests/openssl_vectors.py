"""Shared reader of tests/golden/ed25519_openssl.json (made by tests/golden/make_openssl_vectors.py with
the OpenSSL 3 command-line tool - an Ed25519 implementation that is not this repository's).

Everything here is plain integer bookkeeping (RFC 8032 clamping, SHA-512, byte decoding); the curve
arithmetic under test is supplied by the caller."""
import hashlib

P = 2**255 - 19
ELL = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def secret_scalar(seed):
    a = int.from_bytes(hashlib.sha512(seed).digest()[:32], "little")
    a &= (1 << 254) - 8
    a |= 1 << 254
    return a


def decode_point(b):
    """RFC 8032 section 5.1.3, written out here so that the vectors are decoded without the oracle"""
    v = int.from_bytes(b, "little")
    sign, y = v >> 255, v & ((1 << 255) - 1)
    assert y < P
    y2 = y * y % P
    x2 = (y2 - 1) * pow(D * y2 + 1, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * SQRT_M1 % P
    assert (x * x - x2) % P == 0
    if (x & 1) != sign:
        x = P - x
    return x, y


def parsed(vectors):
    """-> list of dicts: a (secret scalar), A, R (affine pairs), S, h (ints): S*B == R + h*A"""
    out = []
    for v in vectors:
        seed, pub, msg, sig = (bytes.fromhex(v[k]) for k in ("seed", "pub", "msg", "sig"))
        S = int.from_bytes(sig[32:], "little")
        assert S < ELL
        h = int.from_bytes(hashlib.sha512(sig[:32] + pub + msg).digest(), "little") % ELL
        out.append({"a": secret_scalar(seed), "pub": pub, "A": decode_point(pub), "R": decode_point(sig[:32]),
                    "S": S, "h": h})
    return out


def affine_add(p1, p2):
    """The textbook affine twisted-Edwards addition law (a = -1) with two field inversions - a third,
    deliberately naive statement of the group law, independent of the oracle's and the shim's formulas."""
    x1, y1 = p1
    x2, y2 = p2
    t = D * x1 % P * x2 % P * y1 % P * y2 % P
    x3 = (x1 * y2 + y1 * x2) * pow(1 + t, P - 2, P) % P
    y3 = (y1 * y2 + x1 * x2) * pow(1 - t, P - 2, P) % P
    return x3, y3
