"""Canonical byte format of a Protocol-5 proof.

The reference never serialises a proof: it is a dict of live objects
(verifiable_mpc/ac20/compressed_pivot.py:94,112-113,44-45,78).  This module defines a wire
format so that proofs produced on the GPU can leave the process:

    magic  b"AC20P5"  | version u8 = 1 | transcript u8 (0 reference, 1 compact) | rounds u8
    t        32 B  little-endian residue mod l
    A        32 B  RFC 8032 point encoding (y, sign of x in the top bit)
    A_0 .. A_{R-1}, then B_0 .. B_{R-1}     32 B each
    len(z')  u8, then z'  32 B each
Points are group elements (representatives are not preserved: every point of a proof is
normalised before it is hashed, compressed_pivot.py:52,118).
"""
from .groups import ORDER, P, D, Ed25519Point

MAGIC = b"AC20P5"
_SQRT_M1 = pow(2, (P - 1) // 4, P)


def compress_point(pt):
    x, y, _ = pt.normalize().coords
    return (y | ((x & 1) << 255)).to_bytes(32, "little")


def decompress_point(b):
    v = int.from_bytes(b, "little")
    sign, y = v >> 255, v & ((1 << 255) - 1)
    if y >= P:
        raise ValueError("non-canonical point encoding")
    y2 = y * y % P
    u, w = (y2 - 1) % P, (D * y2 + 1) % P
    x2 = u * pow(w, P - 2, P) % P
    x = pow(x2, (P + 3) // 8, P)
    if (x * x - x2) % P:
        x = x * _SQRT_M1 % P
    if (x * x - x2) % P:
        raise ValueError("not a curve point")
    if x == 0 and sign:
        raise ValueError("non-canonical point encoding")
    if (x & 1) != sign:
        x = P - x
    return Ed25519Point((x, y, 1))


def in_prime_subgroup(pt):
    """l * pt == identity (host big-int ladder, ~1.5 ms per point).  decompress_point accepts every
    curve point, including the 8-torsion and its translates; a proof's points must lie in the
    order-l group for the verifier's reduced exponents to mean what the reference's unreduced ones
    do (compressed_pivot.py:66)."""
    return Ed25519Point.repeat(pt, ORDER) == Ed25519Point.identity


def _scalar(v):
    return (int(v) % ORDER).to_bytes(32, "little")


def serialize_proof(proof, transcript="reference"):
    rounds = sum(1 for k in proof if k.startswith("A") and k[1:].isdigit())
    out = [MAGIC, bytes([1, 0 if transcript == "reference" else 1, rounds]),
           _scalar(proof["t"]), compress_point(proof["A"])]
    out += [compress_point(proof[f"A{i}"]) for i in range(rounds)]
    out += [compress_point(proof[f"B{i}"]) for i in range(rounds)]
    z = proof["z_prime"]
    out.append(bytes([len(z)]))
    out += [_scalar(v) for v in z]
    return b"".join(out)


def deserialize_proof(data, gf, check_subgroup=True):
    """-> (proof dict with `gf` scalars and Ed25519Point points, transcript name).  Raises ValueError
    for anything that is not a canonical encoding of scalars < l and points of the order-l group
    (`check_subgroup=False` leaves the group-membership test to protocol_5_verifier, which runs it
    on the device for all points at once)."""
    if data[:6] != MAGIC or data[6] != 1:
        raise ValueError("not an AC20 Protocol-5 proof (v1)")
    transcript = "reference" if data[7] == 0 else "compact"
    rounds = data[8]
    off = 9

    def take(n):
        nonlocal off
        chunk = data[off:off + n]
        if len(chunk) != n:
            raise ValueError("truncated proof")
        off += n
        return chunk

    def scalar():
        v = int.from_bytes(take(32), "little")
        if v >= ORDER:
            raise ValueError("non-canonical scalar")
        return gf(v)

    def point():
        pt = decompress_point(take(32))
        if check_subgroup and not in_prime_subgroup(pt):
            raise ValueError("point outside the prime-order subgroup")
        return pt

    proof = {"t": scalar(), "A": point()}
    for i in range(rounds):
        proof[f"A{i}"] = point()
    for i in range(rounds):
        proof[f"B{i}"] = point()
    nz = take(1)[0]
    proof["z_prime"] = [scalar() for _ in range(nz)]
    if off != len(data):
        raise ValueError("trailing bytes")
    return proof, transcript


def proof_size(rounds, nz=2):
    return 9 + 32 * (2 + 2 * rounds) + 1 + 32 * nz
