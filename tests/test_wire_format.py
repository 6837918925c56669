"""Proof wire format (CPU: host points only)."""
import random

import pytest

import verifiable_mpc_amd as vm
from verifiable_mpc_amd import wire
from oracle import ed25519_ref as ed


def test_point_codec_matches_rfc8032():
    G = vm.Ed25519Point.generator
    assert wire.compress_point(G).hex() == "58" + "66" * 31
    rng = random.Random(1)
    for _ in range(20):
        p = vm.Ed25519Point.repeat(G, rng.randrange(1, ed.ELL))
        enc = wire.compress_point(p)
        assert enc == ed.encode_rfc8032(p.coords)
        assert wire.decompress_point(enc) == p
    assert wire.decompress_point(wire.compress_point(vm.Ed25519Point.identity)) == vm.Ed25519Point.identity
    with pytest.raises(ValueError):
        wire.decompress_point((ed.P + 1).to_bytes(32, "little"))
    with pytest.raises(ValueError):
        wire.decompress_point((2).to_bytes(32, "little"))      # y = 2 is not on the curve


def test_proof_roundtrip():
    gf = vm.GF(ed.ELL)
    G = vm.Ed25519Point.generator
    rng = random.Random(2)
    rounds = 5
    proof = {"t": gf(rng.randrange(ed.ELL)), "A": vm.Ed25519Point.repeat(G, 5)}
    for i in range(rounds):
        proof[f"A{i}"] = vm.Ed25519Point.repeat(G, 100 + i)
        proof[f"B{i}"] = vm.Ed25519Point.operation(vm.Ed25519Point.repeat(G, 200 + i), G)   # Z != 1
    proof["z_prime"] = [gf(-3), gf(rng.randrange(ed.ELL))]
    for mode in ("reference", "compact"):
        blob = wire.serialize_proof(proof, mode)
        assert len(blob) == wire.proof_size(rounds)
        back, m = wire.deserialize_proof(blob, gf)
        assert m == mode and set(back) == set(proof)
        assert all(back[k] == proof[k] for k in proof)
    with pytest.raises(ValueError):
        wire.deserialize_proof(blob[:-1], gf)
    with pytest.raises(ValueError):
        wire.deserialize_proof(blob + b"\x00", gf)
    assert wire.proof_size(19) == 9 + 32 * 40 + 1 + 64      # N = 2^20: 1354 bytes
    # a point with a torsion component is a valid ENCODING but not a proof element: the order-4 point
    # (sqrt(-1), 0) added to A0 keeps it on the curve and takes it out of the order-l group
    t4 = vm.Ed25519Point((pow(2, (ed.P - 1) // 4, ed.P), 0, 1), check=True)
    assert not wire.in_prime_subgroup(t4) and wire.in_prime_subgroup(proof["A0"])
    bad = dict(proof)
    bad["A0"] = vm.Ed25519Point.operation(proof["A0"], t4)
    blob = wire.serialize_proof(bad, "compact")
    with pytest.raises(ValueError, match="subgroup"):
        wire.deserialize_proof(blob, gf)
    back, _ = wire.deserialize_proof(blob, gf, check_subgroup=False)     # the verifier's job then
    assert back["A0"] == bad["A0"]
