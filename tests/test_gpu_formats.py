"""The [mpyc-recall] format layer behind ONE runtime switch (verifiable_mpc_amd.set_reference_format): flipping the
bracket pair of a point, the signedness of a coordinate or of a scalar changes the PRODUCT's pre-image text - host
elements and device-formatted vectors alike - exactly as the oracle's own switches change its restatement
(verifiable_mpc/ac20/pivot.py:131-136, compressed_pivot.py:51-59,117-130), with no rebuild."""
import random

import pytest

from oracle import ac20_ref as ac
from oracle import ed25519_ref as ed

pytestmark = pytest.mark.gpu
ELL = ed.ELL
VARIANTS = [("[]", False, True),        # the defaults (what the fixtures were made with)
            ("[]", True, True),         # signed coordinates - the riskiest recall
            ("()", False, True),
            ("[]", False, False),       # unsigned scalars
            ("()", True, False)]


@pytest.fixture()
def vm():
    import verifiable_mpc_amd as vm_
    return vm_


@pytest.fixture(params=VARIANTS, ids=lambda v: f"{v[0]}-coord{'S' if v[1] else 'U'}-scalar{'S' if v[2] else 'U'}")
def fmt(request, vm):
    b, c, s = request.param
    prev = vm.set_reference_format(point_brackets=b, coord_signed=c, scalar_signed=s)
    ed.set_format(b, c, s)
    try:
        yield request.param
    finally:
        vm.set_reference_format(**prev)
        ed.set_format()


def proj_bytes(pts):
    import numpy as np
    return np.frombuffer(b"".join(c.to_bytes(32, "little") for p in pts for c in p), np.uint8)


def test_device_text_follows_the_switch(vm, fmt):
    ctx = vm.get_context()
    rng = random.Random(16)
    pts = ac.create_generators([rng.randrange(1, ELL) for _ in range(200)])["g"]
    half = (ed.P - 1) // 2
    pts += [ed.IDENTITY, ed.BASE, (0, 5, 7), (10**76, 1, 10**9), (half, half + 1, ed.P - 1), (2**254 - 9, 2**254 - 10, 1)]
    txt = ctx.format_points(ctx.upload(proj_bytes(pts)).ptr, len(pts)).tobytes().decode()
    assert txt == "".join(ed.pt_repr(p) + ", " for p in pts)
    if fmt[1]:
        assert "-" in txt                        # half of all coordinates print negative
    assert txt[0] == fmt[0][0]
    sc = [rng.randrange(ELL) for _ in range(500)] + [0, 1, ELL - 1, ELL // 2, ELL // 2 + 1]
    v = vm.ScalarVector.from_ints(sc)
    assert v.text().tobytes().decode() == "".join(ed.scalar_repr(x) + ", " for x in sc)       # default: the switch
    # host elements print the same way as the device vectors
    gf = vm.GF(ELL)
    assert [repr(gf(x)) for x in sc[-5:]] == [ed.scalar_repr(x) for x in sc[-5:]]
    assert repr(vm.Ed25519Point(pts[3])) == ed.pt_repr(pts[3])
    assert repr(vm.Ed25519Point((half + 1, 2, 3))) == ed.pt_repr((half + 1, 2, 3))


@pytest.mark.parametrize("n", [3, 15])
@pytest.mark.parametrize("device_mode", [False, True])
def test_protocol5_under_every_format_equals_the_oracle(vm, fmt, monkeypatch, n, device_mode):
    """the whole Protocol-5 proof - every challenge included - product against oracle under the same switches"""
    rng = random.Random(20200152 + n)
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    ek = rng.randrange(1, ELL)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    assert gf.is_signed == fmt[2]
    gens = {"g": vm.PointVector.fixed_base(group.generator, exps), "h": group.generator,
            "k": vm.Ed25519Point.repeat(group.generator, ek)}
    ogens = ac.create_generators(exps, ek)
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)
    r = [rng.randrange(ELL) for _ in range(n)]
    oP = ac.vector_commitment(x, gamma, ogens["g"], ogens["h"])
    y = ac.form_eval(coeffs, 0, x)
    trace = {}
    want = ac.protocol_5_prover(ogens, oP, coeffs, 0, y, x, gamma, r, rho, "reference", trace=trace)

    calls = []
    orig, orig_v = vm.pivot.fiat_shamir_hash, vm.pivot.fiat_shamir_hash_variants
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash", lambda il, order: calls.append(orig(il, order)) or calls[-1])
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash_variants",
                        lambda common, tails, order: (lambda cs: calls.extend(cs) or cs)(orig_v(common, tails, order)))
    if device_mode:
        xs = vm.ScalarVector.from_ints(x)
        L = vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
    else:
        xs = [gf(v) for v in x]
        L = vm.pivot.LinearForm([gf(c) for c in coeffs])
    P = vm.pivot.vector_commitment(xs, gamma, gens["g"], gens["h"])
    assert tuple(P.normalize().coords[:2]) == ed.pt_affine(oP)
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, gf(y), xs, gamma, gf, r=list(r), rho=rho)
    assert calls == [trace["c0"], trace["c1"]] + trace["c"]
    for key, val in want.items():
        if key == "t":
            assert int(proof[key]) % ELL == val
        elif key == "z_prime":
            assert [int(v) % ELL for v in proof[key]] == val
        else:
            assert tuple(proof[key].normalize().coords[:2]) == ed.pt_affine(val), key
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, gf(y), proof, gf) is True


def test_the_switch_changes_the_challenges(vm):
    """sanity: the setting is not ignored - the first challenge of the same proof differs between formats"""
    rng = random.Random(5)
    n = 15          # (17 scalars enter the first pre-image: some print differently signed / unsigned for sure)
    exps, ek = [rng.randrange(1, ELL) for _ in range(n)], rng.randrange(1, ELL)
    group = vm.EllipticCurve("Ed25519", "projective")
    gens = {"g": vm.PointVector.fixed_base(group.generator, exps), "h": group.generator,
            "k": vm.Ed25519Point.repeat(group.generator, ek)}
    x, coeffs = [rng.randrange(ELL) for _ in range(n)], [rng.randrange(ELL) for _ in range(n)]
    r, rho, gamma = [rng.randrange(ELL) for _ in range(n)], 7, 11
    seen = {}
    for b, c, s in VARIANTS:
        prev = vm.set_reference_format(point_brackets=b, coord_signed=c, scalar_signed=s)
        try:
            gf = vm.GF(group.order)
            xs = vm.ScalarVector.from_ints(x)
            L = vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
            P = vm.pivot.vector_commitment(xs, gamma, gens["g"], gens["h"])
            proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, gf(L(xs)), xs, gamma, gf, r=list(r), rho=rho)
            seen[(b, c, s)] = tuple(proof["A0"].normalize().coords[:2]) + tuple(int(v) for v in proof["z_prime"])
        finally:
            vm.set_reference_format(**prev)
    assert len(set(seen.values())) == len(VARIANTS), {k: hex(v[0])[:12] for k, v in seen.items()}
    assert vm.get_reference_format() == {"point_brackets": "[]", "coord_signed": False, "scalar_signed": True}


def test_text_delivered_in_pieces_equals_the_whole(vm, monkeypatch):
    """vmpc_format_*_chunked_dev: the transcript text of a long vector reaches the host in pieces that are hashed as
    they land (pivot._feed over text_chunks); here with 4-KiB pieces over a few thousand elements: the pieces joined are
    the text formatted in one go minus its trailing separator, for points and for scalars, and a folded vector whose
    text was produced slice by slice (PointVector.fold(stream_text=True)) hashes like the plain fold's."""
    rng = random.Random(4242)
    n = 3000
    ctx = vm.get_context()
    g = vm.PointVector.fixed_base(vm.Ed25519Point.generator, [rng.randrange(1, ELL) for _ in range(n)], keep_proj=True)
    sv = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)])
    whole_g, whole_s = bytes(g.text()), bytes(sv.text())
    side_cls = type(ctx)
    monkeypatch.setattr(side_cls, "TEXT_CHUNK_BYTES", 4096)
    g.text_begin()
    sv.text_begin()
    assert g._pending_text[1].chunk_bytes == 4096
    pieces = [bytes(p) for p in g.text_chunks()]
    assert len(pieces) > 100 and b"".join(pieces) == whole_g[:-2]
    assert b"".join(bytes(p) for p in sv.text_chunks()) == whole_s[:-2]
    # a second pass over an already delivered text, and the synchronous path, give the same bytes
    assert b"".join(bytes(p) for p in g.text_chunks()) == whole_g[:-2]
    assert bytes(g.text()) == whole_g
    # folds: slice by slice, growing (128 + 256 + 512 + 512 elements here), against one launch
    monkeypatch.setattr(vm.PointVector, "TEXT_SLICE", 512)
    monkeypatch.setattr(vm.PointVector, "TEXT_FIRST_SLICE", 128)
    monkeypatch.setattr(vm.PointVector, "TEXT_SLICED_FROM", 1024)
    c = rng.randrange(ELL)
    half = 1408
    plain = g[:half].fold(g[half:2 * half], c)
    sliced = g[:half].fold(g[half:2 * half], c, stream_text=True)
    assert len(sliced._pending_text[1].parts) == 4
    assert bytes(plain.text())[:-2] == b"".join(bytes(p) for p in sliced.text_chunks())
    assert plain.affine_array().tobytes() == sliced.affine_array().tobytes()
    # g + [h] right after g: the parent's text is reused, the extra point formatted on its own
    h = vm.Ed25519Point.repeat(vm.Ed25519Point.generator, 77)
    gh = g + [h]
    joined = b"".join(bytes(p) for p in gh.text_chunks())
    fresh = vm.PointVector.from_points(g.to_points() + [h])
    assert joined == bytes(fresh.text())[:-2]


@pytest.mark.parametrize("n", [2048, 2049, 8192, 8193, 16385, 32768, 32769])
def test_text_offsets_across_the_scan_forms(vm, n):
    """the formatter's text offsets are an exclusive scan of the items' lengths (csrc/scan.h): one tile, one workgroup
    (2049 .. 8192 items with 8 per thread, .. 32768 with 32) and the three-kernel form must give the same text - checked against Python's own decimals"""
    rng = random.Random(n)
    vals = [rng.randrange(ELL) >> rng.choice([0, 0, 100, 200, 250]) for _ in range(n)]
    sv = vm.ScalarVector.from_ints(vals)
    signed = [v if v <= ELL // 2 else v - ELL for v in vals] if vm.formats.scalar_signed() else vals
    assert bytes(sv.text()).decode() == "".join(f"{v}, " for v in signed)
