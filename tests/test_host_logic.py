"""CPU tests of the host-side mirror of the reference interface (no kernels involved):
forms, field/group glue, the streaming Fiat-Shamir hash, install()."""
import random

import pytest

from oracle import ac20_ref as ac
from oracle import ed25519_ref as ed

import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pivot


def test_linear_form_known_answers():
    # the reference's own known answers: ac20/test/test_pivot.py:84-90
    lf = pivot.LinearForm([0, 1, 2])
    assert (lf + lf + 2 * lf + lf.eval([1, 1, 1]) - lf).eval([1, 2, 3]) == 27
    assert lf([1, 2, 3]) == 8
    assert isinstance(lf + lf, pivot.AffineForm) and not isinstance(lf + lf, pivot.LinearForm)
    assert repr(pivot.AffineForm([1, 2], 5)) == "[1, 2], 5"
    assert pivot.LinearForm([1, 2], 9).constant == 0
    with pytest.raises(AssertionError):
        lf([1, 2])
    assert sum([lf, lf]) == lf * 2


def test_field_elements_follow_signed_repr():
    gf = vm.GF(ed.ELL)
    a = gf(ed.ELL - 5)
    assert repr(a) == "-5" and int(a) == -5 and a.value == ed.ELL - 5
    assert repr(gf(7) * 3 + 1) == "22" and (gf(1) / gf(3)) * 3 == 1 and gf(2) ** -1 == gf(1) / 2
    assert ed.scalar_repr(ed.ELL - 5) == repr(a)
    assert pivot._int(a) == -5 and pivot._int(9) == 9


def test_host_point_matches_oracle_representatives():
    rng = random.Random(1)
    G = vm.Ed25519Point.generator
    for _ in range(5):
        a, b = rng.randrange(ed.ELL), rng.randrange(ed.ELL)
        pa, pb = vm.Ed25519Point.repeat(G, a), vm.Ed25519Point.repeat(G, -b)
        assert pa.coords == ed.pt_repeat(ed.BASE, a) and pb.coords == ed.pt_repeat(ed.BASE, -b)
        s = vm.Ed25519Point.operation(pa, pb)
        assert s.coords == ed.pt_add(pa.coords, pb.coords)
        assert repr(s) == ed.pt_repr(s.coords)
        assert s.normalize().coords == ed.pt_normalize(s.coords)
        assert s == s.normalize() and hash(s) == hash(s.normalize())
    assert vm.Ed25519Point.from_affine_bytes(pa.to_affine_bytes()) == pa
    assert vm.Ed25519Point.from_proj_bytes(pa.to_proj_bytes()).coords == pa.coords
    with pytest.raises(ValueError):
        vm.Ed25519Point((1, 2, 1), check=True)


def test_operator_flags_like_mpyc(monkeypatch):
    G = vm.Ed25519Point.generator
    assert (G + G) == vm.Ed25519Point.repeat(G, 2) and (3 * G) == vm.Ed25519Point.repeat(G, 3)
    monkeypatch.setattr(vm.Ed25519Point, "is_multiplicative", True)
    monkeypatch.setattr(vm.Ed25519Point, "is_additive", False)
    assert (G * G) == G ** 2 and (G ** -1) * G == vm.Ed25519Point.identity


def test_streaming_hash_equals_str_of_list():
    gf = vm.GF(ed.ELL)
    G = vm.Ed25519Point.generator
    pts = [vm.Ed25519Point.repeat(G, i + 2) for i in range(3)]
    form = pivot.AffineForm([gf(3), 10**80, gf(-1)], gf(0))
    lst = [gf(-7), pts[0].normalize(), {"g": pts, "h": G, "k": pts[1]}, form, 0, 1,
           "First hash of compressed pivot", [], [[1, 2], []]]
    want = ac.fiat_shamir_hash_text(str(lst), ed.ELL)
    assert pivot.fiat_shamir_hash(lst, ed.ELL) == want


def test_install_patches_reference_modules(monkeypatch):
    import sys
    import types
    pkg = "fake_ref_pkg"
    mods = {}
    for name in ("", ".pivot", ".compressed_pivot", ".circuit_sat_r1cs", ".circuit_sat_cb"):
        m = types.ModuleType(pkg + name)
        mods[pkg + name] = m
        monkeypatch.setitem(sys.modules, pkg + name, m)
    original_calls = []
    mods[pkg + ".pivot"].vector_commitment = lambda x, gamma, g, h: original_calls.append(h) or "theirs"
    patched = vm.install(pkg)
    assert mods[pkg + ".pivot"].vector_commitment.__vmpc_accelerated__ is pivot.vector_commitment
    assert mods[pkg + ".compressed_pivot"].protocol_5_prover.__vmpc_accelerated__ is vm.compressed_pivot.protocol_5_prover
    assert mods[pkg + ".circuit_sat_r1cs"].create_generators.__vmpc_accelerated__ is vm.circuit_sat.create_generators
    assert mods[pkg + ".pivot"].prove_linear_form_eval.__vmpc_accelerated__ is pivot.prove_linear_form_eval
    assert len(patched) == 10
    # a base that is not an Ed25519 element goes to what was there before; installing twice wraps once
    assert mods[pkg + ".pivot"].vector_commitment([1], 2, [5], 7) == "theirs" and original_calls == [7]
    before = mods[pkg + ".pivot"].vector_commitment
    vm.install(pkg)
    assert mods[pkg + ".pivot"].vector_commitment is before
    assert f"{pkg}.pivot.vector_commitment" in vm.uninstall(pkg)
    assert mods[pkg + ".pivot"].vector_commitment([1], 2, [5], 8) == "theirs"
    assert not hasattr(mods[pkg + ".pivot"], "fiat_shamir_hash")


def test_group_of_a_call_is_read_off_its_arguments():
    """what decides between the GPU path and the reference's own function (dropin.py): order l over GF(2^255 - 19)
    with three coordinates - not the Python class"""
    import types
    from verifiable_mpc_amd import groups
    G = vm.Ed25519Point.generator
    assert groups.is_ed25519_element(G) and groups.is_ed25519_group(vm.Ed25519Point)

    def foreign(order, modulus, ncoords):
        cls = type("Foreign", (), {"order": order, "field": types.SimpleNamespace(modulus=modulus)})
        obj = cls()
        obj.value = [types.SimpleNamespace(value=1)] * ncoords
        return obj
    assert groups.is_ed25519_element(foreign(groups.ORDER, groups.P, 3))
    assert not groups.is_ed25519_element(foreign(groups.ORDER, groups.P, 4))        # 'extended' coordinates
    assert not groups.is_ed25519_element(foreign(groups.ORDER, groups.P, 2))        # 'affine'
    assert not groups.is_ed25519_element(foreign(groups.ORDER + 2, groups.P, 3))    # BN256 (jacobian: 3 coordinates)
    assert not groups.is_ed25519_element(7) and not groups.is_ed25519_element(None)
    import enum
    theirs = enum.Enum("PivotChoice", "pivot compressed koe")
    assert vm.circuit_sat.choice_name(theirs.compressed) == "compressed" == vm.circuit_sat.choice_name("compressed")
    with pytest.raises(NotImplementedError):
        vm.create_generators(3, theirs.koe, vm.Ed25519Point)
    with pytest.raises(NotImplementedError, match="only Ed25519"):
        vm.create_generators(3, theirs.compressed, types.SimpleNamespace(generator=foreign(11, 23, 1), order=11))


def test_circuit_sat_harness_names_delegate_to_the_reference(monkeypatch):
    """circuit_sat_prover / circuit_sat_verifier (circuit_sat_cb.py:255,285) exist under this package's names, install the
    hot path into the reference's modules and call the reference's own function with the reference's enum member"""
    import enum
    import sys
    import types
    pkg = "fake_ref_pkg2"
    mods = {}
    for name in ("", ".pivot", ".compressed_pivot", ".circuit_sat_r1cs", ".circuit_sat_cb"):
        m = types.ModuleType(pkg + name)
        mods[pkg + name] = m
        monkeypatch.setitem(sys.modules, pkg + name, m)
    cb = mods[pkg + ".circuit_sat_cb"]
    cb.PivotChoice = enum.Enum("PivotChoice", "pivot compressed koe")
    calls = []
    cb.circuit_sat_prover = lambda gens, circuit, x, gf, choice: calls.append(("p", choice)) or {"proof": 1}
    cb.circuit_sat_verifier = lambda proof, gens, circuit, gf, choice: calls.append(("v", choice)) or {"ok": True}
    monkeypatch.setattr(vm.circuit_sat, "REFERENCE_PACKAGE", pkg)
    assert vm.circuit_sat_prover({"g": []}, None, [1], None) == {"proof": 1}
    assert vm.circuit_sat_verifier({"proof": 1}, {"g": []}, None, None, vm.PivotChoice.pivot) == {"ok": True}
    assert calls == [("p", cb.PivotChoice.compressed), ("v", cb.PivotChoice.pivot)]
    assert mods[pkg + ".pivot"].vector_commitment.__vmpc_accelerated__ is pivot.vector_commitment    # install() ran
    monkeypatch.setattr(vm.circuit_sat, "REFERENCE_PACKAGE", "no_such_reference_pkg")
    with pytest.raises(ImportError, match="circuit front end"):
        vm.circuit_sat_prover({}, None, [], None)


def test_product_never_imports_the_oracle():
    import os
    root = os.path.dirname(os.path.abspath(vm.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text, f


def test_commitment_fold_on_the_host_matches_the_oracle():
    """vmpc_ed25519_fold_commitment_host (host arithmetic inside the HIP library, no device): Q' = A * Q**c * B**(c**2)
    of compressed_pivot.py:66 as the oracle computes it, normalised - what the reference transcript hashes"""
    import random
    from oracle import ed25519_ref as ed
    rng = random.Random(180)
    for trial in range(6):
        pts = [ed.pt_repeat(ed.BASE, rng.randrange(1, ed.ELL)) for _ in range(3)]
        c = [rng.randrange(ed.ELL), 0, 1, ed.ELL - 1, rng.randrange(ed.ELL), 2][trial]
        want = ed.pt_affine(ed.pt_add(ed.pt_add(pts[0], ed.pt_repeat(pts[1], c)), ed.pt_repeat(pts[2], c * c)))
        enc = [b"".join(v.to_bytes(32, "little") for v in ed.pt_affine(p)) for p in pts]
        raw = vm._native.fold_commitment_host(enc[0], enc[1], enc[2], c)
        assert (int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little")) == want
    with pytest.raises(vm._native.VmpcError):           # a challenge that is not a canonical residue
        lib = vm._native.load_library()
        import ctypes
        out = ctypes.create_string_buffer(64)
        vm._native._check(lib.vmpc_ed25519_fold_commitment_host(enc[0], enc[1], enc[2], (ed.ELL).to_bytes(32, "little"), out),
                          "vmpc_ed25519_fold_commitment_host")


def test_missing_gpu_fails_loudly():
    n, _ = vm._native.backend_info()
    if n >= 1:
        pytest.skip("GPU present")
    vm.device.reset_context()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vm.get_context()
    with pytest.raises(RuntimeError):
        pivot.vector_commitment([1], 1, [vm.Ed25519Point.generator], vm.Ed25519Point.generator)


def test_mpc_party_runtime_opens_and_recombines():
    """verifiable_mpc_amd.mpc_ac20 host logic (no GPU): Shamir dealing, the M-point Lagrange weights,
    linear arithmetic on shares, `output` across three in-process parties."""
    import asyncio
    import random
    from verifiable_mpc_amd import mpc_ac20
    ELL = ed.ELL
    rng = random.Random(9)
    parties, threshold = 3, 1
    hub = mpc_ac20.LocalHub(parties)
    rts = [mpc_ac20.PartyRuntime(p, parties, threshold, random.Random(p), hub) for p in range(parties)]
    assert [rt.lagrange for rt in rts] == mpc_ac20.recombination_vector([1, 2, 3])
    assert sum(rt.lagrange for rt in rts) % ELL == 1
    secrets = [rng.randrange(ELL) for _ in range(4)]
    shares = mpc_ac20.deal(secrets, threshold, parties, rng)
    # any threshold + 1 = 2 parties recombine as well
    lam2 = mpc_ac20.recombination_vector([1, 3])
    assert (lam2[0] * shares[0][0] + lam2[1] * shares[2][0]) % ELL == secrets[0]
    gf = vm.GF(ELL)

    async def party(rt):
        a, b, c, d = (rt.secret(s) for s in shares[rt.pid])
        expr = 3 * a - b + gf(7) * c + 11          # linear: local on shares
        fresh = rt._random()
        opened = await rt.output([expr, d, 5, fresh])
        single = await rt.output(a + d)
        return opened, single

    async def everybody():
        return await asyncio.gather(*[party(rt) for rt in rts])
    results = asyncio.new_event_loop().run_until_complete(everybody())
    want = (3 * secrets[0] - secrets[1] + 7 * secrets[2] + 11) % ELL
    for opened, single in results:
        assert int(opened[0]) % ELL == want and int(opened[1]) % ELL == secrets[3] and opened[2] == 5
        assert opened[3] == results[0][0][3]                   # the jointly random value is the same everywhere
        assert int(single) % ELL == (secrets[0] + secrets[3]) % ELL
    with pytest.raises(NotImplementedError):
        rts[0].secret(1) * rts[0].secret(2)
    assert pivot._int(rts[0].secret(1)).share == 1            # pivot.py:119-128: secure objects pass through


def test_install_mpc_patches_reference_module(monkeypatch):
    import sys
    import types
    from verifiable_mpc_amd import mpc_ac20
    pkg = "fake_ref_pkg_mpc"
    for name in ("", ".mpc_ac20"):
        monkeypatch.setitem(sys.modules, pkg + name, types.ModuleType(pkg + name))
    patched = mpc_ac20.install_mpc(pkg)
    assert sys.modules[pkg + ".mpc_ac20"].protocol_5_prover is mpc_ac20.protocol_5_prover
    assert len(patched) == 4


def test_hash_input_dump_streams_str_of_the_list():
    """compressed_pivot.py:56-58,122-124: logger "compressed_pivot_hash_inputs" at DEBUG gets the pre-image; here in
    pieces (one record per LOG_PIECE bytes) that concatenate to str(input_list); at INFO nothing is formatted"""
    import logging
    lg = logging.getLogger("compressed_pivot_hash_inputs")
    records = []

    class Keep(logging.Handler):
        def emit(self, record):
            records.append(record.getMessage())
    handler = Keep()
    lg.addHandler(handler)
    G = vm.Ed25519Point.generator
    lst = [G, {"g": [G, G]}, "s", pivot.LinearForm([1, 2])]
    try:
        pivot.log_hash_input(lg, "protocol_4_prover", lst)
        assert records == []
        lg.setLevel(logging.DEBUG)
        old, pivot._LogSink.LOG_PIECE = pivot._LogSink.LOG_PIECE, 64
        pivot.log_hash_input(lg, "protocol_4_prover", lst)
        pivot._LogSink.LOG_PIECE = old
    finally:
        lg.setLevel(logging.INFO)
        lg.removeHandler(handler)
    assert records[0].startswith("Method protocol_4_prover: Before fiat_shamir_hash, input_list=\n[[1511")
    assert len(records) > 3 and "".join(r.split("=\n", 1)[1] for r in records) == str(lst)
