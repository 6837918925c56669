"""the MPyC-driver counterpart (verifiable_mpc_amd/mpc_ac20.py) on the GPU.

 (1) m = 1 against tests/golden/mpc_ac20_m1.json - the reference's own mpc_ac20.py
     (create_generators :45-51, vector_commitment :35-42, protocol_5_prover :206-269 ->
     protocol_4_prover :141-203) run over the single-party shim, the way
     test/test_demo_zkp_mpc_ac20.py:17-23 runs it;
 (2) M = 3 parties, threshold 1, as three coroutines on one GPU: every party derives the same proof,
     it equals the PLAIN prover's proof for the recombined masks, and the plain verifier accepts it.
No real MPyC exists on either machine: the party runtime is the stand-in of mpc_ac20.py."""
import asyncio
import random

import pytest

from oracle import ed25519_ref as ed
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

ELL = ed.ELL
hx = lambda v: format(int(v), "x")
h2i = lambda s: int(s, 16)


@pytest.fixture(scope="module")
def vm():
    import verifiable_mpc_amd as v
    v.get_context()
    return v


def aff_hex(pt):
    n = pt.normalize()
    return [hx(n.coords[0]), hx(n.coords[1])]


def run(coro):
    return asyncio.new_event_loop().run_until_complete(coro)


@pytest.mark.parametrize("idx", [0, 1])
def test_single_party_matches_reference_fixture(vm, monkeypatch, idx):
    from verifiable_mpc_amd import mpc_ac20
    case = load_golden("mpc_ac20_m1.json")["cases"][idx]
    n = case["n"]
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    rt = mpc_ac20.PartyRuntime(pid=0, parties=1, threshold=0, rng=random.Random(case["seed"] + 1), gf=gf)
    gens = run(mpc_ac20.create_generators(group, None, n, rt))
    assert [aff_hex(p) for p in gens["g"]] == case["generators"]["g"]
    assert aff_hex(gens["k"]) == case["generators"]["k"] and aff_hex(gens["h"]) == case["generators"]["h"]
    x = [rt.secret(h2i(v)) for v in case["x"]]
    gamma = rt.secret(h2i(case["gamma"]))
    L = vm.pivot.LinearForm([gf(h2i(v)) for v in case["L"]])
    P = run(mpc_ac20.vector_commitment(x, gamma, gens["g"], gens["h"]))
    assert aff_hex(P) == case["P"]
    y = L(x)
    assert isinstance(y, mpc_ac20.SecureScalar) and hx(y.share) == case["y"]
    calls = []
    orig_v = vm.pivot.fiat_shamir_hash_variants
    orig_h = vm.pivot.fiat_shamir_hash
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash_variants",
                        lambda c_, t_, o_: (lambda r_: (calls.extend(r_), r_)[1])(orig_v(c_, t_, o_)))
    monkeypatch.setattr(vm.pivot, "fiat_shamir_hash",
                        lambda i_, o_: (lambda r_: (calls.append(r_), r_)[1])(orig_h(i_, o_)))
    rt.rng = random.Random(case["seed"] + 2)
    proof = run(mpc_ac20.protocol_5_prover(gens, P, L, y, x, gamma, gf))
    assert [hx(c) for c in calls] == [h["c"] for h in case["hashes"]]
    assert list(proof.keys()) == case["proof_keys"]
    pr = case["proof"]
    assert hx(int(proof["t"]) % ELL) == pr["t"] and aff_hex(proof["A"]) == pr["A"]
    for i in range(case["rounds"]):
        assert aff_hex(proof[f"A{i}"]) == pr["A_i"][i] and aff_hex(proof[f"B{i}"]) == pr["B_i"][i]
    assert [hx(int(v) % ELL) for v in proof["z_prime"]] == pr["z_prime"]
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, gf(h2i(case["y"])), proof, gf) is True


@pytest.mark.parametrize("mode", ["reference", "compact"])
def test_three_parties_one_gpu(vm, mode):
    from verifiable_mpc_amd import mpc_ac20
    parties, threshold, n = 3, 1, 31
    rng = random.Random(4711)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    hub = mpc_ac20.LocalHub(parties)
    rts = [mpc_ac20.PartyRuntime(p, parties, threshold, random.Random(100 + p), hub, gf) for p in range(parties)]
    assert sum(rt.lagrange for rt in rts) % ELL == 1

    async def all_parties(fn):
        return await asyncio.gather(*[fn(rt) for rt in rts])

    gens_all = run(all_parties(lambda rt: mpc_ac20.create_generators(group, None, n, rt)))
    gens = gens_all[0]
    for other in gens_all[1:]:                                  # public output: identical everywhere
        assert [p.to_affine_bytes() for p in other["g"]] == [p.to_affine_bytes() for p in gens["g"]]
        assert other["k"] == gens["k"]
    # jointly random: g_i = h ** (sum_p lambda_p u_{p,i}), nobody drew the exponent
    replay = [random.Random(100 + p) for p in range(parties)]
    draws = [[r_.randrange(ELL) for _ in range(n + 1)] for r_ in replay]
    exps = [sum(rt.lagrange * d[i] for rt, d in zip(rts, draws)) % ELL for i in range(n + 1)]
    assert gens["k"] == vm.Ed25519Point.repeat(group.generator, exps[0])
    assert gens["g"][4] == vm.Ed25519Point.repeat(group.generator, exps[5])

    x = [rng.randrange(ELL) for _ in range(n)]
    gamma = rng.randrange(1, ELL)
    shares = mpc_ac20.deal(x + [gamma], threshold, parties, rng)
    L = vm.pivot.LinearForm([gf(rng.randrange(ELL)) for _ in range(n)])
    xs = [[rt.secret(s) for s in shares[p][:n]] for p, rt in enumerate(rts)]
    gs = [rt.secret(shares[p][n]) for p, rt in enumerate(rts)]
    Ps = run(all_parties(lambda rt: mpc_ac20.vector_commitment(xs[rt.pid], gs[rt.pid], gens["g"], gens["h"])))
    P = Ps[0]
    assert all(p == P for p in Ps)
    assert P == vm.pivot.vector_commitment([gf(v) for v in x], gamma, gens["g"], gens["h"])
    ys = [L(xs[p]) for p in range(parties)]
    y = gf(sum(int(c) * v for c, v in zip(L.coeffs, x)) % ELL)

    for p, rt in enumerate(rts):
        rt.rng = random.Random(200 + p)
    proofs = run(all_parties(lambda rt: mpc_ac20.protocol_5_prover(gens, P, L, ys[rt.pid], xs[rt.pid], gs[rt.pid],
                                                                    gf, transcript=mode)))
    ref = proofs[0]
    for other in proofs[1:]:
        assert list(other.keys()) == list(ref.keys())
        for key in ref:
            assert other[key] == ref[key], key
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, ref, gf, transcript=mode) is True
    # the same proof as the single-party prover run on the recombined secrets and masks
    replay = [random.Random(200 + p) for p in range(parties)]
    mdraws = [[r_.randrange(ELL) for _ in range(n + 1)] for r_ in replay]
    r = [sum(rt.lagrange * d[i] for rt, d in zip(rts, mdraws)) % ELL for i in range(n)]
    rho = sum(rt.lagrange * d[n] for rt, d in zip(rts, mdraws)) % ELL
    plain = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, [gf(v) for v in x], gamma, gf,
                                                  transcript=mode, r=r, rho=rho)
    assert list(plain.keys()) == list(ref.keys())
    for key in plain:
        if key == "z_prime":
            assert [int(v) % ELL for v in plain[key]] == [int(v) % ELL for v in ref[key]]
        elif key == "t":
            assert int(plain[key]) % ELL == int(ref[key]) % ELL
        else:
            assert plain[key] == ref[key], key


def test_secure_scalar_is_linear_only(vm):
    from verifiable_mpc_amd import mpc_ac20
    rt = mpc_ac20.PartyRuntime()
    a, b = rt.secret(5), rt.secret(7)
    gf = vm.GF(ELL)
    assert (a + b).share == 12 and (3 * a - b).share == 8 and (a * gf(-1) + 1).share == ELL - 4
    assert (10 - a).share == 5 and (-a).share == ELL - 5
    with pytest.raises(NotImplementedError):
        a * b
    assert vm.pivot._int(a) is a                       # pivot.py:119-128: secure objects pass through
