"""GPU parity at BASELINE.json's configured sizes.

config 2 (n = 2^16 MSM): bit-exact against the C restatement of the REFERENCE algorithm
(oracle/ed25519_oracle.c: one double-and-add ladder per term + product tree), ~9 s of CPU.
config 3 (N = 2^20): size-independent properties - the generator-exponent identity,
linearity of the commitment in the scalars, prove -> verify round trip and rejection of
tampered proofs in both transcript modes' code paths.
"""
import random

import numpy as np
import pytest

from oracle import c_oracle

pytestmark = pytest.mark.gpu

ELL = 2**252 + 27742317777372353535851937790883648493


def rand_scalars(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F
    return a


@pytest.fixture(scope="module")
def vm():
    import verifiable_mpc_amd as v
    v.get_context()
    return v


def test_config2_msm_2_16_bit_exact_vs_reference_algorithm(vm):
    n = 1 << 16
    rng = np.random.default_rng(216)
    group = vm.EllipticCurve("Ed25519", "projective")
    exps = rand_scalars(rng, n)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(exps), keep_proj=False)
    pts = g.affine_array()
    # spot-check the device-made generators against the oracle (first 64)
    base = np.frombuffer(group.generator.to_proj_bytes(), np.uint8)
    _, want_pts = c_oracle.fixed_base(base, exps[:64])
    assert (pts[:64] == want_pts).all()
    sc = rand_scalars(rng, n)
    sc[:100] = 0                                  # zeros
    sc[100:200] = 0
    sc[100:200, 0] = 1                            # ones
    gamma = rand_scalars(rng, 1)[0]
    x = vm.ScalarVector.from_array(sc)
    got = vm.pivot.vector_commitment(x, int.from_bytes(gamma.tobytes(), "little"), g, group.generator)
    _, want = c_oracle.vector_commitment(sc, gamma, pts, np.frombuffer(group.generator.to_affine_bytes(), np.uint8))
    assert got.to_affine_bytes() == bytes(want)


def test_config3_msm_2_20_properties(vm):
    n = 1 << 20
    rng = np.random.default_rng(220)
    group = vm.EllipticCurve("Ed25519", "projective")
    exps = rand_scalars(rng, n)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(exps), keep_proj=False)
    s1, s2 = rand_scalars(rng, n), rand_scalars(rng, n)
    x1, x2 = vm.ScalarVector.from_array(s1), vm.ScalarVector.from_array(s2)
    ident = vm.Ed25519Point.identity
    c1 = vm.pivot.vector_commitment(x1, 0, g, ident)
    c2 = vm.pivot.vector_commitment(x2, 0, g, ident)
    # (1) sum_i s_i (e_i B) == (sum_i s_i e_i) B
    e_int = vm._native.array_to_ints(exps)
    s_int = vm._native.array_to_ints(s1)
    tot = sum(a * b for a, b in zip(s_int, e_int)) % ELL
    assert c1 == vm.Ed25519Point.repeat(group.generator, tot)
    # (2) linearity: commit(a*s1 + s2) == a*commit(s1) + commit(s2)
    a = random.Random(5).randrange(ELL)
    c12 = vm.pivot.vector_commitment(x1.axpy(a, x2), 0, g, ident)
    assert c12 == vm.Ed25519Point.operation(vm.Ed25519Point.repeat(c1, a), c2)
    # (3) window-width independence at full size
    ctx = vm.get_context()
    ctx.set_window(13)
    try:
        assert vm.pivot.vector_commitment(x1, 0, g, ident) == c1
    finally:
        ctx.set_window(0)


@pytest.mark.parametrize("mode", ["compact", "reference"])
def test_config3_protocol5_2_20_round_trip(vm, mode):
    # BASELINE config 3 in BOTH transcripts; the reference transcript (the mode north_star grades: proofs
    # bit-identical to the CPU reference) hashes ~1 GB of decimal text per prove / verify at this size
    N = 1 << 20
    n = N - 1
    rng = np.random.default_rng(2020 + len(mode))
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rand_scalars(rng, n)),
                                  keep_proj=(mode == "reference"))
    gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 987654321)}
    x = vm.ScalarVector.from_array(rand_scalars(rng, n))
    L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rand_scalars(rng, n)))
    y = gf(L(x))
    gamma = 31337
    P = vm.pivot.vector_commitment(x, gamma, g, gens["h"])
    r = vm.ScalarVector.from_array(rand_scalars(rng, n))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf, transcript=mode, r=r, rho=99)
    rounds = N.bit_length() - 2
    assert set(proof) == {"t", "A", "z_prime"} | {f"A{i}" for i in range(rounds)} | {f"B{i}" for i in range(rounds)}
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript=mode) is True
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y + 1, proof, gf, transcript=mode) is False
    bad = dict(proof)
    bad[f"B{rounds // 2}"] = proof["A0"]
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, bad, gf, transcript=mode) is False


def test_config3_tabulated_crs_2_20(vm):
    """fixed-base tables at full size: a commitment over the tabulated CRS equals the variable-base one
    (every row count), and the fold-free compact prover produces the very proof of the folding prover."""
    N = 1 << 20
    n = N - 1
    rng = np.random.default_rng(77)
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, 424242)
    g = vm.PointVector.fixed_base(h, vm.ScalarVector.from_array(rand_scalars(rng, n)), keep_proj=False)
    gens = {"g": g, "h": h, "k": k}
    x = vm.ScalarVector.from_array(rand_scalars(rng, n))
    L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rand_scalars(rng, n)))
    y, gamma = gf(L(x)), 271828
    P = vm.pivot.vector_commitment(x, gamma, g, h)                       # variable base
    r = rand_scalars(rng, n)
    plain = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf, transcript="compact",
                                                  r=vm.ScalarVector.from_array(r), rho=7)

    def flat(proof):
        return {key: (v.to_affine_bytes() if hasattr(v, "to_affine_bytes") else [int(e) for e in v]
                      if isinstance(v, list) else int(v)) for key, v in proof.items()}
    for rows in (None, 16, 1, 13):           # (13: the wide-window table, commitments only)
        g.precompute([h, k], rows=rows)
        assert vm.pivot.vector_commitment(x, gamma, g, h) == P
        assert vm.pivot.vector_commitment(x[:12345], 5, g, k) == vm.pivot.vector_commitment(
            x[:12345], 5, vm.PointVector(g.a, None, g.ctx), k)
        if rows not in (16, 13):
            tab = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf, transcript="compact",
                                                        r=vm.ScalarVector.from_array(r), rho=7)
            assert flat(tab) == flat(plain)
            assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, tab, gf, transcript="compact") is True


def test_config4_shard_size_msm_2_21_properties(vm):
    """n = 2^21 is one GPU's share of BASELINE config 4 (2^24 terms over 8 GPUs): longer segments, more
    sort slices.  Exponent identity and window-width independence."""
    n = 1 << 21
    rng = np.random.default_rng(221)
    group = vm.EllipticCurve("Ed25519", "projective")
    exps = rand_scalars(rng, n)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(exps), keep_proj=False)
    s1 = rand_scalars(rng, n)
    x1 = vm.ScalarVector.from_array(s1)
    ident = vm.Ed25519Point.identity
    c1 = vm.pivot.vector_commitment(x1, 0, g, ident)
    tot = sum(a * b for a, b in zip(vm._native.array_to_ints(s1), vm._native.array_to_ints(exps))) % ELL
    assert c1 == vm.Ed25519Point.repeat(group.generator, tot)
    ctx = vm.get_context()
    ctx.set_window(14)
    try:
        assert vm.pivot.vector_commitment(x1, 0, g, ident) == c1
    finally:
        ctx.set_window(0)


def test_config4_total_size_2_24_as_eight_cyclic_shards(vm):
    """BASELINE config 4 at its TOTAL size on one GPU: the 2^24-term commitment computed whole equals the
    rank-ordered sum of the eight cyclic shards' partial commitments (2^21 terms each) - the arithmetic of the
    8-GPU run (parallel.ShardedMsm: cyclic shards, one partial point per rank, vmpc_points_sum_dev in rank order)
    minus the all-gather, which needs the eight GPUs."""
    from verifiable_mpc_amd import parallel
    n, world = 1 << 24, 8
    rng = np.random.default_rng(224)
    group = vm.EllipticCurve("Ed25519", "projective")
    ctx = vm.get_context()
    exps = rand_scalars(rng, n)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(exps), keep_proj=False)
    sc = rand_scalars(rng, n)
    x = vm.ScalarVector.from_array(sc)
    whole = vm.pivot.vector_commitment(x, 0, g, vm.Ed25519Point.identity)
    pts_host = g.affine_array()
    gathered = ctx.alloc(128 * world)
    for rank in range(world):
        ps = vm.PointVector.from_affine_array(parallel.shard_rows(pts_host, world, rank), validate=False)
        xs = vm.ScalarVector.from_array(parallel.shard_rows(sc, world, rank))
        assert len(ps) == n // world
        ctx.msm(xs.ptr, ps.affine_ptr, len(xs), None, None, 0, gathered.ptr + 128 * rank, None)
        ctx.sync()
        del ps, xs
    out = ctx.alloc(64)
    ctx.points_sum(gathered.ptr, world, None, out.ptr)
    ctx.sync()
    assert vm.Ed25519Point.from_affine_bytes(ctx.download(out.ptr, 64).tobytes()) == whole
    # the exponent identity on the WHOLE 2^24-term commitment: sum_i s_i (e_i B) == (sum_i s_i e_i mod l) B
    tot = 0
    step = 1 << 20
    for lo in range(0, n, step):
        tot += sum(a * b for a, b in zip(vm._native.array_to_ints(sc[lo:lo + step]),
                                         vm._native.array_to_ints(exps[lo:lo + step])))
    assert whole == vm.Ed25519Point.repeat(group.generator, tot % ELL)
