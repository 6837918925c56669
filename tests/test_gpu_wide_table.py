"""The wide-window fixed-base table (round 6; csrc/msm.hip msm_table_batch_wide, csrc/msm_sort.hip wide path): 13 rows
spaced 20 bits, one set of 2^19 buckets per commitment - 13 mixed additions per term of pivot.vector_commitment
(verifiable_mpc/ac20/pivot.py:139-145) instead of 16, no window recombination.  Through the C-ABI against the oracle's
restatement of the reference at sizes it finishes in seconds; against the 16-row table, the exponent identity and the
C oracle (reference algorithm, threaded) above that; digits at the window boundaries, skewed scalars, prefixes of the
table, extras, passes of several commitments."""
import random

import numpy as np
import pytest

from oracle import c_oracle
from oracle import ed25519_ref as ed
from tests.test_gpu_cabi import aff_bytes, gpu_points, make_points, sc_bytes

pytestmark = pytest.mark.gpu
ELL, P = ed.ELL, ed.P
WIDE = 13


@pytest.fixture(scope="module")
def nat():
    from verifiable_mpc_amd import _native
    n, info = _native.backend_info()
    assert n >= 1, info
    return _native


@pytest.fixture()
def ctx(nat):
    c = nat.Context(0)
    yield c
    c.close()


def ext_affine(raw):
    X, Y, Z, T = (int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(4))
    assert (X * Y - T * Z) % P == 0, "not an extended point"
    return ed.pt_affine((X, Y, Z))


def stages(ctx, fn):
    ctx.profile(True)
    ctx.profile_read(reset=True)
    fn()
    ctx.sync()
    st = {k for k, (_, launches) in ctx.profile_read(reset=True).items() if launches}
    ctx.profile(False)
    return st


@pytest.mark.parametrize("n,n_extra", [(1, 0), (5, 1), (300, 2), (8190, 2), (9000, 1)])
def test_wide_table_matches_oracle(nat, ctx, n, n_extra):
    """(8190 + 2 extras fill one 8192-column sort chunk per row exactly; 9000 needs two)"""
    rng = random.Random(1300 + n)
    exps = [rng.randrange(1, ELL) for _ in range(n + n_extra)]
    dp = gpu_points(nat, ctx, exps[:n])
    de = gpu_points(nat, ctx, exps[n:]) if n_extra else None
    table = ctx.msm_table_build(dp.ptr, n, de.ptr if de else None, n_extra, WIDE)
    x = [[rng.randrange(ELL) for _ in range(n)] for _ in range(3)]
    # digits at the edges of the 20-bit windows: 2^19 - 1, 2^19 (first negative digit, carry), all-ones runs, ...
    edge = [0, 1, ELL - 1, 2, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, (1 << 20) - 1, 1 << 20, (1 << 40) - (1 << 19),
            ELL // 2, ELL // 2 + 1, (1 << 252) + 5, (1 << 240) - 1, 1 << 239, ((1 << 252) - 1) // 3]
    for i, v in enumerate(edge):
        if i < n:
            x[0][i] = v
    gam = [[rng.randrange(ELL) for _ in range(n_extra)] for _ in range(3)]
    ds = [ctx.upload(sc_bytes(nat, v)) for v in x]
    dg = [ctx.upload(sc_bytes(nat, v)) for v in gam] if n_extra else None
    out = ctx.alloc(128 * 3)

    def want(k, m, use_extra):          # as exponents of the base point: every generator is exps[i] * B
        tot = sum(a * b for a, b in zip(x[k][:m], exps[:m]))
        if use_extra:
            tot += sum(a * b for a, b in zip(gam[k], exps[n:]))
        return ed.pt_affine(ed.pt_repeat(ed.BASE, tot % ELL))
    for m in sorted({n, n // 2, 1, 0}):
        for use_extra in ([True, False] if n_extra else [False]):
            st = stages(ctx, lambda: ctx.msm_table(table.ptr, n, n_extra, ds[0].ptr, m,
                                                   dg[0].ptr if use_extra else None, out.ptr, None, WIDE))
            assert "msm_bucket" in st and "short_bins" not in st
            assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == want(0, m, use_extra), (m, use_extra)
            for K in (2, 3):
                ctx.msm_table_batch(table.ptr, n, n_extra, [d.ptr for d in ds[:K]], m,
                                    [d.ptr for d in dg[:K]] if use_extra else None, out.ptr, None, WIDE)
                raw = ctx.download(out.ptr, 128 * K).tobytes()
                assert [ext_affine(raw[128 * k:128 * k + 128]) for k in range(K)] == \
                    [want(k, m, use_extra) for k in range(K)], (m, use_extra, K)
    # affine output
    oa = ctx.alloc(64)
    ctx.msm_table(table.ptr, n, n_extra, ds[1].ptr, n, None, None, oa.ptr, WIDE)
    raw = ctx.download(oa.ptr, 64).tobytes()
    assert (int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little")) == want(1, n, False)


def test_wide_table_rows_are_the_multiples(nat, ctx):
    """row r of the table is 2^(20 r) times the generator (checked through commitments to the scalars 2^(20 r))"""
    rng = random.Random(20)
    exps = [rng.randrange(1, ELL) for _ in range(3)]
    dp = gpu_points(nat, ctx, exps)
    table = ctx.msm_table_build(dp.ptr, 3, None, 0, WIDE)
    out = ctx.alloc(128)
    for r in range(13):
        for col in range(3):
            x = [0, 0, 0]
            x[col] = 1 << (20 * r)
            ctx.msm_table(table.ptr, 3, 0, ctx.upload(sc_bytes(nat, x)).ptr, 3, None, out.ptr, None, WIDE)
            assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == \
                ed.pt_affine(ed.pt_repeat(ed.BASE, (exps[col] << (20 * r)) % ELL))


@pytest.mark.parametrize("lg", [13, 16, 18])
def test_wide_table_distributions_against_the_16_row_table(nat, ctx, lg):
    n = (1 << lg) - 3
    rng = random.Random(lg)
    nrng = np.random.default_rng(lg)
    ea = nrng.integers(0, 256, size=(n + 2, 32), dtype=np.uint8)
    ea[:, 31] &= 0x0F
    exps = nat.array_to_ints(ea)
    pts = ctx.alloc(64 * (n + 2))
    base = ctx.upload(np.frombuffer(ed.affine_to_bytes(ed.BASE), np.uint8))
    ctx.fixed_base(base.ptr, ctx.upload(ea).ptr, n + 2, pts.ptr)
    wide = ctx.msm_table_build(pts.ptr, n, pts.ptr + 64 * n, 2, WIDE)
    t16 = ctx.msm_table_build(pts.ptr, n, pts.ptr + 64 * n, 2, 16)
    out = ctx.alloc(256)

    def uniform():
        a = nrng.integers(0, 256, size=(n, 32), dtype=np.uint8)
        a[:, 31] &= 0x0F
        return a
    dists = {"uniform": uniform()}
    # the [z] distribution of circuit_sat_cb.py:91-103: 54 % zeros, 9 % in {1, 2}, the rest uniform
    z = uniform()
    u = nrng.random(n)
    z[u < 0.54] = 0
    small = (u >= 0.54) & (u < 0.63)
    z[small] = 0
    z[small, 0] = nrng.integers(1, 3, size=int(small.sum()))
    dists["commitment"] = z
    bits = np.zeros((n, 32), np.uint8)
    bits[:, 0] = nrng.integers(0, 2, size=n)                      # everything in ONE bucket (heavily split)
    dists["bits"] = bits
    same = np.tile(uniform()[:1], (n, 1))                        # 13 buckets take everything
    dists["same_scalar"] = same
    gam = ctx.upload(sc_bytes(nat, [rng.randrange(ELL), rng.randrange(ELL)]))
    for name, arr in dists.items():
        ds = ctx.upload(arr)
        ctx.msm_table(wide.ptr, n, 2, ds.ptr, n, gam.ptr, out.ptr, None, WIDE)
        ctx.sync()
        got = ext_affine(ctx.download(out.ptr, 128).tobytes())
        ctx.set_short_path(False)
        ctx.msm_table(t16.ptr, n, 2, ds.ptr, n, gam.ptr, out.ptr + 128, None, 16)
        ctx.sync()
        ctx.set_short_path(True)
        assert got == ext_affine(ctx.download(out.ptr + 128, 128).tobytes()), name
    # exponent identity for the uniform vector (independent of every table)
    xs = nat.array_to_ints(dists["uniform"])
    ctx.msm_table(wide.ptr, n, 2, ctx.upload(dists["uniform"]).ptr, n, None, out.ptr, None, WIDE)
    assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == \
        ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(xs, exps)) % ELL))


def test_wide_table_2_16_bit_exact_vs_reference_algorithm(nat):
    """BASELINE config 2's size through pivot.vector_commitment over a 13-row table, against the C restatement of the
    reference algorithm (ladder per term + product tree)"""
    import verifiable_mpc_amd as vm
    n = 1 << 16
    rng = np.random.default_rng(1316)
    group = vm.EllipticCurve("Ed25519", "projective")

    def rs(k):
        a = rng.integers(0, 256, size=(k, 32), dtype=np.uint8)
        a[:, 31] &= 0x0F
        return a
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
    h = group.generator
    g.precompute([h], rows=WIDE)
    assert g._table.rows == WIDE
    sc = rs(n)
    sc[:50] = 0
    gamma = rs(1)[0]
    got = vm.pivot.vector_commitment(vm.ScalarVector.from_array(sc), int.from_bytes(gamma.tobytes(), "little"), g, h)
    prev = c_oracle.set_threads(c_oracle.host_threads())
    try:
        _, want = c_oracle.vector_commitment(sc, gamma, g.affine_array(), np.frombuffer(h.to_affine_bytes(), np.uint8))
    finally:
        c_oracle.set_threads(prev)
    assert got.to_affine_bytes() == bytes(want)
    # a prefix of the tabulated vector, and a pair in one pass
    m = 40000
    a, b = vm.pivot.vector_commitment_pair(vm.ScalarVector.from_array(sc[:m]), 5, g[:m],
                                           vm.ScalarVector.from_array(sc[m:2 * m]), 6, g[:m], h)
    plain = vm.PointVector(g.a, None, g.ctx)
    assert a == vm.pivot.vector_commitment(vm.ScalarVector.from_array(sc[:m]), 5, plain[:m], h)
    assert b == vm.pivot.vector_commitment(vm.ScalarVector.from_array(sc[m:2 * m]), 6, plain[:m], h)


def test_the_prover_does_not_take_a_wide_table(nat):
    """the wide table serves commitments; the round context (fold jump: rows spaced 256 / rows bits) refuses it and
    the Python prover goes its table-free way - same proof"""
    import verifiable_mpc_amd as vm
    rng = random.Random(77)
    n = 255
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    r = [rng.randrange(ELL) for _ in range(n)]
    proofs = []
    for rows in (None, WIDE):
        g = vm.PointVector.fixed_base(h, exps, keep_proj=False)
        if rows:
            g.precompute([h, k], rows=rows)
        gens = {"g": g, "h": h, "k": k}
        xs, L = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
        Pc = vm.pivot.vector_commitment(xs, 99, g, h)
        y = gf(L(xs))
        proof = vm.compressed_pivot.protocol_5_prover(gens, Pc, L, y, xs, 99, gf, transcript="compact", r=list(r), rho=7)
        assert vm.compressed_pivot.protocol_5_verifier(gens, Pc, L, y, proof, gf, transcript="compact") is True
        proofs.append({key: (v.to_affine_bytes() if hasattr(v, "to_affine_bytes") else [int(e) for e in v]
                             if isinstance(v, list) else int(v)) for key, v in proof.items()})
    assert proofs[0] == proofs[1]
    with pytest.raises(nat.VmpcError):
        ctx = vm.get_context()
        t = g._table
        nat.P4Rounds(ctx, t, 0, t.extra_index(k), ctx.alloc(32 * 256).ptr, ctx.alloc(32 * 256).ptr)


@pytest.mark.parametrize("log_n", [12, 13])
def test_round_context_commits_over_the_wide_table(nat, monkeypatch, log_n):
    """vmpc_p4_set_commit_table: a CRS that holds the wide-window table BESIDE its 16-bit-window one
    (PointVector.precompute(wide=True), what circuit_sat.create_generators builds from 2^19 generators) gives the very
    proof of a CRS without it - A (Protocol 5), every A_i / B_i of the round context, z' - and the commitments really
    run over the 13-row table"""
    import verifiable_mpc_amd as vm
    monkeypatch.setenv("VMPC_P4_COMMIT_TABLE_MIN_LOG2", "0")      # (default: only tables beyond the short path's 2^17 columns)
    rng = random.Random(log_n)
    n = (1 << log_n) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    r = [rng.randrange(ELL) for _ in range(n)]
    proofs = []
    for wide in (False, True):
        g = vm.PointVector.fixed_base(h, exps, keep_proj=False)
        g.precompute([h, k], wide=wide)
        assert (g._wide is not None) == wide and (not wide or g._wide.rows == WIDE)
        gens = {"g": g, "h": h, "k": k}
        xs, L = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
        Pc = vm.pivot.vector_commitment(xs, 99, g, h)
        y = gf(L(xs))
        ctx = g.ctx
        ctx.profile(True)
        ctx.profile_read(reset=True)
        proof = vm.compressed_pivot.protocol_5_prover(gens, Pc, L, y, xs, 99, gf, transcript="compact", r=list(r), rho=7)
        st = {k_: c for k_, (_, c) in ctx.profile_read(reset=True).items() if c}
        ctx.profile(False)
        if wide:        # A (Protocol 5) + the pairs of the rounds down to 2^11 generators (then k_p4_direct) over the
            assert "short_bins" not in st and st.get("msm_bucket", 0) >= 1 + (log_n - 11), st    # general pipeline
        else:
            assert "short_bins" in st, st
        assert vm.compressed_pivot.protocol_5_verifier(gens, Pc, L, y, proof, gf, transcript="compact") is True
        proofs.append({key: (v.to_affine_bytes() if hasattr(v, "to_affine_bytes") else [int(e) for e in v]
                             if isinstance(v, list) else int(v)) for key, v in proof.items()})
    assert proofs[0] == proofs[1]
