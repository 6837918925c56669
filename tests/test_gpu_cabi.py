"""GPU parity tests of the C-ABI kernels against the oracle (bit-exact, integer work).

Every call goes through libvmpc_hip.so via ctypes (verifiable_mpc_amd/_native.py).
Sizes are kept where the pure-Python oracle finishes in seconds; larger sizes are covered
by size-independent properties in tests/test_gpu_msm_large.py.
"""
import random

import numpy as np
import pytest

from oracle import ac20_ref as ac
from oracle import ed25519_ref as ed

pytestmark = pytest.mark.gpu

ELL, P = ed.ELL, ed.P


@pytest.fixture(scope="module")
def nat():
    from verifiable_mpc_amd import _native
    n, info = _native.backend_info()
    assert n >= 1, info
    return _native


@pytest.fixture(scope="module")
def ctx(nat):
    c = nat.Context(0)
    yield c
    c.close()


def aff_bytes(pts):
    return np.frombuffer(b"".join(ed.affine_to_bytes(p) for p in pts), dtype=np.uint8).reshape(-1, 64)


def proj_bytes(pts):
    return np.frombuffer(b"".join(ed.proj_to_bytes(p) for p in pts), dtype=np.uint8).reshape(-1, 96)


def sc_bytes(nat, vals):
    return nat.ints_to_array([v % ELL for v in vals], 32)


def make_points(rng, n):
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    return exps, [ed.pt_repeat(ed.BASE, e) for e in exps]


def dl_aff(ctx, ptr, n=1):
    raw = ctx.download(ptr, 64 * n).tobytes()
    return [ed.affine_from_bytes(raw[64 * i:64 * i + 64]) for i in range(n)]


def dl_proj(ctx, ptr, n=1):
    raw = ctx.download(ptr, 96 * n).tobytes()
    return [ed.proj_from_bytes(raw[96 * i:96 * i + 96]) for i in range(n)]


def test_backend_info(nat):
    n, info = nat.backend_info()
    assert "gfx950" in info, info


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 300])
def test_msm_matches_oracle_vector_commitment(nat, ctx, n):
    """vmpc_msm_dev == pivot.vector_commitment restated (pivot.py:139-145), incl. the
    h**gamma term passed as the extra segment."""
    rng = random.Random(100 + n)
    _, g = make_points(rng, n + 1)
    h, g = g[-1], g[:-1]
    x = [rng.randrange(ELL) for _ in range(n)]
    for i, v in enumerate([0, 1, ELL - 1, 2, ELL // 2, ELL // 2 + 1]):
        if i < n:
            x[i] = v
    gamma = rng.randrange(ELL)
    want = ed.pt_affine(ac.vector_commitment(x, gamma, g, h))
    ds, dp = ctx.upload(sc_bytes(nat, x)), ctx.upload(aff_bytes(g))
    dg, dh = ctx.upload(sc_bytes(nat, [gamma])), ctx.upload(aff_bytes([h]))
    out_ext, out_aff = ctx.alloc(128), ctx.alloc(64)
    ctx.msm(ds.ptr, dp.ptr, n, dg.ptr, dh.ptr, 1, out_ext.ptr, out_aff.ptr)
    ctx.sync()
    got = dl_aff(ctx, out_aff.ptr)[0]
    assert got[:2] == want
    # the extended output is the same group element
    raw = ctx.download(out_ext.ptr, 128).tobytes()
    X, Y, Z, T = (int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(4))
    assert ed.pt_affine((X, Y, Z)) == want and (X * Y - T * Z) % P == 0


@pytest.mark.parametrize("c_bits", [4, 5, 8, 11, 13, 16])
def test_msm_every_window_width(nat, ctx, c_bits):
    """the result must not depend on the Pippenger window; exercises carries in the signed
    recoding and the chunked bucket reduction at each width."""
    rng = random.Random(7)
    n = 200
    exps, g = make_points(rng, n)
    x = [rng.randrange(ELL) for _ in range(n)]
    x[0], x[1], x[2] = ELL - 1, 2**252, (1 << 252) - 1
    want = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL))
    ds, dp = ctx.upload(sc_bytes(nat, x)), ctx.upload(aff_bytes(g))
    out = ctx.alloc(64)
    ctx.set_window(c_bits)
    try:
        ctx.msm(ds.ptr, dp.ptr, n, None, None, 0, None, out.ptr)
        ctx.sync()
    finally:
        ctx.set_window(0)
    assert dl_aff(ctx, out.ptr)[0][:2] == want


def gpu_points(nat, ctx, exps):
    """g_i = e_i * B made on the device (vmpc_repeat_dev is checked against the oracle above)."""
    dbase, dexp = ctx.upload(aff_bytes([ed.BASE])), ctx.upload(sc_bytes(nat, exps))
    out = ctx.alloc(64 * len(exps))
    ctx.repeat(dbase.ptr, 1, True, dexp.ptr, len(exps), False, None, out.ptr)
    ctx.sync()
    return out


@pytest.mark.parametrize("dist", ["commitment", "all_ones", "two_values", "top_heavy"])
def test_msm_skewed_scalars(nat, ctx, dist):
    """heavy buckets (zeros / ones / repeated scalars, SURVEY.md 8d cfg 2 'commitment
    distribution') go through the segment split + workgroup finish path."""
    rng = random.Random(21)
    n = 6000
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    if dist == "commitment":       # 54 % zeros, 9 % in {1, 2}, 37 % uniform
        x = [0 if u < 0.54 else (rng.choice([1, 2]) if u < 0.63 else rng.randrange(ELL))
             for u in (rng.random() for _ in range(n))]
    elif dist == "all_ones":
        x = [1] * n
    elif dist == "two_values":
        a, b = rng.randrange(ELL), ELL - 1
        x = [a if i % 3 else b for i in range(n)]
    else:                           # everything lands in a handful of top-window buckets
        x = [(1 << 251) + (i % 3) for i in range(n)]
    pts = gpu_points(nat, ctx, exps)
    ds, out = ctx.upload(sc_bytes(nat, x)), ctx.alloc(64)
    want = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL))
    for c_bits in (0, 7, 12):
        ctx.set_window(c_bits)
        try:
            ctx.msm(ds.ptr, pts.ptr, n, None, None, 0, None, out.ptr)
            ctx.sync()
        finally:
            ctx.set_window(0)
        assert dl_aff(ctx, out.ptr)[0][:2] == want, (dist, c_bits)


def test_msm_edge_cases(nat, ctx):
    rng = random.Random(9)
    exps, g = make_points(rng, 8)
    out = ctx.alloc(64)
    dp = ctx.upload(aff_bytes(g))
    # empty product = identity (pivot.list_mul's initial value)
    ctx.msm(None, None, 0, None, None, 0, None, out.ptr)
    ctx.sync()
    assert dl_aff(ctx, out.ptr)[0][:2] == (0, 1)
    # all-zero scalars = identity
    ds = ctx.upload(sc_bytes(nat, [0] * 8))
    ctx.msm(ds.ptr, dp.ptr, 8, None, None, 0, None, out.ptr)
    ctx.sync()
    assert dl_aff(ctx, out.ptr)[0][:2] == (0, 1)
    # P + (-P): scalars 1 and l-1 on the same point
    dp2 = ctx.upload(aff_bytes([g[0], g[0]]))
    ds2 = ctx.upload(sc_bytes(nat, [1, ELL - 1]))
    ctx.msm(ds2.ptr, dp2.ptr, 2, None, None, 0, None, out.ptr)
    ctx.sync()
    assert dl_aff(ctx, out.ptr)[0][:2] == (0, 1)
    # identity point among the generators
    dp3 = ctx.upload(aff_bytes([ed.IDENTITY, g[1]]))
    ds3 = ctx.upload(sc_bytes(nat, [5, 7]))
    ctx.msm(ds3.ptr, dp3.ptr, 2, None, None, 0, None, out.ptr)
    ctx.sync()
    assert dl_aff(ctx, out.ptr)[0][:2] == ed.pt_affine(ed.pt_repeat(g[1], 7))
    # non-canonical scalar is reported at the sync point
    bad = np.frombuffer(ELL.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32)
    dsb = ctx.upload(bad)
    ctx.msm(dsb.ptr, dp.ptr, 1, None, None, 0, None, out.ptr)
    with pytest.raises(nat.VmpcError) as ei:
        ctx.sync()
    assert ei.value.code == nat.E_NONCANON
    ctx.sync()  # status word was cleared


@pytest.mark.parametrize("n,n_extra,rows", [(1, 0, 16), (5, 1, 1), (64, 2, 2), (301, 2, 16), (301, 2, 4), (70, 1, 8)])
def test_fixed_base_table_matches_oracle(nat, ctx, n, n_extra, rows):
    """vmpc_msm_table_dev over (g, extras) == pivot.vector_commitment restated, for the whole
    vector, a prefix, the empty prefix, and with / without the extra (h ** gamma) terms."""
    rng = random.Random(700 + n)
    _, pts = make_points(rng, n + n_extra)
    g, extras = pts[:n], pts[n:]
    x = [rng.randrange(ELL) for _ in range(n)]
    for i, v in enumerate([0, 1, ELL - 1, 2, ELL // 2, ELL // 2 + 1, (1 << 252) + 5]):
        if i < n:
            x[i] = v
    gam = [rng.randrange(ELL) for _ in range(n_extra)]
    dp = ctx.upload(aff_bytes(g))
    de = ctx.upload(aff_bytes(extras)) if n_extra else None
    table = ctx.msm_table_build(dp.ptr, n, de.ptr if de else None, n_extra, rows)
    ds = ctx.upload(sc_bytes(nat, x))
    dg = ctx.upload(sc_bytes(nat, gam)) if n_extra else None
    out, out_ext = ctx.alloc(64), ctx.alloc(128)
    for m in sorted({n, n // 2, 1, 0}):
        for use_extra in ([True, False] if n_extra else [False]):
            acc = ed.IDENTITY
            for xi, gi in zip(x[:m], g[:m]):
                acc = ed.pt_add(acc, ed.pt_repeat(gi, xi))
            if use_extra:
                for si, ei in zip(gam, extras):
                    acc = ed.pt_add(acc, ed.pt_repeat(ei, si))
            ctx.msm_table(table.ptr, n, n_extra, ds.ptr, m, dg.ptr if use_extra else None, out_ext.ptr, out.ptr, rows)
            ctx.sync()
            assert dl_aff(ctx, out.ptr)[0][:2] == ed.pt_affine(acc), (m, use_extra)
            raw = ctx.download(out_ext.ptr, 128).tobytes()
            X, Y, Z, T = (int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(4))
            assert ed.pt_affine((X, Y, Z)) == ed.pt_affine(acc) and (X * Y - T * Z) % P == 0
    # a non-canonical scalar is reported at the sync point, as for vmpc_msm_dev
    bad = np.frombuffer(ELL.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32)
    dsb = ctx.upload(bad)
    ctx.msm_table(table.ptr, n, n_extra, dsb.ptr, 1, None, None, out.ptr, rows)
    with pytest.raises(nat.VmpcError) as ei:
        ctx.sync()
    assert ei.value.code == nat.E_NONCANON
    ctx.sync()


def test_fixed_base_table_skewed_scalars(nat, ctx):
    """all terms in one bucket / a handful of buckets: the split + finish path of the single
    shared bucket set"""
    rng = random.Random(77)
    n = 5000
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    pts = gpu_points(nat, ctx, exps)
    out = ctx.alloc(64)
    for rows in (16, 2):
        table = ctx.msm_table_build(pts.ptr, n, None, 0, rows)
        for x in ([1] * n, [(1 << 16) + 1] * n, [ELL - 1] * n, [i % 3 for i in range(n)]):
            ds = ctx.upload(sc_bytes(nat, x))
            ctx.msm_table(table.ptr, n, 0, ds.ptr, n, None, None, out.ptr, rows)
            ctx.sync()
            want = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL))
            assert dl_aff(ctx, out.ptr)[0][:2] == want, rows


def test_repeat_replays_reference_sequence(nat, ctx):
    """`g ** n` gives the oracle's exact projective representative (Z included)."""
    rng = random.Random(11)
    _, pts = make_points(rng, 6)
    bases = [ed.BASE, ed.IDENTITY] + pts + [ed.pt_add(pts[0], pts[1]), ed.pt_dbl(pts[2])]
    scal = [0, 1, 2, 3, ELL - 1, ELL // 2, ELL // 2 + 1, 2**252, 2**64, rng.randrange(ELL)]
    scal = (scal * 2)[:len(bases)]
    db, ds = ctx.upload(proj_bytes(bases)), ctx.upload(sc_bytes(nat, scal))
    n = len(bases)
    op, oa = ctx.alloc(96 * n), ctx.alloc(64 * n)
    for signed in (False, True):
        ctx.repeat(db.ptr, n, False, ds.ptr, n, signed, op.ptr, oa.ptr)
        ctx.sync()
        want = [ed.pt_repeat(b, ed.scalar_int(s) if signed else s) for b, s in zip(bases, scal)]
        assert dl_proj(ctx, op.ptr, n) == want
        assert [a[:2] for a in dl_aff(ctx, oa.ptr, n)] == [ed.pt_affine(w) for w in want]
    # broadcast of one affine base: create_generators (circuit_sat_r1cs.py:64-70)
    exps = [rng.randrange(1, ELL) for _ in range(40)]
    dbase, dexp = ctx.upload(aff_bytes([ed.BASE])), ctx.upload(sc_bytes(nat, exps))
    op2 = ctx.alloc(96 * 40)
    ctx.repeat(dbase.ptr, 1, True, dexp.ptr, 40, False, op2.ptr, None)
    ctx.sync()
    assert dl_proj(ctx, op2.ptr, 40) == ac.create_generators(exps)["g"]


def test_fixed_base_comb_matches_oracle(nat, ctx):
    """vmpc_fixed_base_dev: out_i = n_i * B as affine points (circuit_sat_r1cs.py:64-70 when the
    projective representative is not needed), incl. digit-recoding edge cases."""
    rng = random.Random(31)
    base = ed.pt_repeat(ed.BASE, rng.randrange(1, ELL))
    sc = [0, 1, 2, 127, 128, 129, 255, 256, 257, 0x8080, 0x7f7f7f7f, (1 << 252) - 1, 1 << 252, ELL - 1, ELL - 2,
          int.from_bytes(bytes([0x80] * 31 + [0x0f]), "little"), int.from_bytes(bytes([0x81] * 31 + [0x0f]), "little"),
          int.from_bytes(bytes([0xff] * 31 + [0x0f]), "little")] + [rng.randrange(ELL) for _ in range(40)]
    db, ds = ctx.upload(aff_bytes([base])), ctx.upload(sc_bytes(nat, sc))
    out = ctx.alloc(64 * len(sc))
    ctx.fixed_base(db.ptr, ds.ptr, len(sc), out.ptr)
    ctx.sync()
    got = dl_aff(ctx, out.ptr, len(sc))
    for v, g in zip(sc, got):
        assert g[:2] == ed.pt_affine(ed.pt_repeat(base, v)), hex(v)
    bad = ctx.upload(np.frombuffer(ELL.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32))
    ctx.fixed_base(db.ptr, bad.ptr, 1, out.ptr)
    with pytest.raises(nat.VmpcError) as ei:
        ctx.sync()
    assert ei.value.code == nat.E_NONCANON
    ctx.sync()


def test_fold_replays_reference_sequence(nat, ctx):
    """g' = (g_l ** c) * g_r, compressed_pivot.py:64: exact (X, Y, Z) and affine."""
    rng = random.Random(12)
    half = 37
    gl = ac.create_generators([rng.randrange(1, ELL) for _ in range(half)])["g"]   # Z != 1
    gr = ac.create_generators([rng.randrange(1, ELL) for _ in range(half)])["g"]
    for c in (rng.randrange(ELL), 1, 0, ELL - 1):
        want = ac.fold_generators(gl, gr, c)
        dl_, dr_ = ctx.upload(proj_bytes(gl)), ctx.upload(proj_bytes(gr))
        op, oa = ctx.alloc(96 * half), ctx.alloc(64 * half)
        ctx.fold(dl_.ptr, dr_.ptr, False, c, half, op.ptr, oa.ptr)
        ctx.sync()
        assert dl_proj(ctx, op.ptr, half) == want
        assert [a[:2] for a in dl_aff(ctx, oa.ptr, half)] == [ed.pt_affine(w) for w in want]
    # affine inputs (Z = 1)
    gla, gra = [ed.pt_normalize(p) for p in gl], [ed.pt_normalize(p) for p in gr]
    c = rng.randrange(ELL)
    da, dbb = ctx.upload(aff_bytes(gla)), ctx.upload(aff_bytes(gra))
    op = ctx.alloc(96 * half)
    ctx.fold(da.ptr, dbb.ptr, True, c, half, op.ptr, None)
    ctx.sync()
    assert dl_proj(ctx, op.ptr, half) == ac.fold_generators(gla, gra, c)


@pytest.mark.parametrize("half", [1, 15, 16, 17, 50])
def test_fold_two_wave_ladder_bit_lengths(nat, ctx, half):
    """the short-vector fold (csrc/exact.hip k_fold_pipe: the doubling chain and the additions on two waves, batches of
    8 bits through LDS): scalars whose bit lengths sit on and around the batch boundaries, all-ones and single-bit
    scalars, against the oracle's replay of compressed_pivot.py:64 - exact (X, Y, Z)"""
    rng = random.Random(1200 + half)
    gl = ac.create_generators([rng.randrange(1, ELL) for _ in range(half)])["g"]
    gr = ac.create_generators([rng.randrange(1, ELL) for _ in range(half)])["g"]
    dl_, dr_ = ctx.upload(proj_bytes(gl)), ctx.upload(proj_bytes(gr))
    cs = [0, 1, 2, 3, 0x7f, 0x80, 0xff, 0x100, 0x101, 0xffff, 0x10000, (1 << 64) - 1, 1 << 64, 1 << 251, 1 << 252,
          (1 << 252) - 1, ELL - 1, ELL - 2, rng.randrange(ELL), rng.randrange(1 << 129)]
    for c in cs:
        op, oa = ctx.alloc(96 * half), ctx.alloc(64 * half)
        ctx.fold(dl_.ptr, dr_.ptr, False, c, half, op.ptr, oa.ptr)
        ctx.sync()
        want = ac.fold_generators(gl, gr, c)
        assert dl_proj(ctx, op.ptr, half) == want, hex(c)
        assert [a[:2] for a in dl_aff(ctx, oa.ptr, half)] == [ed.pt_affine(w) for w in want], hex(c)


def test_fold_every_kernel_agrees_with_the_oracle(nat, ctx):
    """one vector length per fold kernel (two-wave ladder <= 2^13 < quad <= 2^14 < one lane per element) against the
    threaded C oracle's replay (oracle/ed25519_oracle.c oracle_fold), exact (X, Y, Z)"""
    import numpy as np
    from oracle import c_oracle
    rng = np.random.default_rng(77)
    c_oracle.set_threads(c_oracle.host_threads())
    try:
        for half in (8192, 8193, 16384, 16400):
            ex = rng.integers(0, 256, size=(2 * half, 32), dtype=np.uint8)
            ex[:, 31] &= 0x0F
            base = np.frombuffer(proj_bytes([ed.BASE]), np.uint8)
            proj, _ = c_oracle.fixed_base(base, ex)
            pts = c_oracle.PointArray(proj)
            c = int.from_bytes(rng.integers(0, 256, size=32, dtype=np.uint8).tobytes(), "little") % ELL
            want = pts[:half].fold(pts[half:], c)
            d = ctx.upload(proj.tobytes())
            op = ctx.alloc(96 * half)
            ctx.fold(d.ptr, d.ptr + 96 * half, False, c, half, op.ptr, None)
            ctx.sync()
            assert ctx.download(op.ptr, 96 * half).tobytes() == want.a.tobytes(), half
    finally:
        c_oracle.set_threads(1)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 13, 64, 65])
def test_tree_reduce_order(nat, ctx, n):
    """pivot.list_mul (pivot.py:26-28): same tree shape => same projective representative."""
    rng = random.Random(13 + n)
    pts = ac.create_generators([rng.randrange(1, ELL) for _ in range(n)])["g"]
    for append in (True, False):
        d = ctx.upload(proj_bytes(pts))
        out = ctx.alloc(96)
        ctx.tree_reduce(d.ptr, n, append, out.ptr)
        ctx.sync()
        want = ed.tree_reduce(ed.pt_add, pts, ed.IDENTITY if append else None)
        assert dl_proj(ctx, out.ptr)[0] == want


def test_normalize_and_lift(nat, ctx):
    rng = random.Random(14)
    pts = ac.create_generators([rng.randrange(1, ELL) for _ in range(20)])["g"] + [ed.IDENTITY]
    n = len(pts)
    d = ctx.upload(proj_bytes(pts))
    oa, op = ctx.alloc(64 * n), ctx.alloc(96 * n)
    ctx.normalize(d.ptr, n, oa.ptr)
    ctx.affine_to_proj(oa.ptr, n, op.ptr)
    ctx.sync()
    assert [a[:2] for a in dl_aff(ctx, oa.ptr, n)] == [ed.pt_affine(p) for p in pts]
    assert dl_proj(ctx, op.ptr, n) == [ed.pt_normalize(p) for p in pts]
    assert ctx.validate_points(oa.ptr, n) == 0
    # off-curve and non-canonical encodings are counted
    bad = aff_bytes(pts[:3]).copy()
    bad[0, 0] ^= 1
    bad[1, 32:64] = np.frombuffer((P + 1).to_bytes(32, "little"), dtype=np.uint8)
    assert ctx.validate_points(ctx.upload(bad).ptr, 3) == 2


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 5000])
def test_fr_vector_ops(nat, ctx, n):
    rng = random.Random(15 + n)
    x = [rng.randrange(ELL) for _ in range(n)]
    y = [rng.randrange(ELL) for _ in range(n)]
    x[0], y[0] = ELL - 1, ELL - 1
    c = rng.randrange(ELL)
    dx, dy = ctx.upload(sc_bytes(nat, x)), ctx.upload(sc_bytes(nat, y))
    out = ctx.alloc(32 * n)
    ctx.fr_axpy(c, dx.ptr, dy.ptr, n, out.ptr)
    assert nat.array_to_ints(ctx.download(out.ptr, 32 * n, (n, 32))) == [(c * a + b) % ELL for a, b in zip(x, y)]
    ctx.fr_scale(c, dx.ptr, n, out.ptr)
    assert nat.array_to_ints(ctx.download(out.ptr, 32 * n, (n, 32))) == [(c * a) % ELL for a in x]
    assert ctx.fr_dot(dx.ptr, dy.ptr, n) == sum(a * b for a, b in zip(x, y)) % ELL
    # in-place axpy (z' overwrites z_l)
    ctx.fr_axpy(c, dy.ptr, dx.ptr, n, dx.ptr)
    assert nat.array_to_ints(ctx.download(dx.ptr, 32 * n, (n, 32))) == [(c * b + a) % ELL for a, b in zip(x, y)]


def test_tail_scalars_incremental_equals_direct(nat, ctx):
    """k_fr_tail_scalars (all pending challenges at once) and its round-by-round form
    (products carried in device memory) give the same A_i / B_i scalars, and both match the
    definition: s[j] = prod_{r<t} (c_r if bit (log2_m0-1-r) of j is 0)."""
    rng = random.Random(55)
    log2_m0 = 6
    m0 = 1 << log2_m0
    prod = ctx.alloc(32 * m0)
    cs = []
    for t in range(log2_m0 - 1):
        m = m0 >> t
        h = m // 2
        z = [rng.randrange(ELL) for _ in range(m)]
        dz = ctx.upload(sc_bytes(nat, z))
        a1, b1, a2, b2 = (ctx.alloc(32 * m0) for _ in range(4))
        ctx.fr_tail_scalars(cs, log2_m0, dz.ptr, a1.ptr, b1.ptr)
        ctx.fr_tail_scalars_inc(cs[-1] if cs else 0, t, log2_m0, dz.ptr, prod.ptr, a2.ptr, b2.ptr)
        ctx.sync()
        A1, B1 = (nat.array_to_ints(ctx.download(x.ptr, 32 * m0, (m0, 32))) for x in (a1, b1))
        A2, B2 = (nat.array_to_ints(ctx.download(x.ptr, 32 * m0, (m0, 32))) for x in (a2, b2))
        assert A1 == A2 and B1 == B2, t
        for j in range(m0):
            s = 1
            for r, c in enumerate(cs):
                if ((j >> (log2_m0 - 1 - r)) & 1) == 0:
                    s = s * c % ELL
            u = j & (m - 1)
            wa = z[u - h] * s % ELL if u >= h else 0
            wb = z[u + h] * s % ELL if u < h else 0
            assert (A1[j], B1[j]) == (wa, wb), (t, j)
        cs.append(rng.randrange(ELL))


def test_transcript_text(nat, ctx):
    """device-formatted text == the oracle's str() restatement (pivot.py:134)."""
    rng = random.Random(16)
    pts = ac.create_generators([rng.randrange(1, ELL) for _ in range(300)])["g"]
    pts += [ed.IDENTITY, ed.BASE, (0, 5, 7), (10**76, 1, 10**9)]
    txt = ctx.format_points(ctx.upload(proj_bytes(pts)).ptr, len(pts)).tobytes().decode()
    assert txt == "".join(ed.pt_repr(p) + ", " for p in pts)
    sc = [rng.randrange(ELL) for _ in range(2500)] + [0, 1, ELL - 1, ELL // 2, ELL // 2 + 1, 10**9, 10**9 - 1]
    d = ctx.upload(sc_bytes(nat, sc))
    txt = ctx.format_scalars(d.ptr, len(sc), True).tobytes().decode()
    assert txt == "".join(ed.scalar_repr(v) + ", " for v in sc)
    txt = ctx.format_scalars(d.ptr, len(sc), False).tobytes().decode()
    assert txt == "".join(str(v) + ", " for v in sc)


def test_host_one_shots(nat):
    rng = random.Random(17)
    exps, g = make_points(rng, 24)
    x = [rng.randrange(ELL) for _ in range(24)]
    out = nat.ed25519_msm(sc_bytes(nat, x), aff_bytes(g))
    want = ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL)
    assert ed.affine_from_bytes(out.tobytes())[:2] == ed.pt_affine(want)
    c = rng.randrange(ELL)
    got = nat.ed25519_fold(aff_bytes(g[:12]), aff_bytes(g[12:]), c)
    want = [ed.pt_affine(p) for p in ac.fold_generators(g[:12], g[12:], c)]
    assert [ed.affine_from_bytes(got[i].tobytes())[:2] for i in range(12)] == want
    got = nat.ed25519_fixed_base_batch(aff_bytes([ed.BASE]), sc_bytes(nat, exps))
    assert [ed.affine_from_bytes(got[i].tobytes())[:2] for i in range(24)] == [ed.pt_affine(p) for p in g]
    y = [rng.randrange(ELL) for _ in range(24)]
    assert nat.array_to_ints(nat.fr_axpy(c, sc_bytes(nat, x), sc_bytes(nat, y))) == \
        [(c * a + b) % ELL for a, b in zip(x, y)]
    assert nat.fr_dot(sc_bytes(nat, x), sc_bytes(nat, y)) == sum(a * b for a, b in zip(x, y)) % ELL
    # error conventions of the boundary
    offcurve = aff_bytes(g).copy()
    offcurve[3, 5] ^= 0x40
    with pytest.raises(nat.VmpcError) as ei:
        nat.ed25519_msm(sc_bytes(nat, x), offcurve)
    assert ei.value.code == nat.E_NOTONCURVE
    bad_sc = sc_bytes(nat, x).copy()
    bad_sc[0] = np.frombuffer((ELL + 5).to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(nat.VmpcError) as ei:
        nat.ed25519_msm(bad_sc, aff_bytes(g))
    assert ei.value.code == nat.E_NONCANON


@pytest.mark.parametrize("nbytes", [1, 55, 56, 63, 64, 65, 4095, 4096, 4097, 3 * 4096 + 100, 70001])
def test_sha256_chunks(nat, ctx, nbytes):
    import hashlib
    rng = np.random.default_rng(nbytes)
    data = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d = ctx.upload(data)
    for chunk in (4096, 64, 1000):
        got = ctx.sha256_chunks(d.ptr, nbytes, chunk)
        raw = data.tobytes()
        want = b"".join(hashlib.sha256(raw[o:o + chunk]).digest() for o in range(0, nbytes, chunk))
        assert got == want, (nbytes, chunk)


@pytest.fixture()
def table_window(ctx, request):
    ctx.set_window(request.param)
    yield request.param
    ctx.set_window(0)


@pytest.mark.parametrize("table_window", [0, 4, 8, 16], indirect=True)
@pytest.mark.parametrize("rows", [1, 4, 16])
def test_table_batch_equals_single_commitments(nat, ctx, rows, table_window):
    """vmpc_msm_table_batch_dev: K commitments over one table in one pass are the K single results - also
    with different scalar distributions per commitment (uniform, sparse, all-equal), extras on some only;
    for every digit width the table path takes (0 = its own choice, 16 bits)."""
    rng = random.Random(700 + rows)
    n, K = 700, 5
    _, pts = make_points(rng, n + 2)
    g, extras = pts[:n], pts[n:]
    dp, de = ctx.upload(aff_bytes(g)), ctx.upload(aff_bytes(extras))
    table = ctx.msm_table_build(dp.ptr, n, de.ptr, 2, rows)
    vecs = [[rng.randrange(ELL) for _ in range(n)],
            [rng.randrange(ELL) if i % 2 else 0 for i in range(n)],        # half-populated (A_i / B_i shape)
            [1] * n,
            [0] * n,
            [ELL - 1 - i for i in range(n)]]
    exs = [[rng.randrange(ELL), 0], None, [0, 5], None, [ELL - 1, ELL - 2]]
    dv = [ctx.upload(sc_bytes(nat, v)) for v in vecs]
    dx = [ctx.upload(sc_bytes(nat, e)) if e is not None else None for e in exs]
    single = []
    for v, e in zip(dv, dx):
        out = ctx.alloc(64)
        ctx.msm_table(table.ptr, n, 2, v.ptr, n, e.ptr if e is not None else None, None, out.ptr, rows=rows)
        single.append(dl_aff(ctx, out.ptr)[0])
    outs, oute = ctx.alloc(64 * K), ctx.alloc(128 * K)
    ctx.msm_table_batch(table.ptr, n, 2, [v.ptr for v in dv], n, [e.ptr if e is not None else None for e in dx],
                        oute.ptr, outs.ptr, rows=rows)
    assert dl_aff(ctx, outs.ptr, K) == single
    raw = ctx.download(oute.ptr, 128 * K).tobytes()
    for k in range(K):
        X, Y, Z = (int.from_bytes(raw[128 * k + 32 * i:128 * k + 32 * i + 32], "little") for i in range(3))
        assert ed.pt_affine((X, Y, Z)) == single[k][:2]
    # and against the oracle for one of them
    want = ed.pt_affine(ac.vector_commitment(vecs[0] + exs[0], 0, g + extras, ed.IDENTITY, signed_exponents=False))
    assert single[0][:2] == want
    # fewer terms than the table holds
    ctx.msm_table_batch(table.ptr, n, 2, [dv[0].ptr, dv[4].ptr], 123, None, None, outs.ptr, rows=rows)
    got = dl_aff(ctx, outs.ptr, 2)
    for k, v in zip(range(2), (vecs[0], vecs[4])):
        assert got[k][:2] == ed.pt_affine(ac.vector_commitment(v[:123], 0, g[:123], ed.IDENTITY, signed_exponents=False))


@pytest.mark.parametrize("rows,n_main,n_extra,k", [(4, 63, 2, 3), (16, 31, 1, 5), (1, 64, 0, 1), (2, 255, 3, 6),
                                                   (8, 16, 0, 4), (4, 300, 5, 2)])
def test_table_fold_equals_k_reference_folds(nat, ctx, rows, n_main, n_extra, k):
    """vmpc_msm_table_fold_dev: k rounds of g' = g_l^c * g_r (compressed_pivot.py:64, oracle fold_generators
    applied k times with independent challenges) in one pass over the unfolded table; the columns folded are
    the generators followed by the first extras (g_hat = g || h), as many as the largest power of two."""
    rng = random.Random(rows * 1000 + n_main)
    _, pts = make_points(rng, n_main + n_extra)
    dp = ctx.upload(aff_bytes(pts[:n_main]))
    de = ctx.upload(aff_bytes(pts[n_main:])) if n_extra else None
    table = ctx.msm_table_build(dp.ptr, n_main, de.ptr if de else None, n_extra, rows)
    n_cols = 1 << ((n_main + n_extra).bit_length() - 1)
    for trial in range(2):
        cs = [rng.randrange(ELL) for _ in range(k)] if trial == 0 else [0, 1, ELL - 1, 2, ELL - 2, 3][:k]
        g = list(pts[:n_cols])
        for c in cs:
            half = len(g) // 2
            g = ac.fold_generators(g[:half], g[half:], c)
        # s_b = prod over rounds i (first round = top bit of b) of c_i where that bit is 0
        s = []
        for b in range(1 << k):
            v = 1
            for i, c in enumerate(cs):
                if not (b >> (k - 1 - i)) & 1:
                    v = v * c % ELL
            s.append(v)
        out = ctx.alloc(64 * (n_cols >> k))
        ctx.msm_table_fold(table.ptr, n_main, n_extra, rows, n_cols, s, out.ptr)
        ctx.sync()
        got = dl_aff(ctx, out.ptr, n_cols >> k)
        want = [ed.pt_affine(p) for p in g]
        # an output that is the neutral element normalises to (0, 1); Z = 0 never occurs on a complete curve
        assert [a[:2] for a in got] == want


@pytest.mark.parametrize("rows,n_main,n_extra,k,out_rows", [(4, 63, 2, 3, 16), (8, 127, 1, 5, 4), (16, 256, 0, 2, 1),
                                                            (1, 31, 1, 1, 8), (8, 1023, 2, 5, 16)])
def test_table_fold_table_equals_fold_then_build(nat, ctx, rows, n_main, n_extra, k, out_rows):
    """vmpc_msm_table_fold_table_dev (fold, Horner, row doublings and normalisation fused, a quad of lanes per
    output) leaves byte for byte the table vmpc_msm_table_build_dev makes from vmpc_msm_table_fold_dev's vector."""
    rng = random.Random(rows * 77 + n_main)
    _, pts = make_points(rng, n_main + n_extra + 2)
    dp = ctx.upload(aff_bytes(pts[:n_main]))
    de = ctx.upload(aff_bytes(pts[n_main:n_main + n_extra])) if n_extra else None
    table = ctx.msm_table_build(dp.ptr, n_main, de.ptr if de else None, n_extra, rows)
    n_cols = 1 << ((n_main + n_extra).bit_length() - 1)
    m_out = n_cols >> k
    s = [rng.randrange(ELL) for _ in range(1 << k)]
    new_extras = ctx.upload(aff_bytes(pts[-2:]))
    folded = ctx.alloc(64 * m_out)
    ctx.msm_table_fold(table.ptr, n_main, n_extra, rows, n_cols, s, folded.ptr)
    want = ctx.msm_table_build(folded.ptr, m_out, new_extras.ptr, 2, out_rows)
    got = ctx.msm_table_fold_table(table.ptr, n_main, n_extra, rows, n_cols, s, new_extras.ptr, 2, out_rows)
    ctx.sync()
    nbytes = out_rows * ((m_out + 2 + 7) // 8 * 8) * 128
    assert ctx.download(got.ptr, nbytes).tobytes() == ctx.download(want.ptr, nbytes).tobytes()


def test_table_fold_argument_checks(nat, ctx):
    rng = random.Random(5)
    _, pts = make_points(rng, 8)
    dp = ctx.upload(aff_bytes(pts))
    table = ctx.msm_table_build(dp.ptr, 8, None, 0, 4)
    out = ctx.alloc(64 * 8)
    with pytest.raises(nat.VmpcError):
        ctx.msm_table_fold(table.ptr, 8, 0, 4, 16, [1, 2], out.ptr)          # more columns than the table has
    with pytest.raises(nat.VmpcError):
        ctx.msm_table_fold(table.ptr, 8, 0, 4, 6, [1, 2], out.ptr)           # not a power of two
    with pytest.raises(nat.VmpcError):
        ctx.msm_table_fold(table.ptr, 8, 0, 3, 8, [1, 2], out.ptr)           # rows
    with pytest.raises(nat.VmpcError) as e:
        lib = ctx.lib
        import ctypes
        raw = ctypes.create_string_buffer(b"\xff" * 64, 64)                   # non-canonical scalars
        nat._check(lib.vmpc_msm_table_fold_dev(ctx.handle, ctypes.c_void_p(table.ptr), 8, 0, 4, 8, 1, raw,
                                               ctypes.c_void_p(out.ptr)), "fold")
    assert e.value.code == nat.E_NONCANON


def test_measurement_probes(nat, ctx):
    """the probes bench.py prices the kernels against: integer-ALU ceilings (Ed25519 7M mixed addition; BN-256
    Jacobian mixed addition over F_p and F_p^2) and the known-size gather the HBM counter is calibrated on"""
    ed_rate = ctx.madd_rate(50)
    g1, g2 = ctx.bn256_madd_rate(1, 20), ctx.bn256_madd_rate(2, 10)
    assert 5e9 < ed_rate < 1e11, ed_rate             # ~3e10 on an MI355X
    assert 1e9 < g1 < ed_rate and 1e8 < g2 < g1, (ed_rate, g1, g2)   # Jacobian 11-multiplication additions, generic prime
    lines = (8 << 20) >> 7
    table = ctx.alloc(lines * 128)
    for mode in (0, 1, 2):
        ms = ctx.gather_probe(table.ptr, lines, 1 << 20, mode, seed=3)
        assert 0 < ms < 100, (mode, ms)
    with pytest.raises(nat.VmpcError):
        ctx.gather_probe(table.ptr, lines, 1 << 20, 3)
    with pytest.raises(nat.VmpcError):
        ctx.bn256_madd_rate(3, 10)
