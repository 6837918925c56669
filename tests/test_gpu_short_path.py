"""The fused three-launch path for commitments over a 16-row table of at most 2^17 columns (csrc/msm_short.hip:
BASELINE config 2 and every compact-prover round after the fold jump), through the C-ABI: against the oracle's
restatement of pivot.vector_commitment (verifiable_mpc/ac20/pivot.py:139-145) at sizes it finishes in seconds, against
the exponent identity and the general pipeline above that; skewed scalars (whole-workgroup buckets), the capacity
overflow answer (VMPC_E_AGAIN -> the same call on the general path), non-canonical scalars."""
import random

import numpy as np
import pytest

from oracle import ed25519_ref as ed
from tests.test_gpu_cabi import aff_bytes, gpu_points, make_points, sc_bytes

pytestmark = pytest.mark.gpu
ELL, P = ed.ELL, ed.P


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


def ran_short(ctx, fn):
    """fn() with stage timing on: did the fused path run it?"""
    ctx.profile(True)
    ctx.profile_read(reset=True)
    fn()
    ctx.sync()
    stages = {k for k, (_, launches) in ctx.profile_read(reset=True).items() if launches}
    ctx.profile(False)
    return "short_bins" in stages and "msm_bucket" not in stages


@pytest.mark.parametrize("n,n_extra", [(1, 0), (5, 1), (300, 2), (1100, 2)])
def test_short_path_matches_oracle(nat, ctx, n, n_extra):
    rng = random.Random(900 + n)
    _, pts = make_points(rng, n + n_extra)
    g, extras = pts[:n], pts[n:]
    x = [[rng.randrange(ELL) for _ in range(n)] for _ in range(2)]
    for i, v in enumerate([0, 1, ELL - 1, 2, ELL // 2, ELL // 2 + 1, (1 << 252) + 5, 0x8000, 0x7fff, 0x18000]):
        if i < n:
            x[0][i] = v
    gam = [[rng.randrange(ELL) for _ in range(n_extra)] for _ in range(2)]
    dp = ctx.upload(aff_bytes(g))
    de = ctx.upload(aff_bytes(extras)) if n_extra else None
    table = ctx.msm_table_build(dp.ptr, n, de.ptr if de else None, n_extra, 16)
    ds = [ctx.upload(sc_bytes(nat, v)) for v in x]
    dg = [ctx.upload(sc_bytes(nat, v)) for v in gam] if n_extra else None
    out = ctx.alloc(256)

    def want(k, m, use_extra):
        acc = ed.IDENTITY
        for xi, gi in zip(x[k][:m], g[:m]):
            acc = ed.pt_add(acc, ed.pt_repeat(gi, xi))
        if use_extra:
            for si, ei in zip(gam[k], extras):
                acc = ed.pt_add(acc, ed.pt_repeat(ei, si))
        return ed.pt_affine(acc)
    for m in sorted({n, n // 2, 1, 0}):
        for use_extra in ([True, False] if n_extra else [False]):
            # one commitment (two workgroups per bin) ...
            assert ran_short(ctx, lambda: ctx.msm_table(table.ptr, n, n_extra, ds[0].ptr, m,
                                                         dg[0].ptr if use_extra else None, out.ptr, None, 16))
            assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == want(0, m, use_extra), (m, use_extra)
            # ... and a pair in one pass (the A_i, B_i of a prover round)
            assert ran_short(ctx, lambda: ctx.msm_table_batch(table.ptr, n, n_extra, [d.ptr for d in ds], m,
                                                               [d.ptr for d in dg] if use_extra else None,
                                                               out.ptr, None, 16))
            raw = ctx.download(out.ptr, 256).tobytes()
            assert [ext_affine(raw[:128]), ext_affine(raw[128:])] == [want(0, m, use_extra), want(1, m, use_extra)]


@pytest.mark.parametrize("lg", [12, 16, 17])
def test_short_path_exponent_identity_and_general_path(nat, ctx, lg):
    n = (1 << lg) - 3
    rng = random.Random(lg)
    exps = [rng.randrange(1, ELL) for _ in range(n + 2)]
    pts = gpu_points(nat, ctx, exps)
    table = ctx.msm_table_build(pts.ptr, n, pts.ptr + 64 * n, 2, 16)
    out = ctx.alloc(256)
    dists = {
        "uniform": [rng.randrange(ELL) for _ in range(n)],
        # the [z] distribution of circuit_sat_cb.py:91-103: 54 % zeros, 9 % in {1, 2}, the rest uniform
        "commitment": [0 if (r := rng.random()) < 0.54 else rng.randrange(1, 3) if r < 0.63 else rng.randrange(ELL)
                       for _ in range(n)],
        "bits": [rng.randrange(2) for _ in range(n)],
        "half_zero": [rng.randrange(ELL) if (i >> 5) & 1 else 0 for i in range(n)],
    }
    gam = [rng.randrange(ELL), rng.randrange(ELL)]
    dg = ctx.upload(sc_bytes(nat, gam))
    for name, x in dists.items():
        ds = ctx.upload(sc_bytes(nat, x))
        tot = (sum(a * b for a, b in zip(x, exps)) + gam[0] * exps[n] + gam[1] * exps[n + 1]) % ELL
        want = ed.pt_affine(ed.pt_repeat(ed.BASE, tot))

        def launch():
            ctx.msm_table(table.ptr, n, 2, ds.ptr, n, dg.ptr, out.ptr, None, 16)
        if name == "bits" and lg >= 16:
            # 2^15+ entries in ONE bucket: beyond the fixed capacities -> the answer is "repeat on the general path"
            launch()
            with pytest.raises(nat.VmpcError) as ei:
                ctx.sync()
            assert ei.value.code == nat.E_AGAIN
            ctx.sync()                                     # the status words were cleared
            ctx.on_general_path(lambda: (launch(), ctx.sync()))
            # after an overflow the context goes straight to the general path for a while (api.hip short_backoff) ...
            assert not ran_short(ctx, launch)
            ctx.set_short_path(True, forget_overflow=True)       # ... unless told to forget
        else:
            assert ran_short(ctx, launch), name
        assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == want, name
        ctx.set_short_path(False)
        assert not ran_short(ctx, launch)
        ctx.set_short_path(True)
        assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == want, name
    # back-to-back calls re-arm the bin cursors: the same call again, twice, and the pair
    ds = ctx.upload(sc_bytes(nat, dists["uniform"]))
    ds2 = ctx.upload(sc_bytes(nat, dists["half_zero"]))
    for _ in range(2):
        ctx.msm_table(table.ptr, n, 2, ds.ptr, n, None, out.ptr, None, 16)
    ctx.msm_table_batch(table.ptr, n, 2, [ds.ptr, ds2.ptr], n, None, out.ptr, None, 16)
    ctx.sync()
    raw = ctx.download(out.ptr, 256).tobytes()
    for x, r in ((dists["uniform"], raw[:128]), (dists["half_zero"], raw[128:])):
        assert ext_affine(r) == ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL))


@pytest.mark.parametrize("kind", ["three_values", "same_scalar", "same_small"])
def test_short_path_heavy_buckets(nat, ctx, kind):
    """several whole-workgroup buckets in one bin (scalars from {1, 2, 3}); every term the same scalar (one bucket per
    window takes everything: 6000 entries per workgroup, or - when two windows' digits share a bin - more than a bin
    holds, and then the answer must be VMPC_E_AGAIN, never a wrong point)"""
    rng = random.Random(len(kind))
    n = 12345 if kind != "three_values" else 6000
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    pts = gpu_points(nat, ctx, exps)
    table = ctx.msm_table_build(pts.ptr, n, None, 0, 16)
    out = ctx.alloc(128)
    for trial in range(4):
        if kind == "three_values":
            x = [rng.randrange(1, 4) for _ in range(n)]
        elif kind == "same_scalar":
            x = [rng.randrange(ELL)] * n
        else:
            x = [rng.randrange(1, 1 << 40)] * n
        ds = ctx.upload(sc_bytes(nat, x))
        want = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(x, exps)) % ELL))

        def launch():
            ctx.msm_table(table.ptr, n, 0, ds.ptr, n, None, out.ptr, None, 16)
            ctx.sync()
        try:
            launch()
        except nat.VmpcError as e:
            assert e.code == nat.E_AGAIN and kind != "three_values"
            ctx.on_general_path(launch)
        assert ext_affine(ctx.download(out.ptr, 128).tobytes()) == want, (kind, trial)


def test_short_path_reports_non_canonical_scalars(nat, ctx):
    rng = random.Random(5)
    exps = [rng.randrange(1, ELL) for _ in range(40)]
    pts = gpu_points(nat, ctx, exps)
    table = ctx.msm_table_build(pts.ptr, 40, None, 0, 16)
    bad = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in [3, ELL, 7] + [1] * 37), np.uint8).reshape(40, 32)
    out = ctx.alloc(128)
    ctx.msm_table(table.ptr, 40, 0, ctx.upload(bad).ptr, 40, None, out.ptr, None, 16)
    with pytest.raises(nat.VmpcError) as ei:
        ctx.sync()
    assert ei.value.code == nat.E_NONCANON
    ctx.sync()


def test_vector_commitment_repeats_an_overflowing_call_on_the_general_path():
    """the Python layer's answer to VMPC_E_AGAIN (pivot._PendingCommitment, vector_commitment_pair)"""
    import verifiable_mpc_amd as vm
    group = vm.EllipticCurve("Ed25519", "projective")
    rng = random.Random(8)
    n = 1 << 16
    exps = np.frombuffer(rng.randbytes(32 * n), np.uint8).reshape(n, 32).copy()
    exps[:, 31] &= 0x0f
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(exps), keep_proj=False)
    h = group.generator
    g.precompute([h])
    assert g._table.rows == 16
    ones = vm.ScalarVector.from_ints([1] * n)
    tot = sum(int.from_bytes(bytes(e), "little") for e in exps) % ELL
    want = vm.Ed25519Point.repeat(group.generator, (tot + 5) % ELL)
    assert vm.pivot.vector_commitment(ones, 5, g, h) == want
    a, b = vm.pivot.vector_commitment_pair(ones, 5, g, ones, 6, g, h)
    assert a == want and b == vm.Ed25519Point.repeat(group.generator, (tot + 6) % ELL)
    # two commitments pending on ONE context, the second one overflows: whoever synchronises first collects the
    # context's status word, but a void result says so itself (Z = 0) - both come out right
    uni = vm.ScalarVector.from_array(exps)
    ctx = g.ctx
    first = vm.pivot._commit_launch(uni, 7, g, h, ctx)
    second = vm.pivot._commit_launch(ones, 9, g, h, ctx)
    dot = sum(int.from_bytes(bytes(e), "little") ** 2 for e in exps) % ELL
    assert first.result() == vm.Ed25519Point.repeat(group.generator, (dot + 7) % ELL)
    assert second.result() == vm.Ed25519Point.repeat(group.generator, (tot + 9) % ELL)
    # the path is still switched on (a user's own setting is restored, not forced), but after an overflow the context
    # sends its next eligible commitments straight to the general path (scalars like these tend to come again) ...
    ctx = g.ctx
    assert ctx.get_short_path() is True

    def stages_of_one_commitment():
        ctx.profile(True)
        ctx.profile_read(reset=True)
        assert vm.pivot.vector_commitment(uni, 7, g, h) == vm.Ed25519Point.repeat(group.generator, (dot + 7) % ELL)
        st = {k for k, (_, launches) in ctx.profile_read(reset=True).items() if launches}
        ctx.profile(False)
        return st
    assert "short_bins" not in stages_of_one_commitment()
    # ... until the back-off is over or cleared
    ctx.set_short_path(True, forget_overflow=True)
    assert "short_bins" in stages_of_one_commitment()
    # a caller's "off" survives on_general_path
    ctx.set_short_path(False)
    ctx.on_general_path(lambda: None)
    assert ctx.get_short_path() is False
    ctx.set_short_path(True, forget_overflow=True)
