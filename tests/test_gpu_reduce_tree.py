"""The two bucket reductions agree: the quad weight tree (csrc/msm_reduce_tree.hip, the default for chunks of up to
eight buckets) against the per-lane offset ladders (k_msm_reduce, forced with VMPC_REDUCE_TREE=0) on the same
commitments - variable-base and tabulated (1, 4, 16 rows), sizes that give one workgroup per bucket set (G = 1) up to
G = 128, chunks of 1, 2, 4 and 8 buckets, scalars that leave most buckets empty or fill a single one.  The results
are products of the reference's list_mul over the same lists (pivot.py:28-33, 143-144); the small cases are also
checked by the exponent identity against the oracle's scalar arithmetic."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)
L_ORDER = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def contexts():
    from verifiable_mpc_amd import _native as nat
    tree = nat.Context(0)
    saved = {k: os.environ.get(k) for k in ("VMPC_EXPERIMENTAL", "VMPC_REDUCE_TREE")}
    os.environ["VMPC_EXPERIMENTAL"] = "1"
    os.environ["VMPC_REDUCE_TREE"] = "0"
    try:
        ladder = nat.Context(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    yield tree, ladder
    tree.close()
    ladder.close()


def _scalars(rng, n, kind):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F
    if kind == "small":           # 20-bit scalars: every window but the lowest two is empty
        a[:, 3:] = 0
        a[:, 2] &= 0x0F
    elif kind == "equal":         # one bucket per window holds everything
        a[:] = a[0]
    elif kind == "top":           # l - 1: the largest canonical scalar in every position
        a[:] = np.frombuffer((L_ORDER - 1).to_bytes(32, "little"), dtype=np.uint8)
    return a


def _int(a):
    return [int.from_bytes(bytes(r), "little") for r in a]


@pytest.mark.parametrize("lg", [6, 9, 12, 14, 16, 18])
@pytest.mark.parametrize("kind", ["uniform", "small", "equal", "top"])
def test_tree_and_ladder_reductions_agree(contexts, lg, kind):
    tree, ladder = contexts
    rng = np.random.default_rng(1000 * lg + len(kind))
    n = (1 << lg) - 3
    exps = _scalars(rng, n, "uniform")
    sc = _scalars(rng, n, kind)
    base = np.frombuffer(BASE[0].to_bytes(32, "little") + BASE[1].to_bytes(32, "little"), dtype=np.uint8)
    outs = []
    for ctx in (tree, ladder):
        dbase, dexp, dsc = ctx.upload(base), ctx.upload(exps), ctx.upload(sc)
        dpts = ctx.alloc(64 * n)
        ctx.repeat(dbase.ptr, 1, True, dexp.ptr, n, False, None, dpts.ptr)
        got = []
        o = ctx.alloc(64)
        ctx.msm(dsc.ptr, dpts.ptr, n, None, None, 0, None, o.ptr)
        ctx.sync()
        got.append(ctx.download(o.ptr, 64).tobytes())
        for rows in (1, 4, 16):
            table = ctx.msm_table_build(dpts.ptr, n, None, 0, rows)
            ctx.msm_table(table.ptr, n, 0, dsc.ptr, n, None, None, o.ptr, rows)
            ctx.sync()
            got.append(ctx.download(o.ptr, 64).tobytes())
        if ctx is tree:
            # exponent identity: sum_i s_i (e_i B) = (sum_i s_i e_i mod l) B
            e = sum(s * x for s, x in zip(_int(sc), _int(exps))) % L_ORDER
            de = ctx.upload(np.frombuffer(e.to_bytes(32, "little"), dtype=np.uint8))
            ctx.repeat(dbase.ptr, 1, True, de.ptr, 1, False, None, o.ptr)
            ctx.sync()
            expect = ctx.download(o.ptr, 64).tobytes()
        outs.append(got)
    assert outs[0] == outs[1]
    assert all(g == expect for g in outs[0])
