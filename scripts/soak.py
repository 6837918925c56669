#!/usr/bin/env python3
"""Randomised parity soak (developer tool, GPU): many sizes / seeds / scalar distributions through the C-ABI,
checked by the exponent identity  sum_i s_i (e_i B) == (sum_i s_i e_i mod order) B  - the generators are made
from known exponents, so every commitment has a closed form - for
  * Ed25519: variable-base MSM, table MSM over 1/2/4/8/16 rows, batches of 2-3 commitments;
  * BN-256 G1 / G2: variable-base and prepared-key (table) sums.
    python3 scripts/soak.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
ELL = vm.groups.ORDER
BN_ORDER = 65000549695646603732796438742359905742570406053903786389881062969044166799969
P_BN = 65000549695646603732796438742359905742825358107623003571877145026864184071783
G1 = (1).to_bytes(32, "little") + (P_BN - 2).to_bytes(32, "little")
G2 = b"".join(v.to_bytes(32, "little") for v in (
    64746500191241794695844075326670126197795977525365406531717464316923369116492,
    21167961636542580255011770066570541300993051739349375019639421053990175267184,
    17778617556404439934652658462602675281523610326338642107814333856843981424549,
    20666913350058776956210519119118544732556678129809273996262322366050359951122))


def scalars(n, order_bits, kind):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F if order_bits == 253 else 0x7F
    if kind == "skew":          # mostly 0 / 1 / -1-like / small: the witness distribution of the demo circuit
        pick = rng.random(n)
        a[pick < 0.5] = 0
        ones = (pick >= 0.5) & (pick < 0.65)
        a[ones] = 0
        a[ones, 0] = 1
        small = (pick >= 0.65) & (pick < 0.75)
        a[small, 2:] = 0
    elif kind == "same":        # every term the same scalar: one bucket per window takes everything
        a[:] = a[0]
    return a


def as_int(row):
    return int.from_bytes(bytes(row), "little")


def dot(sc, ex, order):
    return sum(as_int(x) * as_int(y) for x, y in zip(sc, ex)) % order


t_end = time.time() + budget
counts = {"ed_var": 0, "ed_table": 0, "ed_batch": 0, "bn_g1": 0, "bn_g2": 0}
it = 0
while time.time() < t_end:
    it += 1
    n = int(rng.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 257, 1000, 4095, 4097, 8193, 12345, 20000, 1 << 15])) \
        if it % 3 else int(rng.integers(1, 1 << 14))
    kind = ["uniform", "skew", "same"][int(rng.integers(0, 3))]
    # ---- Ed25519
    ex = scalars(n, 253, "uniform")
    ex[:, 0] |= 1
    pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(ex), keep_proj=False)
    sc = scalars(n, 253, kind)
    want = vm.PointVector.fixed_base(group.generator, [dot(sc, ex, ELL)], keep_proj=False)[0]
    out = ctx.alloc(128)
    sv = vm.ScalarVector.from_array(sc)
    ctx.msm(sv.ptr, pts.affine_ptr, n, None, None, 0, out.ptr, None)
    got = vm.Ed25519Point.from_proj_bytes(ctx.download(out.ptr, 128).tobytes()[:96])
    assert got == want, ("ed var", n, kind, seed, it)
    counts["ed_var"] += 1
    rows = int(rng.choice([1, 2, 4, 8, 16]))
    tab = vm.PointVector(pts.a, None, ctx).precompute([], rows=rows)
    t = tab._table
    def table_commitment():
        ctx.msm_table(t.ptr, t.n, 0, sv.ptr, n, None, out.ptr, None, rows=rows)
        ctx.sync()
    try:
        table_commitment()
    except vm._native.VmpcError as e:       # the fused short path's answer to scalars beyond its capacities ("same")
        assert e.code == vm._native.E_AGAIN and rows == 16 and kind != "uniform", (n, rows, kind, seed, it)
        counts["ed_table_repeated_on_general_path"] = counts.get("ed_table_repeated_on_general_path", 0) + 1
        ctx.on_general_path(table_commitment)
    got = vm.Ed25519Point.from_proj_bytes(ctx.download(out.ptr, 128).tobytes()[:96])
    assert got == want, ("ed table", n, rows, kind, seed, it)
    counts["ed_table"] += 1
    if n >= 8:
        K = int(rng.integers(2, 4))
        scs = [scalars(n, 253, ["uniform", "skew"][int(rng.integers(0, 2))]) for _ in range(K)]
        svs = [vm.ScalarVector.from_array(s) for s in scs]
        outk = ctx.alloc(128 * K)
        ctx.msm_table_batch(t.ptr, t.n, 0, [s.ptr for s in svs], n, None, outk.ptr, None, rows=rows)
        raw = ctx.download(outk.ptr, 128 * K).tobytes()
        for k in range(K):
            wantk = vm.PointVector.fixed_base(group.generator, [dot(scs[k], ex, ELL)], keep_proj=False)[0]
            assert vm.Ed25519Point.from_proj_bytes(raw[128 * k:128 * k + 96]) == wantk, ("ed batch", n, rows, k, seed, it)
        counts["ed_batch"] += 1
    # ---- BN-256 (smaller sizes: the exponent dot product is host big-int work)
    if it % 2 == 0:
        nb = min(n, 6000)
        for grp, gen, width, key in ((1, G1, 64, "bn_g1"), (2, G2, 128, "bn_g2")):
            exb = scalars(nb, 256, "uniform")
            scb = scalars(nb, 256, kind)
            dg, de = ctx.upload(np.frombuffer(gen, np.uint8)), ctx.upload(exb)
            dp, ds, res, wantb = ctx.alloc(width * nb), ctx.upload(scb), ctx.alloc(width), ctx.alloc(width)
            ctx.bn256_fixed_base(grp, dg.ptr, de.ptr, nb, dp.ptr)
            tot = dot(scb, exb, BN_ORDER)
            dt_ = ctx.upload(np.frombuffer(tot.to_bytes(32, "little"), np.uint8))
            ctx.bn256_fixed_base(grp, dg.ptr, dt_.ptr, 1, wantb.ptr)
            ctx.bn256_msm(grp, ds.ptr, dp.ptr, nb, res.ptr)
            w = ctx.download(wantb.ptr, width).tobytes()
            assert ctx.download(res.ptr, width).tobytes() == w, (key, "var", nb, kind, seed, it)
            tb = ctx.bn256_table_build(grp, dp.ptr, nb)
            ctx.bn256_table_msm(grp, tb.ptr, nb, ds.ptr, nb, res.ptr, None)
            assert ctx.download(res.ptr, width).tobytes() == w, (key, "table", nb, kind, seed, it)
            counts[key] += 1
print("soak ok:", counts, f"{it} iterations, seed {seed}")
