#!/usr/bin/env python3
"""Developer probe: stage times of a big prover round's commitment pair (two vectors over the 8-row CRS table, each
zero on half of its positions) against ONE full vector with the same number of non-zero terms."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
rng = np.random.default_rng(5)


def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F
    return a


lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 1 << lg
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
pts.precompute([], rows=rows)
t_ = pts._table
a, b = rs(n), rs(n)
for split_bit in (lg - 1, lg - 3):
    half = (np.arange(n) >> split_bit) & 1
    a2, b2 = a.copy(), b.copy()
    a2[half == 0] = 0
    b2[half == 1] = 0
    sa, sa2, sb2 = (vm.ScalarVector.from_array(x) for x in (a, a2, b2))
    out = ctx.alloc(256)
    cases = {"one full vector": lambda: ctx.msm_table(t_.ptr, t_.n, 0, sa.ptr, n, None, out.ptr, None, rows=rows),
             f"pair, each zero on half (split at bit {split_bit})":
                 lambda: ctx.msm_table_batch(t_.ptr, t_.n, 0, [sa2.ptr, sb2.ptr], n, None, out.ptr, None, rows=rows)}
    for name, fn in cases.items():
        for fill in (0, 2):
            if "pair" not in name and fill:
                continue
            for _ in range(3):
                fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
                ctx.sync()
            dt = (time.perf_counter() - t0) / 10 * 1e3
            ctx.profile(True)
            ctx.profile_read(reset=True)
            for _ in range(5):
                fn()
            ctx.sync()
            st = {k: ms / c * 1e3 for k, (ms, c) in ctx.profile_read(reset=True).items() if c}
            ctx.profile(False)
            print(f"n=2^{lg} rows={rows} {name}: {dt:.3f} ms [" + " ".join(f"{k.replace('msm_', '')} {v:.0f}" for k, v in st.items()) + "]")
