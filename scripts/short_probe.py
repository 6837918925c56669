#!/usr/bin/env python3
"""Developer probe: one commitment / a pair over a 16-row table, fused short path against the general pipeline
(stage times with events on, latency without)."""
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


for lg in [int(a) for a in sys.argv[1:]] or [12, 15, 16, 17]:
    n = 1 << lg
    pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
    pts.precompute([], rows=16)
    t_ = pts._table
    a, b = rs(n), rs(n)
    half = (np.arange(n) >> max(lg - 3, 0)) & 1
    a2, b2 = a.copy(), b.copy()
    a2[half == 0] = 0
    b2[half == 1] = 0
    sa, sb, sa2, sb2 = (vm.ScalarVector.from_array(x) for x in (a, b, a2, b2))
    out = ctx.alloc(256)
    cases = {"one commitment": lambda: ctx.msm_table(t_.ptr, t_.n, 0, sa.ptr, n, None, out.ptr, None, rows=16),
             "pair": lambda: ctx.msm_table_batch(t_.ptr, t_.n, 0, [sa.ptr, sb.ptr], n, None, out.ptr, None, rows=16),
             "pair, each zero on half (a prover round)":
                 lambda: ctx.msm_table_batch(t_.ptr, t_.n, 0, [sa2.ptr, sb2.ptr], n, None, out.ptr, None, rows=16)}
    for name, fn in cases.items():
        line = f"n=2^{lg} {name}:"
        for short in (True, False):
            ctx.set_short_path(short)
            for _ in range(3):
                fn()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
                ctx.sync()
            dt = (time.perf_counter() - t0) / 20 * 1e3
            ctx.profile(True)
            ctx.profile_read(reset=True)
            for _ in range(5):
                fn()
            ctx.sync()
            st = {k: ms / c * 1e3 for k, (ms, c) in ctx.profile_read(reset=True).items() if c}
            ctx.profile(False)
            line += f"  {'short' if short else 'general'} {dt:.3f} ms [" + " ".join(f"{k.replace('msm_', '').replace('short_', '')} {v:.0f}" for k, v in st.items()) + "]"
        ctx.set_short_path(True)
        print(line)
