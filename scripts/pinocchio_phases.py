#!/usr/bin/env python3
"""Developer probe: where the time of a prepared-key Pinocchio proof goes (uploads, the multi-key pass, the twist sum,
h's sum, and the whole in two launch orders)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pynocchio as pn
from verifiable_mpc_amd.device import get_aux_context

ctx = vm.get_context()
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
key = pn.PreparedKey.synthetic(ctx, n)
rng = np.random.default_rng(3)
c = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); c[:, 31] &= 0x7F
h = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); h[:, 31] &= 0x7F


def t(fn, reps=5):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


dc, dh = ctx.upload(c), ctx.upload(h)
print(f"upload of one (n, 32) scalar array: {t(lambda: ctx.upload(c)):.3f} ms")
g1 = [key.vectors[name] for name in pn._SHARED_G1]
out = ctx.alloc(96 * 8)
print(f"six G1 sums, one pass: {t(lambda: ctx.bn256_table_msm_multi(1, [v.table.ptr for v in g1], g1[0].n, dc.ptr, n, out.ptr)):.3f} ms")
tw = key.vectors["r_w*w_mid*g2"]
o2 = ctx.alloc(192)
print(f"twist sum: {t(lambda: ctx.bn256_table_msm(2, tw.table.ptr, tw.n, dc.ptr, n, None, o2.ptr)):.3f} ms")
hv = key.vectors["h*g1"]
print(f"h sum: {t(lambda: ctx.bn256_table_msm(1, hv.table.ptr, hv.n, dh.ptr, n, None, out.ptr)):.3f} ms")
aux = get_aux_context(20)


def whole(twist_first, streams):
    a = aux if streams else ctx
    if streams:
        aux.wait_for(ctx)
    if twist_first:
        a.bn256_table_msm(2, tw.table.ptr, tw.n, dc.ptr, n, None, o2.ptr)
    ctx.bn256_table_msm_multi(1, [v.table.ptr for v in g1], g1[0].n, dc.ptr, n, out.ptr)
    if not twist_first:
        a.bn256_table_msm(2, tw.table.ptr, tw.n, dc.ptr, n, None, o2.ptr)
    ctx.bn256_table_msm(1, hv.table.ptr, hv.n, dh.ptr, n, None, out.ptr + 96 * 6)
    if streams:
        aux.sync()


for tf in (True, False):
    for st in (False, True):
        print(f"all eight sums, scalars resident, twist {'first' if tf else 'last'}, {'two streams' if st else 'one stream'}: "
              f"{t(lambda: whole(tf, st)):.3f} ms")


class D:
    v, w, y = 11, 22, 33
print(f"compute_proof (uploads + eight sums + host glue): {t(lambda: pn.compute_proof(None, c, h, key, D)):.3f} ms")
