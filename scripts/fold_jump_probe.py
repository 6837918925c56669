#!/usr/bin/env python3
"""time vmpc_msm_table_fold_dev at n = 2^20 (k = 1..6, table rows 4) - development probe"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a

ctx = vm.get_context(); rng = np.random.default_rng(3)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << logn
group = vm.EllipticCurve("Ed25519", "projective")
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)))
g.precompute([group.generator])
t = g._table
print("rows", t.rows, flush=True)
for k in (1, 2, 3, 4, 5, 6):
    s = [int.from_bytes(rs(rng, 1)[0].tobytes(), "little") % vm.groups.ORDER for i in range(1 << k)]
    out = ctx.alloc(64 * (n >> k))
    ts = []
    for _ in range(4):
        ctx.sync(); t0 = time.perf_counter()
        ctx.msm_table_fold(t.ptr, t.n, len(t.extra_bytes), t.rows, n, s, out.ptr)
        ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(k, [round(x, 3) for x in ts], flush=True)
