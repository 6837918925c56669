#!/usr/bin/env python3
"""Developer probe: host-side cost of enqueueing one commitment (no waiting) and of fetching a result."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
ctx = vm.get_context()
rng = np.random.default_rng(1)
n = 1 << 20
a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0f
b = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); b[:, 31] &= 0x0f
pts = vm.PointVector.fixed_base(vm.Ed25519Point.generator, vm.ScalarVector.from_array(a), keep_proj=False)
pts.precompute([], rows=1)
sc = vm.ScalarVector.from_array(b)
out = ctx.alloc(128)
t = pts._table
for _ in range(3):
    ctx.msm_table(t.ptr, t.n, 0, sc.ptr, n, None, out.ptr, None, rows=1)
ctx.sync()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    ctx.msm_table(t.ptr, t.n, 0, sc.ptr, n, None, out.ptr, None, rows=1)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    raw = ctx.download(out.ptr, 96).tobytes()
    p = vm.Ed25519Point.from_proj_bytes(raw).normalize()
    t3 = time.perf_counter()
    ts.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
print("enqueue us, wait us, fetch+normalize us")
for r in ts: print("  %.0f  %.0f  %.0f" % r)
