#!/usr/bin/env python3
"""Developer probe: fold kernel time vs size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
ctx = vm.get_context(); rng = np.random.default_rng(1)
def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
G = vm.Ed25519Point.generator
for lg in [int(a) for a in sys.argv[1:]] or [10, 12, 13, 14, 15, 16, 17]:
    n = 1 << lg
    g = vm.PointVector.fixed_base(G, vm.ScalarVector.from_array(rs(2 * n)), keep_proj=False)
    c = int.from_bytes(rs(1)[0].tobytes(), "little")
    for rep in range(2):
        ctx.sync(); t0 = time.perf_counter()
        o = g[:n].fold(g[n:], c); ctx.sync(); dt = time.perf_counter() - t0
    print(f"half=2^{lg}: {dt*1e3:.2f} ms")
