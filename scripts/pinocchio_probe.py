#!/usr/bin/env python3
"""Developer probe: the eight sums of a Pinocchio proof over a prepared key of 2^k terms (synthetic key
built on the device; timing only - correctness is tests/test_gpu_bn256.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pynocchio as pn

ctx = vm.get_context()
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << lg
key = pn.PreparedKey.synthetic(ctx, n)
rng = np.random.default_rng(2)


class H:
    coeffs = [int(x) for x in rng.integers(1, 2**62, size=n)]
    def __len__(self): return len(self.coeffs)


class D:
    v, w, y = 11, 22, 33
c = [int(x) for x in rng.integers(1, 2**62, size=n)]       # host conversion cost is not the subject here
for rep in range(3):
    t0 = time.perf_counter()
    proof = pn.compute_proof(None, c, H(), key, D)
    print(f"compute_proof over a prepared key, 2^{lg} terms: {(time.perf_counter() - t0) * 1e3:.1f} ms "
          f"(includes {2 * n} host int -> bytes conversions)")
t0 = time.perf_counter()
c_arr, h_arr = pn.scalars_to_array(c), pn.scalars_to_array(H.coeffs)
print(f"  of which host conversion: {(time.perf_counter() - t0) * 1e3:.1f} ms")
for rep in range(3):
    t0 = time.perf_counter()
    proof2 = pn.compute_proof(None, c_arr, h_arr, key, D)
    print(f"  the same with c and h handed over as (n, 32) uint8 arrays: {(time.perf_counter() - t0) * 1e3:.1f} ms")
assert proof2 == proof
