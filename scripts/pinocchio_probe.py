#!/usr/bin/env python3
"""Developer probe: the eight sums of a Pinocchio proof over a prepared key of 2^k terms (synthetic key
built on the device; timing only - correctness is tests/test_gpu_bn256.py)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pynocchio as pn

ctx = vm.get_context()
G1 = (1).to_bytes(32, "little") + (pn.P - 2).to_bytes(32, "little")
G2v = (64746500191241794695844075326670126197795977525365406531717464316923369116492,
       21167961636542580255011770066570541300993051739349375019639421053990175267184,
       17778617556404439934652658462602675281523610326338642107814333856843981424549,
       20666913350058776956210519119118544732556678129809273996262322366050359951122)
G2 = b"".join(v.to_bytes(32, "little") for v in G2v)
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << lg
key = pn.PreparedKey.__new__(pn.PreparedKey)
key.ctx, key.mid, key.vectors = ctx, list(range(n)), {}
key.mid_index, key.zk_missing = np.arange(n), {}
for name in list(pn._ELEMENTS) + ["h*g1"]:
    grp, gen, width = (2, G2, 128) if name.endswith("g2") else (1, G1, 64)
    extra = len(pn._ELEMENTS[name][1]) if name in pn._ELEMENTS else 0
    pts = ctx.upload(np.tile(np.frombuffer(gen, np.uint8), (n + extra, 1)))
    key.vectors[name] = pn._KeyVector.from_device(ctx, grp, pts, n + extra)
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
