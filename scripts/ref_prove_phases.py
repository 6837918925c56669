#!/usr/bin/env python3
"""Where the reference-transcript prove spends its wall time besides SHA-256: cumulative host time inside the calls of
the round loop (monkeypatched timers).  python3 scripts/ref_prove_phases.py [k]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import compressed_pivot as cp
from verifiable_mpc_amd import device, pivot

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context()
rng = np.random.default_rng(99)
N = 1 << k
n = N - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 0x1234567)}
if os.environ.get("VMPC_CRS_TABLE", "1") != "0":       # as circuit_sat.create_generators hands a CRS over
    g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n))
L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x))
P = vm.pivot.vector_commitment(x, 0x7654321, g, gens["h"])
T = {}


def timed(mod, name, label=None):
    fn = getattr(mod, name)
    label = label or name

    def w(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            T[label] = T.get(label, 0.0) + time.perf_counter() - t0
    setattr(mod, name, w)


timed(pivot, "vector_commitment_pair")
timed(pivot, "vector_commitment")
timed(pivot, "fiat_shamir_hash")
timed(pivot, "fiat_shamir_hash_variants")
timed(cp, "_fold_commitment")
timed(cp, "_fold_form")
timed(cp, "_fold_witness")
timed(cp, "_round_prover_scalars")
for _name in ("round", "round_begin", "round_end", "prefold"):
    timed(vm._native.P4Rounds, _name, "P4Rounds." + _name)
timed(device.PointVector, "fold")
timed(device.PointVector, "text", "PointVector.text (wait for format + copy)")
timed(device.ScalarVector, "text", "ScalarVector.text (wait)")
timed(device.PointVector, "text_begin", "PointVector.text_begin")
timed(device.ScalarVector, "text_begin", "ScalarVector.text_begin")
for rep in range(3):
    r = vm.ScalarVector.from_array(bench.rand_scalars(rng, n))
    ctx.sync()
    T.clear()
    pivot.hash_stats(reset=True)
    t0 = time.perf_counter()
    proof = cp.protocol_5_prover(gens, P, L, y, x, 0x7654321, gf, transcript="reference", r=r, rho=0x1111)
    ctx.sync()
    total = time.perf_counter() - t0
hs = pivot.hash_stats()
print(f"prove {total * 1e3:.1f} ms; inside sha256.update {hs['seconds'] * 1e3:.1f} ms over {hs['bytes'] / 1e6:.0f} MB")
for name, sec in sorted(T.items(), key=lambda kv: -kv[1]):
    print(f"  {name:50s} {sec * 1e3:8.1f} ms")
print("  (fiat_shamir_* include the text waits and the hashing)")
