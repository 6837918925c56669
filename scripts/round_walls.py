#!/usr/bin/env python3
"""Wall time of every Protocol-4 round of one compact prove at N = 2^k, rounds driven from Python (one C call per
round), NO stage events on the stream.   python3 scripts/round_walls.py [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context()
rng = np.random.default_rng(3)
n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=False)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rs(rng, n))
L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x))
P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
vm.compressed_pivot.generators_digest(gens)
vm.compressed_pivot.NATIVE_CHAIN = False
real = vm._native.P4Rounds.round
log = []
def wrapped(self, c=None):
    t0 = time.perf_counter()
    out = real(self, c)
    log.append((time.perf_counter() - t0) * 1e3)
    return out
vm._native.P4Rounds.round = wrapped
for rep in range(3):
    log.clear()
    r = vm.ScalarVector.from_array(rs(rng, n))
    ctx.sync()
    t0 = time.perf_counter()
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
    ctx.sync()
    total = (time.perf_counter() - t0) * 1e3
print(f"prove (rounds from Python) {total:.2f} ms; rounds sum {sum(log):.2f} ms")
print(" ".join(f"{v:.3f}" for v in log))
