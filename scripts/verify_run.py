#!/usr/bin/env python3
"""time protocol_5_verifier (compact) at N = 2^k, several calls in one process"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context(); rng = np.random.default_rng(3); n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)))
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rs(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
r = vm.ScalarVector.from_array(rs(rng, n))
proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
ts = []
for _ in range(6):
    ctx.sync(); t0 = time.perf_counter()
    ok = vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript="compact")
    ts.append((time.perf_counter() - t0) * 1e3); assert ok is True
bad = dict(proof); bad["z_prime"] = [proof["z_prime"][0] + 1, proof["z_prime"][1]]
assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, bad, gf, transcript="compact") is False
print("verify_ms", [round(t, 2) for t in ts])
