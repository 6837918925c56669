#!/usr/bin/env python3
"""compact and reference proofs interleaved in one process (pinned-buffer lifetime check), N = 2^k"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
rng = np.random.default_rng(5); n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rs(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
for mode in ("compact", "reference", "compact", "reference", "compact", "compact"):
    r = vm.ScalarVector.from_array(rs(rng, n))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript=mode, r=r, rho=5)
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript=mode) is True
    print(mode, "ok", flush=True)
