#!/usr/bin/env python3
"""Many compact proofs back to back over random sizes (2^3 .. 2^13), every one verified - the rounds queued ahead of
their challenge (vmpc_p4_run_compact) must neither hang nor ever hand a round a stale challenge.
    python3 scripts/prove_stress.py [seconds] [seed]"""
import os, sys, time, random
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
nrng = np.random.default_rng(seed)
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
ELL = group.order


def rs(n):
    a = nrng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a


crs = {}
t_end = time.time() + budget
count = 0
while time.time() < t_end:
    k = rng.randrange(3, 14)
    n = (1 << k) - 1
    if k not in crs:
        g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
        gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 1000 + k)}
        g.precompute([gens["h"], gens["k"]])
        crs[k] = gens
    gens = crs[k]
    x = vm.ScalarVector.from_array(rs(n))
    L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(n)))
    y = gf(L(x))
    gamma = rng.randrange(1, ELL)
    P = vm.pivot.vector_commitment(x, gamma, gens["g"], gens["h"])
    r = vm.ScalarVector.from_array(rs(n))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, gamma, gf, transcript="compact", r=r, rho=rng.randrange(ELL))
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript="compact") is True, (k, count)
    count += 1
print(f"prove stress ok: {count} proofs verified in {budget:.0f} s")
