#!/usr/bin/env python3
"""compact prove at N = 2^k with a fold jump forced at small sizes - development reproducer"""
import os, sys, random
os.environ.setdefault("VMPC_P4_JUMP", "5"); os.environ.setdefault("VMPC_P4_JUMP_MIN_LOG2", "6")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(1); ELL = vm.groups.ORDER
n = (1 << log_n) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
h, k = group.generator, vm.Ed25519Point.repeat(group.generator, 77)
g = vm.PointVector.fixed_base(h, [rng.randrange(1, ELL) for _ in range(n)]); g.precompute([h, k])
gens = {"g": g, "h": h, "k": k}
xs = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)])
Lf = vm.pivot.LinearForm(vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(n)]))
P = vm.pivot.vector_commitment(xs, 5, g, h); y = gf(Lf(xs))
print("proving", flush=True)
proof = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, 5, gf, transcript="compact", r=[1] * n, rho=3)
print("ok", vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, proof, gf, transcript="compact"))
