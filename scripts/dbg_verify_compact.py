"""where a compact-transcript verify at N = 2^20 spends its wall time (host view, call by call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pivot, compressed_pivot as cp
ctx = vm.get_context()
rng = np.random.default_rng(99)
N = 1 << 20; n = N - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=False)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 0x1234567)}
g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 5, g, gens["h"])
cp.generators_digest(gens)
proof = cp.protocol_5_prover(gens, P, L, y, x, 5, gf, transcript="compact", r=vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), rho=3)
log = []
def wrap(mod, name):
    orig = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); log.append((name, (t0 - T0[0]) * 1e3, (time.perf_counter() - t0) * 1e3)); return r
    setattr(mod, name, w)
T0 = [0.0]
for mod, name in ((cp, "_valid_group_elements_begin"), (cp, "_form_digest_begin"), (cp, "_p5_challenges"), (cp, "_extend_form"),
                  (cp, "_unfold_commitment"), (pivot, "_commit_launch"), (cp._GroupCheck, "result"), (pivot, "affine_to_linear"),
                  (cp, "_protocol_4_verifier_compact"), (pivot._PendingCommitment, "result"), (vm.ScalarVector, "from_ints")):
    wrap(mod, name)
for rep in range(4):
    del log[:]
    ctx.sync()
    T0[0] = time.perf_counter()
    ok = cp.protocol_5_verifier(gens, P, L, y, proof, gf, transcript="compact")
    total = (time.perf_counter() - T0[0]) * 1e3
print(f"verify {total:.2f} ms")
for name, at, ms in log:
    print(f"  at {at:6.2f}  {name:34s} {ms:6.2f} ms")
