import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pivot, compressed_pivot as cp
ctx = vm.get_context()
rng = np.random.default_rng(99)
N = 1 << 20; n = N - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 0x1234567)}
g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 5, g, gens["h"])
proof = cp.protocol_5_prover(gens, P, L, y, x, 5, gf, r=vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), rho=3)
log = []
def wrap(mod, name):
    orig = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = orig(*a, **k); log.append((name, (time.perf_counter() - t0) * 1e3, time.perf_counter())); return r
    setattr(mod, name, w)
wrap(pivot, "vector_commitment"); wrap(cp._GroupCheck, "result"); wrap(pivot, "fiat_shamir_hash"); wrap(cp, "_fold_commitment"); wrap(cp, "_fold_form")
for rep in range(2):
    del log[:]
    t0 = time.perf_counter()
    ok = cp.protocol_5_verifier(gens, P, L, y, proof, gf)
    t1 = time.perf_counter()
last_hash_end = [e[2] for e in log if e[0] == "fiat_shamir_hash"][-1]
print("after last hash:", round((t1 - last_hash_end) * 1e3, 2), "ms")
for name, ms, t in log:
    if t > last_hash_end:
        print(f"  {name} {ms:.2f} ms (ends {1e3*(t-last_hash_end):.2f})")
