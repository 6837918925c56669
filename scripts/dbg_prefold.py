import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import _native
times = {}
def wrap(cls, name):
    orig = getattr(cls, name)
    def w(self, *a, **k):
        t0 = time.perf_counter()
        try:
            return orig(self, *a, **k)
        finally:
            times.setdefault(name, []).append(round((time.perf_counter() - t0) * 1e3, 2))
    setattr(cls, name, w)
for nm in ("prefold", "round", "round_begin", "round_end"):
    wrap(_native.P4Rounds, nm)
rng = np.random.default_rng(1)
N = 1 << 20; n = N - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 77)}
g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 5, g, gens["h"])
for rep in range(3):
    times.clear()
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 5, gf, r=vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), rho=3)
print(times)
