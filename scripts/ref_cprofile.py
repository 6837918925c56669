import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import verifiable_mpc_amd as vm
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
k = 20
ctx = vm.get_context(); rng = np.random.default_rng(3); n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(rs(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
for rep in range(2):
    r = vm.ScalarVector.from_array(rs(rng, n))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="reference", r=r, rho=5)
r = vm.ScalarVector.from_array(rs(rng, n))
pr = cProfile.Profile()
ctx.sync(); t0 = time.perf_counter()
pr.enable()
proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="reference", r=r, rho=5)
pr.disable()
print("prove ms", (time.perf_counter() - t0) * 1e3)
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
st.print_callers("sync")
st.print_callers("upload")
st.print_callers("download")
