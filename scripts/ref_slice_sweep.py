import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import verifiable_mpc_amd as vm
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
k = 20
ctx = vm.get_context(); rng = np.random.default_rng(3); n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)))
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rs(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
for sl in [1 << 16, 1 << 17, 1 << 18, 1 << 30, 1 << 16, 1 << 18, 1 << 30]:
    vm.PointVector.TEXT_SLICE = sl
    ts, tv = [], []
    for rep in range(3):
        r = vm.ScalarVector.from_array(rs(rng, n))
        ctx.sync(); t0 = time.perf_counter()
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="reference", r=r, rho=5)
        ts.append((time.perf_counter() - t0) * 1e3)
        ctx.sync(); t0 = time.perf_counter()
        ok = vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript="reference")
        tv.append((time.perf_counter() - t0) * 1e3)
        assert ok
    print("slice 2^%d" % (sl.bit_length() - 1), "prove", [round(t, 1) for t in ts], "verify", [round(t, 1) for t in tv])
