import sys, time, json
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
ctx = vm.get_context(); rng = np.random.default_rng(3)
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
for k in (10, 12, 14, 15, 16, 17, 18):
  for pre in (False, True):
    n = (1 << k) - 1
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)))
    gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
    if pre: g.precompute([gens["h"], gens["k"]])
    x = vm.ScalarVector.from_array(rs(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
    y = gf(L(x)); P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
    vm.compressed_pivot.generators_digest(gens)
    ts = []
    for _ in range(4):
        r = vm.ScalarVector.from_array(rs(rng, n)); ctx.sync(); t0 = time.perf_counter()
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
        ctx.sync(); ts.append((time.perf_counter() - t0) * 1e3)
    print(k, pre, [round(t, 2) for t in ts[1:]], flush=True)
