#!/usr/bin/env python3
"""Host wall time of the steps of one compact Protocol-5 prove around the native round chain (the prelude the
interpreter drives): python3 scripts/prelude_walls.py [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import compressed_pivot as cp, pivot


def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a


k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context()
rng = np.random.default_rng(3)
n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rs(rng, n))
L = pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x))
P = pivot.vector_commitment(x, 777, g, gens["h"])
cp.generators_digest(gens)

log = []


def timed(mod, name):
    real = getattr(mod, name)

    def w(*a, **kw):
        t0 = time.perf_counter()
        out = real(*a, **kw)
        log.append((name, (time.perf_counter() - t0) * 1e3))
        return out
    setattr(mod, name, w)


for mod, name in ((cp, "_form_digest_begin"), (pivot, "_commit_launch"), (cp, "_form_digest"), (cp, "_p5_challenges"),
                  (cp, "_extend_form"), (cp, "_p5_setup"), (cp, "_protocol_4_native_rounds")):
    timed(mod, name)
for rep in range(4):
    log.clear()
    r = vm.ScalarVector.from_array(rs(rng, n))
    ctx.sync()
    t0 = time.perf_counter()
    proof = cp.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
    ctx.sync()
    total = (time.perf_counter() - t0) * 1e3
print(f"prove {total:.2f} ms")
print("  ".join(f"{n}={t:.3f}" for n, t in log))
