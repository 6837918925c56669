#!/usr/bin/env python3
"""Developer probe: wall time per Protocol-4 round (compact mode)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import compressed_pivot as cp, pivot


def rand_scalars(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "compact"
ctx = vm.get_context(); rng = np.random.default_rng(3); n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rand_scalars(rng, n)), keep_proj=(mode == "reference"))
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
if os.environ.get("PRECOMPUTE", "1") == "1": g.precompute([gens["h"], gens["k"]])
x = vm.ScalarVector.from_array(rand_scalars(rng, n)); L = pivot.LinearForm(vm.ScalarVector.from_array(rand_scalars(rng, n)))
y = gf(L(x)); P = pivot.vector_commitment(x, 777, g, gens["h"])
stamps = []
orig = cp._Transcript.round_challenge
def rc(self, *a, **kw):
    t0 = time.perf_counter(); c = orig(self, *a, **kw); stamps.append((t0, time.perf_counter())); return c
cp._Transcript.round_challenge = rc
for rep in range(2):
    stamps.clear()
    r = vm.ScalarVector.from_array(rand_scalars(rng, n))
    if mode == "compact": cp.generators_digest(gens)
    ctx.sync(); t0 = time.perf_counter()
    proof = cp.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript=mode, r=r, rho=5)
    ctx.sync(); t1 = time.perf_counter()
print(f"total {1e3*(t1-t0):.1f} ms; before first round hash: {1e3*(stamps[0][0]-t0):.1f} ms")
prev = stamps[0][0]
for i, (a, b) in enumerate(stamps):
    nxt = stamps[i + 1][0] if i + 1 < len(stamps) else t1
    print(f"round {i:2d}: hash {1e3*(b-a):6.2f} ms | until next round's hash starts {1e3*(nxt-b):6.2f} ms")
