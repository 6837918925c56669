#!/usr/bin/env python3
"""Reference-transcript prove / verify at N = 2^k, hash call by hash call: when the call starts (since the start of the
proof), how long it lasts, how much of that is SHA-256 and how much is WAITING for the next piece of text (the stall
the serial chain hash -> fold -> pair -> hash leaves), and the gap before the call (pairs, glue).
    python3 scripts/ref_stall_probe.py [k] [prove|verify]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import device, pivot

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
what = sys.argv[2] if len(sys.argv) > 2 else "prove"
ctx = vm.get_context()
rng = np.random.default_rng(99)
N = 1 << k
n = N - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 0x1234567)}
if os.environ.get("VMPC_CRS_TABLE", "1") != "0":       # as circuit_sat.create_generators hands a CRS over
    g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n))
L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x))
P = vm.pivot.vector_commitment(x, 0x7654321, g, gens["h"])

calls = []
cur = {}
T0 = [0.0]


def wrap_chunks(cls):
    orig = cls.text_chunks

    def text_chunks(self, *a, **kw):
        it = iter(orig(self, *a, **kw))
        while True:
            t0 = time.perf_counter()
            try:
                piece = next(it)
            except StopIteration:
                cur["stall"] = cur.get("stall", 0.0) + time.perf_counter() - t0
                return
            cur["stall"] = cur.get("stall", 0.0) + time.perf_counter() - t0
            yield piece
    cls.text_chunks = text_chunks


wrap_chunks(device.PointVector)
wrap_chunks(device.ScalarVector)


def wrap_hash(name):
    orig = getattr(pivot, name)

    def w(*a, **kw):
        cur.clear()
        before = pivot.hash_stats()
        t0 = time.perf_counter()
        out = orig(*a, **kw)
        t1 = time.perf_counter()
        after = pivot.hash_stats()
        calls.append((t0 - T0[0], t1 - t0, after["seconds"] - before["seconds"], cur.get("stall", 0.0),
                      after["bytes"] - before["bytes"]))
        return out
    setattr(pivot, name, w)


wrap_hash("fiat_shamir_hash")
wrap_hash("fiat_shamir_hash_variants")
proof = None
for rep in range(3):
    r = vm.ScalarVector.from_array(bench.rand_scalars(rng, n))
    ctx.sync()
    del calls[:]
    pivot.hash_stats(reset=True)
    T0[0] = time.perf_counter()
    if what == "prove" or proof is None:
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 0x7654321, gf, r=r, rho=5)
        if what != "prove":
            continue
    else:
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf) is True
    total = time.perf_counter() - T0[0]
st = pivot.hash_stats()
print(f"{what} {total * 1e3:.1f} ms; inside sha256.update {st['seconds'] * 1e3:.1f} ms over {st['bytes'] / 1e6:.0f} MB")
print(" call   start ms   gap before   lasts ms    sha ms   stall ms  other ms      MB")
end_prev = 0.0
for i, (t0, dt, sha, stall, nbytes) in enumerate(calls):
    print(f"{i:5d} {t0 * 1e3:10.2f} {(t0 - end_prev) * 1e3:12.2f} {dt * 1e3:10.2f} {sha * 1e3:9.2f} {stall * 1e3:10.2f} "
          f"{(dt - sha - stall) * 1e3:9.2f} {nbytes / 1e6:7.1f}")
    end_prev = t0 + dt
print(f"after the last call: {(total - end_prev) * 1e3:.2f} ms")
print(f"sums: gaps {sum(c[0] for c in calls[:0]) :.0f}", end="")
gaps = calls[0][0] + sum(calls[i][0] - (calls[i - 1][0] + calls[i - 1][1]) for i in range(1, len(calls)))
print(f" gaps {gaps * 1e3:.1f} ms, stalls {sum(c[3] for c in calls) * 1e3:.1f} ms, other inside calls "
      f"{sum(c[1] - c[2] - c[3] for c in calls) * 1e3:.1f} ms")
