#!/usr/bin/env python3
"""Developer probe: where Protocol-5 prove time goes at N = 2^k (not product/tests)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm


def rand_scalars(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F
    return a


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["compact", "reference"]
    ctx = vm.get_context()
    rng = np.random.default_rng(3)
    n = (1 << k) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rand_scalars(rng, n)))
    gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
    if os.environ.get("PRECOMPUTE", "1") == "1":
        g.precompute([gens["h"], gens["k"]])
    x = vm.ScalarVector.from_array(rand_scalars(rng, n))
    L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rand_scalars(rng, n)))
    y = gf(L(x))
    P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
    for mode in modes:
        r = vm.ScalarVector.from_array(rand_scalars(rng, n))
        if mode == "compact":
            vm.compressed_pivot.generators_digest(gens)
        ctx.sync()
        ctx.profile(True)
        ctx.profile_read(reset=True)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript=mode, r=r, rho=5)
        pr.disable()
        ctx.sync()
        dt = time.perf_counter() - t0
        prof = ctx.profile_read(reset=True)
        ctx.profile(False)
        print(f"== {mode}: prove {dt*1e3:.1f} ms; kernel stage totals (ms):")
        tot = 0
        for name, (ms, cnt) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
            print(f"   {name:20s} {ms:9.2f}  x{cnt}")
            tot += ms
        print(f"   {'sum of kernels':20s} {tot:9.2f}")
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


if __name__ == "__main__":
    main()
