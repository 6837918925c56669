#!/usr/bin/env python3
"""Protocol-5 prove at N = 2^k in ONE transcript mode, a few times - the program rocprofv3 wraps for the
profiles/*_prove_<mode>_kernel_stats.csv summaries (scripts/profile_round.sh).  Prints one JSON line."""
import json
import os
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
    mode = sys.argv[1] if len(sys.argv) > 1 else "compact"
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ctx = vm.get_context()
    rng = np.random.default_rng(3)
    n = (1 << k) - 1
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rand_scalars(rng, n)))
    gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
    g.precompute([gens["h"], gens["k"]], wide=os.environ.get("VMPC_CRS_WIDE", "1") != "0")
    x = vm.ScalarVector.from_array(rand_scalars(rng, n))
    L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rand_scalars(rng, n)))
    y = gf(L(x))
    P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
    if mode == "compact":
        vm.compressed_pivot.generators_digest(gens)
    times = []
    for _ in range(reps + 1):
        r = vm.ScalarVector.from_array(rand_scalars(rng, n))
        ctx.sync()
        t0 = time.perf_counter()
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript=mode, r=r, rho=5)
        ctx.sync()
        times.append((time.perf_counter() - t0) * 1e3)
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, proof, gf, transcript=mode) is True
    print(json.dumps({"mode": mode, "N": 1 << k, "prove_ms": [round(t, 2) for t in times[1:]],
                      "first_call_ms": round(times[0], 2), "proves_in_this_process": reps + 1,
                      "algorithmic_bytes": 768 << k}))


if __name__ == "__main__":
    main()
