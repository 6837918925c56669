import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import pivot
ctx = vm.get_context()
rng = np.random.default_rng(99)
N = 1 << 20; n = N - 1
group = vm.EllipticCurve("Ed25519", "projective"); gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=True)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 0x1234567)}
g.precompute([gens["h"], gens["k"]], wide=True)
x = vm.ScalarVector.from_array(bench.rand_scalars(rng, n)); L = vm.pivot.LinearForm(vm.ScalarVector.from_array(bench.rand_scalars(rng, n)))
y = gf(L(x)); P = vm.pivot.vector_commitment(x, 0x7654321, g, gens["h"])
def run(mode, reps):
    for rep in range(reps):
        r = vm.ScalarVector.from_array(bench.rand_scalars(rng, n))
        ctx.sync(); pivot.hash_stats(reset=True)
        c0 = time.process_time(); t0 = time.perf_counter()
        proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 0x7654321, gf, transcript=mode, r=r, rho=5)
        t1 = time.perf_counter(); ctx.sync(); t2 = time.perf_counter()
        st = pivot.hash_stats()
        print(mode, f"prove {1e3*(t1-t0):.1f} ms + sync {1e3*(t2-t1):.2f}; sha {st['seconds']*1e3:.1f}; outside {1e3*(t2-t0)-st['seconds']*1e3:.1f}; cpu {1e3*(time.process_time()-c0):.0f} ms", flush=True)
    return proof
what = sys.argv[1] if len(sys.argv) > 1 else "ref"
if what.startswith("compact_first"):
    if "nodigest" not in what:
        vm.compressed_pivot.generators_digest(gens)
    p = run("compact", 3)
    if "noverify" not in what:
        assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, p, gf, transcript="compact")
if what == "aux0":
    from verifiable_mpc_amd.device import get_aux_context
    a = get_aux_context(0); a.sync()
if what == "aux0_msm":
    from verifiable_mpc_amd.device import get_aux_context
    a = get_aux_context(0)
    pts = [vm.Ed25519Point.repeat(group.generator, 3 + i) for i in range(39)]
    pv = vm.PointVector.from_points(pts, a, keep_proj=False)
    print(pivot._commit_launch(vm.ScalarVector.from_ints(list(range(1, 40)), a), 0, pv, vm.Ed25519Point.identity, a).result() is not None)
if what == "bigcommit":
    v = vm.ScalarVector.from_array(bench.rand_scalars(rng, N))
    gh = vm.PointVector.concat(g, vm.PointVector.from_points([gens["h"]], ctx)) if hasattr(vm.PointVector, "concat") else None
    print(pivot._commit_launch(v[:n], 5, g, gens["k"], ctx).result() is not None)
if "digest_only" in what:
    vm.compressed_pivot.generators_digest(gens)
p = run("reference", 4)
t0 = time.perf_counter(); assert vm.compressed_pivot.protocol_5_verifier(gens, P, L, y, p, gf, transcript="reference"); print("verify", 1e3*(time.perf_counter()-t0))
p = run("reference", 2)
