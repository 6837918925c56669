"""Stage-by-stage GPU time of one compact Protocol-5 prove at N = 2^k (HIP events of the library's own stage
brackets on the prover's stream)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
def rs(rng, n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = vm.get_context()
rng = np.random.default_rng(3)
n = (1 << k) - 1
group = vm.EllipticCurve("Ed25519", "projective")
gf = vm.GF(group.order)
g = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(rng, n)), keep_proj=False)
gens = {"g": g, "h": group.generator, "k": vm.Ed25519Point.repeat(group.generator, 12345)}
g.precompute([gens["h"], gens["k"]], wide=os.environ.get("VMPC_CRS_WIDE", "1") != "0")
x = vm.ScalarVector.from_array(rs(rng, n))
L = vm.pivot.LinearForm(vm.ScalarVector.from_array(rs(rng, n)))
y = gf(L(x))
P = vm.pivot.vector_commitment(x, 777, g, gens["h"])
vm.compressed_pivot.generators_digest(gens)
for rep in range(3):
    r = vm.ScalarVector.from_array(rs(rng, n))
    ctx.sync()
    if rep == 2:
        ctx.profile(True); ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
    ctx.sync()
    dt = (time.perf_counter() - t0) * 1e3
st = ctx.profile_read(reset=True)
ctx.profile(False)
print(f"prove {dt:.2f} ms (profiled run)")
tot = 0
for name, (ms, c) in sorted(st.items(), key=lambda kv: -kv[1][0]):
    if c:
        print(f"  {name:22s} {c:4d} x {ms / c * 1e3:8.1f} us = {ms:7.3f} ms")
        tot += ms
print(f"  sum of stages {tot:.3f} ms")

# per round: drive the rounds from Python (one C call per round) and read the stage sums after each
if len(sys.argv) > 2:
    vm.compressed_pivot.NATIVE_CHAIN = False
    real = vm._native.P4Rounds.round
    log = []
    def wrapped(self, c=None):
        t0 = time.perf_counter()
        out = real(self, c)
        wall = (time.perf_counter() - t0) * 1e3
        log.append((wall, {k: round(ms * 1e3) for k, (ms, cnt) in ctx.profile_read(reset=True).items() if cnt}))
        return out
    vm._native.P4Rounds.round = wrapped
    ctx.profile(True); ctx.profile_read(reset=True)
    r = vm.ScalarVector.from_array(rs(rng, n))
    proof = vm.compressed_pivot.protocol_5_prover(gens, P, L, y, x, 777, gf, transcript="compact", r=r, rho=5)
    for i, (wall, st) in enumerate(log):
        if "short_bins" in st:       # the fused three-launch path (csrc/msm_short.hip)
            print(f"round {i:2d} wall {wall:6.3f} ms  short path: scatter {st.get('short_scatter')} bins {st.get('short_bins')} "
                  f"combine {st.get('short_combine')} fold {st.get('table_fold')} "
                  f"scal {sum(st.get(k, 0) for k in ('p4_fold_dots','fr_tail_scalars','p4_extras'))}")
            continue
        print(f"round {i:2d} wall {wall:6.3f} ms  bucket {st.get('msm_bucket')} reduce {st.get('msm_reduce')} final {st.get('msm_final')} "
              f"sort {sum(st.get(k, 0) for k in ('msm_recode','msm_hist','msm_part','msm_sort','msm_plan'))} fold {st.get('table_fold')} "
              f"finish {st.get('msm_bucket_finish')} scal {sum(st.get(k, 0) for k in ('p4_fold_dots','fr_tail_scalars','p4_extras'))}")
