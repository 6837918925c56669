"""K commitments per pass over prepared generators, DISTINCT vs identical scalar vectors, alone on the GPU and
with three passes in flight: where the batched pass's time goes (stage profile)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import parallel

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
rng = np.random.default_rng(3)
def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
prep = vm.PointVector(pts.a, None, ctx).precompute([], rows=1)
vecs = [vm.ScalarVector.from_array(rs(n)) for _ in range(4)]
t = prep._table
out = ctx.alloc(128 * 16)
for K in (1, 2, 3, 4):
    for label, sc in (("distinct", vecs[:K]), ("identical", [vecs[0]] * K)):
        ptrs = [s.ptr for s in sc]
        for _ in range(2):
            ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows=1)
        ctx.sync()
        ctx.profile(True); ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows=1)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps * 1e3
        st = {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in ctx.profile_read(reset=True).items()}
        ctx.profile(False)
        print(f"K={K} {label:9s} {dt:7.3f} ms/pass {dt / K:6.3f} ms/commitment  {st}", flush=True)
