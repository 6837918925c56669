"""One 2^k-term commitment alone over a fixed-base table of r rows (r = 1: prepared generators only), stage by
stage: what the shorter recombination chain buys against the larger table's gather."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
rng = np.random.default_rng(3)
def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
sc = vm.ScalarVector.from_array(rs(n))
out = ctx.alloc(128)
for rows in (1, 2, 4, 8, 16):
    prep = vm.PointVector(pts.a, None, ctx).precompute([], rows=rows)
    t = prep._table
    for _ in range(3):
        ctx.msm_table(t.ptr, t.n, 0, sc.ptr, n, None, out.ptr, None, rows=rows)
    ctx.sync()
    ctx.profile(True); ctx.profile_read(reset=True)
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_table(t.ptr, t.n, 0, sc.ptr, n, None, out.ptr, None, rows=rows)
        ctx.sync()
    dt = (time.perf_counter() - t0) / reps * 1e3
    st = {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in ctx.profile_read(reset=True).items() if ms > 0}
    ctx.profile(False)
    print(f"rows={rows:2d} table {rows * n * 128 >> 20:5d} MiB  {dt:6.3f} ms  {st}", flush=True)
    del prep, t
