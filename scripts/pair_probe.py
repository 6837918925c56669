"""Why is the prover's A_i / B_i pair (two half-populated N-term commitments) slower in the bucket stage than one
full N-term commitment with the same number of additions?"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm
n = 1 << 20
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(3)
def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
prep = vm.PointVector(pts.a, None, ctx).precompute([], rows=rows)
t = prep._table
full = [rs(n) for _ in range(2)]
def masked(a, keep):
    b = a.copy(); b[~keep] = 0; return b
idx = np.arange(n)
cases = {
    "K=1 full": [full[0]],
    "K=2 full": full,
    "K=2 halves (left/right)": [masked(full[0], idx < n // 2), masked(full[1], idx >= n // 2)],
    "K=2 halves (blocks of 2^15)": [masked(full[0], (idx >> 15) % 2 == 0), masked(full[1], (idx >> 15) % 2 == 1)],
    "K=2 halves (odd/even)": [masked(full[0], idx % 2 == 0), masked(full[1], idx % 2 == 1)],
    "K=1 half (left)": [masked(full[0], idx < n // 2)],
}
out = ctx.alloc(128 * 4)
for label, arrs in cases.items():
    vecs = [vm.ScalarVector.from_array(a) for a in arrs]
    ptrs = [v.ptr for v in vecs]
    for _ in range(2):
        ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows=rows)
    ctx.sync()
    ctx.profile(True); ctx.profile_read(reset=True)
    for _ in range(5):
        ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows=rows)
        ctx.sync()
    st = {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in ctx.profile_read(reset=True).items() if ms > 0}
    ctx.profile(False)
    print(f"rows={rows} {label:30s} {st}", flush=True)
