"""VERDICT r05 item 2, measured: what would a 13-row table with 20-bit windows (one set of 2^19 buckets, 13 mixed
additions per term instead of 16, no recombination chain) cost at n = 2^20?

Its bucket pass = 13 x 2^20 mixed additions whose table lines are gathered from 13 rows x 128 MiB = 1.66 GB (far
beyond the 256-MiB Infinity Cache).  EXACTLY that pass runs today, with no new kernel, as a commitment to scalars
below 2^207 over the 16-row table (rows spaced 16 bits: 13 non-zero digit rows, each gathered from its own
128-MiB row; top digit < 2^15 so that no carry reaches row 13).  What differs from the real thing is only the bucket
geometry (2^15 buckets of ~416 entries instead of 2^19 of ~26: more split buckets here, a larger reduction there);
the sort moves the same 13.6 M entries.  Beside it: the full 253-bit commitment over 1 / 4 / 8 / 16 rows, alone
(latency, stage by stage) and as passes of three on one stream (throughput).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import verifiable_mpc_amd as vm  # noqa: E402

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
rng = np.random.default_rng(3)


def rs(n, top_bits=252):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    full, part = divmod(top_bits, 8)
    a[:, full + (1 if part else 0):] = 0
    if part:
        a[:, full] &= (1 << part) - 1
    return a


ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
full = [vm.ScalarVector.from_array(rs(n)) for _ in range(3)]
short = [vm.ScalarVector.from_array(rs(n, 207)) for _ in range(3)]          # 13 digits of 16 bits, top one < 2^15
out = ctx.alloc(128 * 3)


def measure(label, t, rows, scs):
    for _ in range(3):
        ctx.msm_table(t.ptr, t.n, 0, scs[0].ptr, n, None, out.ptr, None, rows=rows)
    ctx.sync()
    ctx.profile(True)
    ctx.profile_read(reset=True)
    reps = 8
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_table(t.ptr, t.n, 0, scs[0].ptr, n, None, out.ptr, None, rows=rows)
        ctx.sync()
    alone = (time.perf_counter() - t0) / reps * 1e3
    st = {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in ctx.profile_read(reset=True).items() if ms > 0}
    ctx.profile(False)
    # passes of three commitments, back to back on the one stream
    ptrs = [s.ptr for s in scs]
    for _ in range(2):
        ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows)
    ctx.sync()
    per = (time.perf_counter() - t0) / reps / 3 * 1e3
    ctx.profile(True)
    ctx.profile_read(reset=True)
    for _ in range(4):
        ctx.msm_table_batch(t.ptr, t.n, 0, ptrs, n, None, out.ptr, None, rows)
        ctx.sync()
    st3 = {k: round(ms / max(c, 1) * 1e3) for k, (ms, c) in ctx.profile_read(reset=True).items() if ms > 0}
    ctx.profile(False)
    print(f"{label:34s} table {rows * n * 128 >> 20:5d} MiB  alone {alone:6.3f} ms  3-per-pass {per:6.3f} ms/commitment  {st}\n{'':34s} stages of a pass of three: {st3} = {sum(st3.values())} us",
          flush=True)


for rows in [int(r) for r in os.environ.get('ROWS', '1,4,8,16,13').split(',')]:
    prep = vm.PointVector(pts.a, None, ctx).precompute([], rows=rows)
    measure(f"rows={rows:2d} 253-bit scalars ({13 if rows == 13 else 16} digits)", prep._table, rows, full)
    if rows == 16:
        measure("rows=16 207-bit scalars (13 digits)", prep._table, rows, short)
    del prep
