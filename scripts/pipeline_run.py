#!/usr/bin/env python3
"""The bench's timed region alone (prepared generators, `batch` commitments per pass, `depth` passes in flight),
bracketed by marker kernels (k_madd_rate) so that scripts/timeline.py can cut it out of a rocprofv3 kernel trace.
    python3 scripts/pipeline_run.py [log2n] [steps] [batch] [depth]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import parallel

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 36
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 3
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = int(os.environ.get("VMPC_BENCH_TABLE_ROWS", "1"))
n = 1 << log2n
ctx = vm.get_context()
rng = np.random.default_rng(20200153)
group = vm.EllipticCurve("Ed25519", "projective")
points = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(bench.rand_scalars(rng, n)), keep_proj=False)
prepared = vm.PointVector(points.a, None, ctx).precompute([], rows=rows)
svs = [vm.ScalarVector.from_array(bench.rand_scalars(rng, n)) for _ in range(batch)]
shard = parallel.ShardedMsm(ctx, 1, 0)
depth = min(depth, shard.n_slots)
import gc
gc.collect()
gc.freeze()
for slot in range(depth):
    shard.finish(shard.launch(svs[:batch] if batch > 1 else svs[0], prepared, slot))
bench.run_steps(shard, 2 * batch * depth, svs, prepared, depth, batch)
res = []
for rep in range(3):
    for c in shard.backend.ctxs[:depth]:
        c.sync()
    ctx.madd_rate(1)            # marker
    t0 = time.perf_counter()
    bench.run_steps(shard, steps, svs, prepared, depth, batch)
    for c in shard.backend.ctxs[:depth]:
        c.sync()
    dt = time.perf_counter() - t0
    ctx.madd_rate(1)            # marker
    res.append(dt / steps * 1e3)
print(json.dumps({"log2n": log2n, "steps": steps, "batch": batch, "depth": depth, "rows": rows,
                  "ms_per_step": [round(r, 4) for r in res]}))
