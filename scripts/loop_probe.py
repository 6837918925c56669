"""bench.py's launch/finish loop in isolation: per-iteration wall time for the prepared-generator path."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import verifiable_mpc_amd as vm
from verifiable_mpc_amd import parallel

n = 1 << 20
rng = np.random.default_rng(3)
def rs(n):
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x0F; return a
ctx = vm.get_context()
group = vm.EllipticCurve("Ed25519", "projective")
pts = vm.PointVector.fixed_base(group.generator, vm.ScalarVector.from_array(rs(n)), keep_proj=False)
prep = vm.PointVector(pts.a, None, ctx).precompute([], rows=1)
nvec = int(sys.argv[1]) if len(sys.argv) > 1 else 3
vecs = [vm.ScalarVector.from_array(rs(n)) for _ in range(nvec)]
shard = parallel.ShardedMsm(ctx, 1, 0, None, torch)
for label, prof in (("profile off", False), ("profile on", True)):
    for K in (1, 3):
        sc = vecs[0] if K == 1 else [vecs[i % nvec] for i in range(K)]
        for _ in range(2):
            shard.finish(shard.launch(sc, prep, 0))
        ctx.profile(prof)
        ts = []
        for i in range(10):
            t0 = time.perf_counter()
            shard.finish(shard.launch(sc, prep, 0))
            ts.append((time.perf_counter() - t0) * 1e3)
        if prof:
            ctx.profile_read(reset=True)
        ctx.profile(False)
        print(f"nvec={nvec} {label} K={K}:", " ".join(f"{t:.2f}" for t in ts), flush=True)

import threading
def variant(label, query, timer):
    t = None
    if timer:
        t = threading.Timer(1000.0, lambda: None); t.daemon = True; t.start()
    for K in (1, 3):
        sc = vecs[0] if K == 1 else [vecs[i % nvec] for i in range(K)]
        ts = []
        for i in range(10):
            t0 = time.perf_counter()
            h = shard.launch(sc, prep, 0)
            if query:
                shard.ready(0)
            shard.finish(h)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{label} K={K}:", " ".join(f"{t:.2f}" for t in ts), flush=True)
    if t: t.cancel()
variant("plain          ", False, False)
variant("query after launch", True, False)
variant("timer thread   ", False, True)
variant("query + timer  ", True, True)
