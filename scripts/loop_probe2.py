import sys, time, os, threading
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
vecs = [vm.ScalarVector.from_array(rs(n)) for _ in range(3)]
shard = parallel.ShardedMsm(ctx, 1, 0, None, torch)
def variant(label, query, prof, sync_before):
    sc = vecs
    for _ in range(2):
        shard.finish(shard.launch(sc, prep, 0))
    if sync_before:
        torch.cuda.synchronize()
    ctx.profile(prof)
    ts = []
    for i in range(30):
        t0 = time.perf_counter()
        h = shard.launch(sc, prep, 0)
        if query:
            shard.ready(0)
        shard.finish(h)
        ts.append((time.perf_counter() - t0) * 1e3)
    if prof:
        ctx.profile_read(reset=True)
    ctx.profile(False)
    print(f"{label}: max {max(ts):.2f}  ", " ".join(f"{t:.1f}" for t in ts), flush=True)
variant("plain               ", False, False, False)
variant("query               ", True, False, False)
variant("profile             ", False, True, False)
variant("query+profile       ", True, True, False)
variant("torch sync + q + p  ", True, True, True)
variant("torch sync          ", False, False, True)
