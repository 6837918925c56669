"""world_size-2 gloo test (CPU) of the multi-GPU commitment's host logic: cyclic sharding,
the single all-gather of 128-byte partial points, rank-ordered combine.  The oracle stands in
for the two device routines (local MSM, ordered point sum)."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ed25519_ref as ed  # noqa: E402


def ext_bytes(pt):
    """(X:Y:Z) -> 128-byte extended encoding X||Y||Z||T with T = XY/Z scaled: use Z' = Z^2."""
    x, y, z = pt
    X, Y, Z, T = x * z % ed.P, y * z % ed.P, z * z % ed.P, x * y % ed.P
    return b"".join(v.to_bytes(32, "little") for v in (X, Y, Z, T))


class OracleBackend:
    n_slots = 2

    def __init__(self):
        self.partials = {}

    def launch_partial(self, scalars, points, slot, want_affine):
        assert not want_affine
        acc = ed.IDENTITY
        for s, p in zip(scalars, points):
            acc = ed.pt_add(acc, ed.pt_repeat(p, s))
        self.partials[slot] = torch.frombuffer(bytearray(ext_bytes(acc)), dtype=torch.uint8)

    def wait(self, slot):
        pass

    def partial_tensor(self, slot):
        return self.partials[slot]

    def new_gather_buffer(self, world):
        return torch.zeros((world, 128), dtype=torch.uint8)

    def combine(self, gathered, world, slot=0):
        acc = ed.IDENTITY
        raw = gathered.numpy().tobytes()
        for r in range(world):          # rank order
            X, Y, Z = (int.from_bytes(raw[128 * r + 32 * i:128 * r + 32 * i + 32], "little") for i in range(3))
            acc = ed.pt_add(acc, (X, Y, Z))
        return ed.pt_affine(acc)


def worker(rank, world, port, n, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from verifiable_mpc_amd import parallel
    rng = random.Random(42)                     # same global inputs on every rank
    exps = [rng.randrange(1, ed.ELL) for _ in range(n)]
    pts = [ed.pt_repeat(ed.BASE, e) for e in exps]
    sc = [rng.randrange(ed.ELL) for _ in range(n)]
    idx = parallel.cyclic_indices(n, world, rank)
    sh = parallel.ShardedMsm(None, world, rank, dist, torch, backend=OracleBackend())
    got = sh.commit([sc[i] for i in idx], [pts[i] for i in idx])
    want = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(sc, exps)) % ed.ELL))
    # two commitments in flight (slots 0 and 1), finished in order
    sc2 = [(v * 3 + 1) % ed.ELL for v in sc]
    h0 = sh.launch([sc[i] for i in idx], [pts[i] for i in idx], 0)
    h1 = sh.launch([sc2[i] for i in idx], [pts[i] for i in idx], 1)
    got0, got1 = sh.finish(h0), sh.finish(h1)
    want2 = ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(sc2, exps)) % ed.ELL))
    ret[rank] = (got == want and got0 == want and got1 == want2, list(idx[:3]))
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_commit_world2():
    world, n = 2, 11                           # ragged: shards of 6 and 5 terms
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), n, ret), nprocs=world, join=True)
    assert ret[0][0] and ret[1][0]
    assert ret[0][1] == [0, 2, 4] and ret[1][1] == [1, 3, 5]


def test_cyclic_sharding_helpers():
    from verifiable_mpc_amd import parallel
    a = np.arange(10 * 4).reshape(10, 4)
    parts = [parallel.shard_rows(a, 4, r) for r in range(4)]
    assert sum(len(p) for p in parts) == 10
    assert [list(parallel.cyclic_indices(10, 4, r)) for r in range(4)] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    assert (parts[1] == a[[1, 5, 9]]).all()


# ---- bench.py's own driver loop (run_steps) through the collective branch, two ranks ------------------------
class BatchOracleBackend(OracleBackend):
    """three slots, launches of several commitments (the layout HipBackend produces: b points per rank)"""
    n_slots = 3
    max_batch = 4

    def __init__(self):
        super().__init__()
        self.batch_of = {}

    def launch_partial(self, scalars, points, slot, want_affine):
        assert not want_affine
        many = scalars if isinstance(scalars[0], (list, tuple)) else [scalars]
        raw = b""
        for sc in many:
            acc = ed.IDENTITY
            for s, p in zip(sc, points.pts):
                acc = ed.pt_add(acc, ed.pt_repeat(p, s))
            raw += ext_bytes(acc)
        self.batch_of[slot] = len(many)
        self.partials[slot] = torch.frombuffer(bytearray(raw), dtype=torch.uint8)

    def new_gather_buffer(self, world):
        return torch.zeros((world, 128 * self.max_batch), dtype=torch.uint8)

    def combine(self, gathered, world, slot=0):
        b = self.batch_of[slot]
        raw = gathered.view(-1)[:world * 128 * b].numpy().tobytes()
        out = []
        for j in range(b):
            acc = ed.IDENTITY
            for r in range(world):          # rank order
                o = 128 * (r * b + j)
                X, Y, Z = (int.from_bytes(raw[o + 32 * i:o + 32 * i + 32], "little") for i in range(3))
                acc = ed.pt_add(acc, (X, Y, Z))
            out.append(ed.pt_affine(acc))
        return out if b > 1 else out[0]


class _Pts:
    def __init__(self, pts, tabulated):
        self.pts, self._table = pts, (object() if tabulated else None)


def bench_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from verifiable_mpc_amd import parallel
    n = 9
    rng = random.Random(77)
    exps = [rng.randrange(1, ed.ELL) for _ in range(n)]
    pts = [ed.pt_repeat(ed.BASE, e) for e in exps]
    vecs = [[rng.randrange(ed.ELL) for _ in range(n)] for _ in range(3)]
    idx = parallel.cyclic_indices(n, world, rank)
    shard = parallel.ShardedMsm(None, world, rank, dist, torch, backend=BatchOracleBackend())
    assert shard.collective and shard.n_slots == 3
    local = [[v[i] for i in idx] for v in vecs]
    want = [ed.pt_affine(ed.pt_repeat(ed.BASE, sum(a * b for a, b in zip(v, exps)) % ed.ELL)) for v in vecs]
    ok = True
    # batch 3 over "tabulated" generators, three launches in flight; 7 steps = launches of 3, 3, 1
    res, which = bench.run_steps(shard, 7, local, _Pts([pts[i] for i in idx], True), depth=3, batch=3)
    ok &= which == [0] and res == [want[0]]
    res, which = bench.run_steps(shard, 6, local, _Pts([pts[i] for i in idx], True), depth=3, batch=3)
    ok &= which == [0, 1, 2] and res == want
    # one commitment per launch (plain generators): cycles through the vectors
    res, which = bench.run_steps(shard, 5, local, _Pts([pts[i] for i in idx], False), depth=3, batch=3)
    ok &= which == [4 % 3] and res == [want[1]]
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_run_steps_collective_world2():
    """bench.py's launch/refill loop with a collective: batch 3, three slots, two ranks - every rank must enter the
    all-gathers in the same order and end with the commitment over BOTH shards"""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(bench_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    assert ret[0] and ret[1]
