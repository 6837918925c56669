"""The exchange step of the multi-GPU path (include/vmpc.h vmpc_comm_*, csrc/comm.hip) and the sharded round
context (vmpc_p4_create_sharded, csrc/prover.hip) on ONE GPU:

 * rank-ordered sums of gathered partial points against the oracle's pt_add chain (kernel row k8);
 * every transport a single box can run: none (world 1), RCCL with ONE rank, and a callback transport that
   carries the bytes between G ranks living in G THREADS of this process (one vmpc context each);
 * the sharded prover through the C round context for G = 2, 4, 8 blocks, with and without the block-local fold
   of the generators: the proof must be the unsharded compact prover's, value for value.
Multi-process runs (gloo, two ranks sharing the GPU) are in tests/test_gpu_bench_multirank.py.
"""
import random
import threading

import numpy as np
import pytest

from oracle import ed25519_ref as ed

pytestmark = pytest.mark.gpu
ELL = ed.ELL


@pytest.fixture(scope="module")
def vm():
    import verifiable_mpc_amd as v
    v.get_context()
    return v


def ext_bytes(pt, scale):
    """affine oracle point -> 128-byte extended encoding X||Y||Z||T with Z = scale"""
    x, y = ed.pt_affine(pt)
    z = scale % ed.P
    return b"".join(v.to_bytes(32, "little") for v in (x * z % ed.P, y * z % ed.P, z, x * y * z % ed.P))


def oracle_sum(points):
    acc = ed.IDENTITY
    for p in points:            # rank order
        acc = ed.pt_add(acc, p)
    return ed.pt_affine(acc)


def test_points_sum_many_against_oracle_chain(vm):
    """vmpc_points_sum_many_dev on 8 x 2 random extended points (the layout an all-gather of 2 partial points per
    rank produces) == the oracle's pt_add chain in rank order, for both sums"""
    ctx = vm.get_context()
    rng = random.Random(88)
    world, k = 8, 2
    pts = [[ed.pt_repeat(ed.BASE, rng.randrange(1, ELL)) for _ in range(k)] for _ in range(world)]
    raw = b"".join(ext_bytes(pts[r][j], rng.randrange(2, ed.P)) for r in range(world) for j in range(k))
    buf = ctx.upload(np.frombuffer(raw, np.uint8))
    out_ext, out_aff = ctx.alloc(128 * k), ctx.alloc(64 * k)
    ctx.points_sum_many(buf.ptr, world, k, out_ext.ptr, out_aff.ptr)
    ctx.sync()
    aff = ctx.download(out_aff.ptr, 64 * k).tobytes()
    ext = ctx.download(out_ext.ptr, 128 * k).tobytes()
    for j in range(k):
        want = oracle_sum([pts[r][j] for r in range(world)])
        got = (int.from_bytes(aff[64 * j:64 * j + 32], "little"), int.from_bytes(aff[64 * j + 32:64 * j + 64], "little"))
        assert got == want
        X, Y, Z = (int.from_bytes(ext[128 * j + 32 * i:128 * j + 32 * i + 32], "little") for i in range(3))
        assert ed.pt_affine((X, Y, Z)) == want
    # identity and a point with its negative among the summands
    neg = lambda p: ((-ed.pt_affine(p)[0]) % ed.P, ed.pt_affine(p)[1], 1)
    trio = [pts[0][0], ed.IDENTITY, neg(pts[0][0]), pts[1][1]]
    raw = b"".join(ext_bytes(p, 7) for p in trio)
    buf = ctx.upload(np.frombuffer(raw, np.uint8))
    ctx.points_sum_many(buf.ptr, len(trio), 1, None, out_aff.ptr)
    ctx.sync()
    aff = ctx.download(out_aff.ptr, 64).tobytes()
    assert (int.from_bytes(aff[:32], "little"), int.from_bytes(aff[32:], "little")) == ed.pt_affine(pts[1][1])


def test_comm_world1_transports(vm):
    """world = 1: no transport, a callback transport and a real one-rank RCCL communicator give the same sums"""
    from verifiable_mpc_amd._native import Comm
    ctx = vm.get_context()
    rng = random.Random(5)
    k = 3
    pts = [ed.pt_repeat(ed.BASE, rng.randrange(1, ELL)) for _ in range(k)]
    mine = ctx.upload(np.frombuffer(b"".join(ext_bytes(p, rng.randrange(2, ed.P)) for p in pts), np.uint8))
    want = b"".join(a.to_bytes(32, "little") + b.to_bytes(32, "little") for a, b in map(ed.pt_affine, pts))
    calls = []

    def copy_exchange(mine_ptr, gathered_ptr, nbytes):
        calls.append(nbytes)
        ctx.upload_into(gathered_ptr, ctx.download(mine_ptr, nbytes))

    comms = [Comm.solo(), Comm.callback(1, 0, copy_exchange), Comm.rccl(ctx, Comm.unique_id(), 1, 0)]
    assert [c.kind for c in comms] == ["self", "callback", "rccl"]
    for c in comms:
        scratch, out = ctx.alloc(128 * k), ctx.alloc(64 * k)
        c.points_allsum(ctx, mine.ptr, k, scratch.ptr, None, out.ptr)
        ctx.sync()
        assert ctx.download(out.ptr, 64 * k).tobytes() == want, c.kind
        c.close()
    assert calls == [128 * k]


class ThreadRanks:
    """G ranks as G threads of this process, one vmpc context each; a callback transport moves the bytes"""

    def __init__(self, vm, world):
        from verifiable_mpc_amd import _native
        self.world = world
        self.barrier = threading.Barrier(world, timeout=120)
        self.stash = [None] * world
        self.ctxs = [_native.Context(vm.get_context().device) for _ in range(world)]
        self.comms = [_native.Comm.callback(world, r, self._exchange(r)) for r in range(world)]

    def _exchange(self, rank):
        def exchange(mine_ptr, gathered_ptr, nbytes):
            self.stash[rank] = self.ctxs[rank].download(mine_ptr, nbytes).copy()
            self.barrier.wait()
            self.ctxs[rank].upload_into(gathered_ptr, np.concatenate(self.stash))
            self.barrier.wait()                 # nobody refills the stash before everyone has read it
        return exchange

    def run(self, fn):
        out, errs = [None] * self.world, []

        def body(rank):
            try:
                out[rank] = fn(rank, self.ctxs[rank], self.comms[rank])
            except BaseException as e:          # noqa: B902 - reported below, the other ranks must not hang
                errs.append((rank, e))
                self.barrier.abort()
        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(600)
        if errs:
            raise errs[0][1]
        return out

    def close(self):
        for c in self.comms:
            c.close()
        for c in self.ctxs:
            c.close()


def _flat(proof):
    return {k: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                [int(e) for e in v] if isinstance(v, list) else int(v)) for k, v in proof.items()}


@pytest.mark.parametrize("log_n,world,jump_k,min_log2", [
    (7, 2, 0, 30),        # no fold of the generators: every round on the block's table
    (10, 2, 3, 6),        # blocks of 512 fold 2^3 / 2 = 4 strides each into a 128-point partial vector
    (10, 4, 3, 6),        # two strides per block
    (10, 8, 3, 6),        # 2^3 < 2 * 8: a block holds one stride, no fold
    (12, 4, 5, 5),        # fold after 5 rounds, and once more on the (now full-length) partial vectors
    (9, 1, 4, 6),         # one rank through the sharded entry point = the unsharded context plus the exchange
])
def test_sharded_round_context_equals_unsharded_prover(vm, monkeypatch, log_n, world, jump_k, min_log2):
    from verifiable_mpc_amd import _native, sharded
    monkeypatch.setenv("VMPC_P4_JUMP", str(jump_k))
    monkeypatch.setenv("VMPC_P4_JUMP_MIN_LOG2", str(min_log2))
    rng = random.Random(31 * log_n + world)
    N = 1 << log_n
    n = N - 1
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    group = vm.EllipticCurve("Ed25519", "projective")
    gf = vm.GF(group.order)
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, rng.randrange(1, ELL))
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    r = [rng.randrange(ELL) for _ in range(n)]
    gamma, rho = rng.randrange(1, ELL), rng.randrange(ELL)

    g = vm.PointVector.fixed_base(h, exps, keep_proj=False)
    g.precompute([h, k])
    gens = {"g": g, "h": h, "k": k}
    xs, Lf = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
    P = vm.pivot.vector_commitment(xs, gamma, g, h)
    y = gf(Lf(xs))
    want = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                 r=list(r), rho=rho)
    digest = vm.compressed_pivot.generators_digest(gens)
    exps_arr = _native.ints_to_array(exps, 32)
    ranks = ThreadRanks(vm, world)
    native_calls = []
    real_run = _native.P4Rounds.run_compact
    monkeypatch.setattr(_native.P4Rounds, "run_compact",
                        lambda self, *a: native_calls.append(self.comm is not None) or real_run(self, *a))

    def prove(rank, ctx, comm):
        crs = sharded.ShardedCrs.from_exponents(h, k, exps_arr, world, [rank], ctx=ctx, comm=comm)
        d = crs.digest()
        Lr = vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs, ctx))
        xr = vm.ScalarVector.from_ints(x, ctx)
        Pr = crs.commit([(xr.concat([gamma]), None)])[0]
        proof = sharded.protocol_5_prover(crs, Pr, Lr, y, xr, gamma, gf, vm.ScalarVector.from_ints(r, ctx), rho)
        return d, tuple(Pr.normalize().coords), _flat(proof), proof
    try:
        results = ranks.run(prove)
    finally:
        ranks.close()
    assert native_calls == [True] * world            # the rounds ran in the C library, on every rank
    for d, p_coords, flat, _ in results:
        assert d == digest
        assert p_coords == tuple(P.normalize().coords)
        assert flat == _flat(want)
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, results[-1][3], gf, transcript="compact") is True


def test_sharded_round_context_argument_checks(vm):
    from verifiable_mpc_amd import _native
    ctx = vm.get_context()
    h = vm.EllipticCurve("Ed25519", "projective").generator
    k = vm.Ed25519Point.repeat(h, 99)
    g = vm.PointVector.fixed_base(h, list(range(2, 10)), keep_proj=False)       # a block of 8
    g.precompute([k])
    z = vm.ScalarVector.from_ints(list(range(1, 25)))
    three = _native.Comm.callback(3, 0, lambda *a: None)                         # world must be a power of two
    with pytest.raises(_native.VmpcError):
        _native.P4Rounds(ctx, g._table, 0, 0, z.ptr, z.ptr, comm=three)
    three.close()
    solo = _native.Comm.solo()
    rounds = _native.P4Rounds(ctx, g._table, 0, 0, z.ptr, z.ptr, n_total=8, comm=solo)
    # the round context holds the vmpc context's arena and a pointer to it: the context refuses to go
    assert ctx.lib.vmpc_ctx_destroy(ctx.handle) == _native.E_INVAL
    a0, b0 = rounds.round(None)
    with pytest.raises(_native.VmpcError):          # non-canonical challenge: rejected before the state advances
        rounds.round((1 << 256) - 1)
    a1, b1 = rounds.round(5)                        # ... so the context is still usable
    z0, z1 = rounds.finish(7)
    rounds.close()
    solo.close()
    # the same rounds through the unsharded entry point (h given as the table's first extra there)
    g2 = vm.PointVector.fixed_base(h, list(range(2, 9)), keep_proj=False)        # 7 generators + ...
    pt9 = vm.Ed25519Point.repeat(h, 9)
    g2.precompute([pt9, k])                                                      # ... the 8th as the tail slot
    plain = _native.P4Rounds(ctx, g2._table, 1, 1, z.ptr, z.ptr)
    assert plain.round(None) == (a0, b0) and plain.round(5) == (a1, b1) and plain.finish(7) == (z0, z1)
    plain.close()


def test_round_context_rejects_noncanonical_witness_on_the_bucket_free_path(vm):
    """N = 128 with a 16-row table: the rounds commit without buckets (k_p4_direct) and never recode a scalar, so the
    canonical-residue check has to come from vmpc_p4_create itself: a z_hat entry >= l must surface as
    VMPC_E_NONCANON at the first round's synchronisation, on this path like on the bucket path."""
    import numpy as np
    from verifiable_mpc_amd import _native
    ctx = vm.get_context()
    h = vm.EllipticCurve("Ed25519", "projective").generator
    k = vm.Ed25519Point.repeat(h, 99)
    g = vm.PointVector.fixed_base(h, list(range(2, 129)), keep_proj=False)       # 127 generators + h = 128
    g.precompute([h, k], rows=16)
    good = vm.ScalarVector.from_ints(list(range(1, 129)))
    rounds = _native.P4Rounds(ctx, g._table, 1, 1, good.ptr, good.ptr)
    rounds.round(None)                                                            # canonical input: accepted
    rounds.close()
    raw = np.zeros((128, 32), np.uint8)
    raw[:, 0] = 3
    raw[77] = 0xff                                                                # 2^256 - 1 >= l
    bad = vm.ScalarVector.from_array(raw)
    for z_ptr, l_ptr in ((bad.ptr, good.ptr), (good.ptr, bad.ptr)):
        rounds = _native.P4Rounds(ctx, g._table, 1, 1, z_ptr, l_ptr)
        with pytest.raises(_native.VmpcError) as e:
            rounds.round(None)
        assert e.value.code == _native.E_NONCANON
        rounds.close()
