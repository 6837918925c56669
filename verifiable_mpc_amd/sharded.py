"""Compact-transcript Protocol 5 prover with the generators sharded over GPUs.

What shards is the group work AND the O(N)-per-round scalar work that feeds it.  `g_hat = g || h`
(N = 2^k points) is cut into G blocks, one per rank (one process per GPU).  For every commitment of the
proof a rank computes the partial sum over ITS block on its own fixed-base table
(include/vmpc.h vmpc_msm_table_dev), with scalars it derives for its block only
(vmpc_fr_tail_scalars_block_dev).  The only exchange is an all-gather of 128-byte extended points - one
for the announcement A, one per round carrying A_i and B_i together (compressed_pivot.py:110, :41-42) -
after which every rank adds the G partial points in rank order with the same device routine, so all ranks
hold bit-identical A_i, B_i, derive the same challenges, and need no broadcast.  The generators are never
folded (compressed_pivot._tabulated explains why that is the cheaper form even on one GPU): round i
commits to the UNFOLDED block with the pending challenge products multiplied into the scalars, so no
generator ever crosses a link.

Blocks are CONTIGUOUS ([r N/G, (r+1) N/G)), not the cyclic i mod G that SURVEY.md 8e sketched: cyclic
sharding keeps a FOLD local (index i pairs with i + half), but this prover does not fold - and contiguous
blocks are what make the CRS digest (whole 4096-byte chunks per rank) and the per-block scalar kernel
(one index range) local.  The cyclic layout remains what parallel.ShardedMsm uses for a single commitment.

What stays replicated: the witness-side vectors z_hat and L~ (32 bytes per entry) and their folds
z' = z_l + c z_r, L' = c L_l + L_r - together 2N elements over the whole proof, against 3N per ROUND for
the per-generator scalars that are now block-local.

The proof is the same dict, with the same values, as
compressed_pivot.protocol_5_prover(..., transcript="compact") and verifies with
compressed_pivot.protocol_5_verifier.

With a `comm` (verifiable_mpc_amd._native.Comm: RCCL inside a node, or a callback transport) the rounds run through
the device-resident round context of the single-GPU prover in its sharded form (include/vmpc.h
vmpc_p4_create_sharded, csrc/prover.hip): one C call for all rounds, the exchange enqueued on the context's own
stream right behind the partial sums, and the fold of the generators after five rounds done block-locally - a block
holds 2^5 / G whole strides of the fold, so a rank folds its strides into a partial vector and nothing crosses a
link (needs 2^5 >= 2 G and blocks of >= 2^18 generators; otherwise the rounds stay on the block's table).  On one
rank that is exactly the unsharded prover plus 20 all-gathers.  Without a comm (the `ops` stand-ins of the CPU
tests, blocks held in one process) the rounds are driven from here, one exchange each.

Everything that touches a device goes through a small `ops` object (`DeviceOps` below), so that the
host logic - block arithmetic, which rank adds the k term, gather layout, rank-ordered combine, challenge
derivation - also runs under `gloo` with world_size 2 on CPU, where the test supplies host `ops` built
on the oracle (tests/test_sharded_gloo.py).  `ShardedCrs` can also hold ALL G blocks in one process
(`loopback`): that is how the block arithmetic is tested on a single-GPU box (tests/test_gpu_sharded.py).
"""
import hashlib

import numpy as np

from . import compressed_pivot as cp
from . import pivot
from .device import DeviceScalar, PointVector, ScalarVector, get_context, reduce_scalar
from .groups import Ed25519Point


class DeviceOps:
    """The device side of the sharded prover on this process's GPU (csrc/msm.hip, frvec.hip, sha256.hip)."""

    def __init__(self, ctx=None):
        self.ctx = ctx or get_context()

    # -- CRS blocks ---------------------------------------------------------------------------------
    def make_block(self, h, exponents, append_h, k, rows):
        """points = [h ** e for e in exponents] (+ h itself for the last block), tabulated with extra k"""
        pts = PointVector.fixed_base(h, ScalarVector.from_array(exponents, self.ctx), self.ctx, keep_proj=False)
        if append_h:
            pts = pts.concat([h])
        pts.precompute([k], rows=rows)
        return pts

    def leaf_digests(self, block):
        return block.ctx.sha256_chunks(block.affine_ptr, 64 * len(block), cp.CHUNK)

    # -- replicated scalar vectors ----------------------------------------------------------------------
    def vector(self, v):
        return v if isinstance(v, ScalarVector) else ScalarVector.from_ints([pivot._residue(e) for e in v], self.ctx)

    def form_digest(self, L):
        return cp._form_digest(L)

    def axpy(self, c, x, y):
        return x.axpy(c, y)                # csrc/frvec.hip

    def concat(self, v, tail):
        return v + list(tail)

    def to_field_list(self, v, gf):
        return [gf(x) for x in v.to_ints()]

    # -- one rank's block ------------------------------------------------------------------------------------
    def new_products(self, n_loc):
        return ScalarVector.empty(n_loc, self.ctx)

    def block_scalars(self, newest_challenge, t, log2_n, z_hat, lo, n_loc, products):
        """this block's slice of the round's commitment scalars v_a, v_b (length n_loc each)"""
        v_a, v_b = ScalarVector.empty(n_loc, self.ctx), ScalarVector.empty(n_loc, self.ctx)
        self.ctx.fr_tail_scalars_block(newest_challenge, t, log2_n, z_hat.ptr, lo, n_loc, products.ptr,
                                       v_a.ptr, v_b.ptr)
        return v_a, v_b

    def block_slice(self, v, lo, n_loc):
        return v[lo:lo + n_loc]

    def partial(self, block, v_block, gamma, out_ptr, stream_index):
        """out (128-byte extended point at device address out_ptr) = <v_block, block> (+ gamma k)"""
        from .device import get_aux_context
        ctx = self.ctx
        if stream_index:
            ctx = get_aux_context()
            ctx.wait_for(self.ctx)
        esc = ctx.upload(np.zeros(32, np.uint8))
        if isinstance(gamma, DeviceScalar):
            ctx.copy(esc.ptr, gamma.ptr, 32)
        elif gamma is not None:
            ctx.upload_into(esc.ptr, np.frombuffer(reduce_scalar(gamma).to_bytes(32, "little"), np.uint8))
        t = block._table
        # general path only: the fused short path's "repeat this call" (VMPC_E_AGAIN at a later synchronisation) is
        # an answer the exchange that follows - it has summed the void partial on every rank by then - cannot act on
        ctx.on_general_path(lambda: ctx.msm_table(t.ptr, len(block), 1, v_block.ptr, len(block), esc.ptr, out_ptr, None,
                                                  rows=t.rows))
        return ctx, (esc, v_block)          # stream to wait for, buffers to keep alive until then

    # -- exchange buffers and the ordered combine ----------------------------------------------------------------
    def buffers(self, W, K, torch):
        """(mine, gathered, address of mine, address of gathered): K points from each of W ranks"""
        if torch is None:                   # loopback: the blocks' partial points land directly in `gathered`
            g = self.ctx.alloc(128 * W * K)
            return None, g, None, g.ptr
        # torch.empty: no fill kernel on torch's stream that the vmpc streams (non-blocking, unordered
        # with it) could race with; every byte is written by the partial sums / the all-gather
        mine = torch.empty(128 * K, dtype=torch.uint8, device="cuda")
        gathered = torch.empty(128 * K * W, dtype=torch.uint8, device="cuda")
        return mine, gathered, mine.data_ptr(), gathered.data_ptr()

    def wait_collective(self, torch):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        ev.synchronize()

    def combine(self, gathered_ptr, W, K):
        """K sums of W points each, added in rank order: the same bits on every rank"""
        res = self.ctx.alloc(128 * K)
        self.ctx.points_sum_many(gathered_ptr, W, K, res.ptr, None)
        self.ctx.sync()
        raw = self.ctx.download(res.ptr, 128 * K).tobytes()
        return [Ed25519Point.from_proj_bytes(raw[128 * j:128 * j + 96]).normalize() for j in range(K)]


class CrsShard:
    """Block `index` of g_hat: points [lo, lo + n) (tabulated with extra k when on a device)."""

    def __init__(self, index, lo, points):
        self.index, self.lo, self.n = index, lo, len(points)
        self.points = points


class ShardedCrs:
    """g_hat = g || h in `world` blocks.  `shards` holds this process's blocks: one in the
    multi-process setting (`dist` given), all of them in loopback."""

    def __init__(self, N, world, shards, h, k, dist=None, torch=None, ctx=None, ops=None, comm=None):
        assert N & (N - 1) == 0 and N % world == 0 and N // world >= 64, "block = whole digest chunks"
        self.N, self.world, self.shards = N, world, sorted(shards, key=lambda s: s.index)
        self.h, self.k = h, k
        self.dist, self.torch = dist, torch
        self.ops = ops or DeviceOps(ctx)
        self.ctx = getattr(self.ops, "ctx", None)
        # comm: the exchange runs inside the C library (one block per process, any transport); dist: through
        # torch.distributed from here; neither: every block lives in this process
        self.comm = comm
        assert comm is None or (comm.world == world and len(shards) == 1 and shards[0].index == comm.rank)
        self.loopback = dist is None and comm is None
        if self.loopback:
            assert [s.index for s in self.shards] == list(range(world)), "loopback holds every block"
        else:
            assert len(self.shards) == 1
        for s in self.shards:
            assert s.lo == s.index * (N // world) and s.n == N // world
        self._digest = None

    # -- construction from the exponents (tests / bench: g_i = r_i * h as create_generators does) ----
    @classmethod
    def from_exponents(cls, h, k, exponents, world, ranks, dist=None, torch=None, ctx=None, rows=None, ops=None,
                       comm=None):
        """exponents: (N - 1, 32) uint8 array (the same on every rank); `ranks`: the blocks to build"""
        ops = ops or DeviceOps(ctx)
        N = len(exponents) + 1
        n_loc = N // world
        shards = []
        for r in ranks:
            lo, hi = r * n_loc, (r + 1) * n_loc
            # the last block ends with h itself (g_hat = g || h, compressed_pivot.py:138)
            shards.append(CrsShard(r, lo, ops.make_block(h, exponents[lo:min(hi, N - 1)], hi == N, k, rows)))
        return cls(N, world, shards, h, k, dist, torch, ctx, ops, comm)

    # -- compact CRS digest, identical to compressed_pivot.generators_digest(g, h, k) ---------------------
    def digest(self):
        if self._digest is None:
            local = {s.index: self.ops.leaf_digests(s.points) for s in self.shards}
            if self.loopback:
                blocks = [local[i] for i in range(self.world)]
            elif self.comm is not None:
                mine = local[self.shards[0].index]
                src = self.ctx.upload(np.frombuffer(mine, np.uint8))
                dst = self.ctx.alloc(len(mine) * self.world)
                self.comm.allgather(self.ctx, src.ptr, dst.ptr, len(mine))
                blocks = [self.ctx.download(dst.ptr, len(mine) * self.world).tobytes()]
            else:
                gathered = [None] * self.world
                self.dist.all_gather_object(gathered, local[self.shards[0].index])
                blocks = gathered
            leaves = b"".join(blocks) + hashlib.sha256(self.k.to_affine_bytes()).digest()
            nbytes = 64 * (self.N + 1)
            self._digest = hashlib.sha256(b"vmpc-ac20/gens/v1" + nbytes.to_bytes(8, "little") + leaves).digest()
        return self._digest

    # -- commitments ------------------------------------------------------------------------------------
    def commit_blocks(self, items):
        """items: [(per_shard, gamma)]: per_shard maps a shard index to that block's scalar vector (length
        N / world), gamma is the exponent of k (None = no k term; added by block 0's owner only).  Returns
        the commitments, identical on every rank.  One exchange for all of them: every rank contributes
        len(items) partial points of 128 bytes."""
        ops, W, K = self.ops, self.world, len(items)
        if self.comm is not None:
            return self._commit_blocks_comm(items)
        mine, gathered, mine_ptr, gathered_ptr = ops.buffers(W, K, None if self.loopback else self.torch)
        pending = []
        try:
            for j, (per_shard, gamma) in enumerate(items):
                for s in self.shards:
                    dst = (gathered_ptr + 128 * (s.index * K + j)) if self.loopback else (mine_ptr + 128 * j)
                    # A_i and B_i of a round run side by side on two streams
                    pending.append(ops.partial(s.points, per_shard[s.index], gamma if s.index == 0 else None,
                                               dst, j % 2))
        finally:
            for stream, _ in pending:       # partial points complete (also before buffers are dropped on error)
                if stream is not None:
                    stream.sync()
        if not self.loopback:
            self.dist.all_gather_into_tensor(gathered, mine)      # the exchange: W x K x 128 bytes
            ops.wait_collective(self.torch)
        out = ops.combine(gathered_ptr, W, K)
        del pending
        return out

    def _commit_blocks_comm(self, items):
        """the same through the C library's exchange: partial sums, all-gather and rank-ordered add are enqueued
        back to back on the context's stream; the host waits once, for the result"""
        ctx, W, K, s = self.ctx, self.world, len(items), self.shards[0]
        t = s.points._table
        mine, scratch, res = ctx.alloc(128 * K), ctx.alloc(128 * K * W), ctx.alloc(128 * K)
        keep = []
        for j, (per_shard, gamma) in enumerate(items):
            esc = ctx.upload(np.zeros(32, np.uint8))
            if s.index == 0 and isinstance(gamma, DeviceScalar):
                ctx.copy(esc.ptr, gamma.ptr, 32)
            elif s.index == 0 and gamma is not None:
                ctx.upload_into(esc.ptr, np.frombuffer(reduce_scalar(gamma).to_bytes(32, "little"), np.uint8))
            ctx.on_general_path(lambda: ctx.msm_table(t.ptr, s.n, 1, per_shard[s.index].ptr, s.n, esc.ptr,      # (as in
                                                      mine.ptr + 128 * j, None, rows=t.rows))                  # partial)
            keep.append(esc)
        self.comm.points_allsum(ctx, mine.ptr, K, scratch.ptr, res.ptr)
        raw = ctx.download(res.ptr, 128 * K).tobytes()
        return [Ed25519Point.from_proj_bytes(raw[128 * j:128 * j + 96]).normalize() for j in range(K)]

    def commit(self, items):
        """items: [(v, gamma)] with v a full-length (N) scalar vector over g_hat"""
        n_loc = self.N // self.world
        for v, _ in items:
            assert len(v) == self.N
        return self.commit_blocks([({s.index: self.ops.block_slice(v, s.lo, n_loc) for s in self.shards}, gamma)
                                   for v, gamma in items])


def protocol_5_prover(crs, P, L, y, x, gamma, gf, r, rho):
    """compressed_pivot.protocol_5_prover (compressed_pivot.py:89-145) with transcript="compact",
    the group work sharded over `crs`.  x, r: the witness and its masks (length N - 1), rho: int."""
    order = gf.order
    ops = crs.ops
    n = len(x)
    assert n + 1 == crs.N, "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    L, y = pivot.affine_to_linear(L, y, n)
    x, r = ops.vector(x), ops.vector(r)
    L = pivot.AffineForm(ops.vector(L.coeffs), L.constant)
    proof = {}
    t = L(r)
    if isinstance(t, int):
        t = gf(t)
    A = crs.commit([(ops.concat(r, [rho]), None)])[0]            # sum r_i g_i + rho h
    proof["t"], proof["A"] = t, A
    P = cp._pt(P)
    seed = hashlib.sha256(b"vmpc-ac20/p5/v1" + crs.digest() + ops.form_digest(L) + P.to_affine_bytes()
                          + cp._sc_bytes(pivot._residue(y)) + cp._sc_bytes(pivot._residue(t))
                          + A.to_affine_bytes()).digest()
    c0 = cp._challenge(hashlib.sha256(seed + b"\x00").digest(), order)
    c1 = cp._challenge(hashlib.sha256(seed + b"\x01").digest(), order)
    z_hat = ops.concat(ops.axpy(c0, x, r), [gf(c0 * gamma + rho)])
    L_tilde = cp._extend_form(L, c1)
    transcript = cp._p5_setup(None, crs.k, seed, "compact", order)

    if crs.comm is not None and cp.NATIVE_ROUNDS and isinstance(ops, DeviceOps):
        # all rounds in one C call on this rank's block (vmpc_p4_create_sharded): identical A_i, B_i and challenges
        # on every rank, one exchange per round inside the library
        from ._native import P4Rounds
        table = crs.shards[0].points._table
        rounds = P4Rounds(crs.ctx, table, 0, table.extra_index(crs.k), z_hat.ptr, cp._coeffs_dev(L_tilde).ptr,
                          n_total=crs.N, comm=crs.comm)
        try:
            n_rounds = crs.N.bit_length() - 2
            _, pairs, z_prime = rounds.run_compact(transcript.state, 0, n_rounds)
        finally:
            rounds.close()
        for i, (a, b) in enumerate(pairs):
            proof["A" + str(i)] = Ed25519Point.from_affine_bytes(a)
            proof["B" + str(i)] = Ed25519Point.from_affine_bytes(b)
        proof["z_prime"] = [gf(v) for v in z_prime]
        return proof

    log2_n = crs.N.bit_length() - 1
    n_loc = crs.N // crs.world
    challenges, round_i = [], 0
    products = {s.index: ops.new_products(n_loc) for s in crs.shards}     # challenge products per generator
    while True:
        half = len(z_hat) // 2
        z_l, z_r, gamma_a, gamma_b = cp._round_prover_scalars(L_tilde, z_hat, half, gf)
        v_a, v_b = {}, {}
        for s in crs.shards:                 # scalar work proportional to the block, not to N
            v_a[s.index], v_b[s.index] = ops.block_scalars(challenges[-1] if challenges else 0, len(challenges),
                                                           log2_n, z_hat, s.lo, n_loc, products[s.index])
        A_i, B_i = crs.commit_blocks([(v_a, gamma_a), (v_b, gamma_b)])
        proof["A" + str(round_i)], proof["B" + str(round_i)] = A_i, B_i
        c = transcript.round_challenge(round_i, A_i, B_i, None, crs.k, None, None)
        challenges.append(c)
        L_tilde = cp._fold_form(L_tilde, c, half, gf)
        z_hat = cp._fold_witness(z_l, z_r, c, half)
        if len(z_hat) <= 2:
            proof["z_prime"] = ops.to_field_list(z_hat, gf)
            return proof
        round_i += 1
