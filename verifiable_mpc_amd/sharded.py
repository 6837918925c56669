"""Compact-transcript Protocol 5 prover with the generators sharded over GPUs.

What shards is the group work.  `g_hat = g || h` (N = 2^k points) is cut into G contiguous blocks,
one per rank (one process per GPU); every rank keeps the scalar vectors whole - x, r, L are
32 bytes per entry and their folds are O(N) - and computes, for every commitment of the proof,
the partial sum over ITS block on its own fixed-base table (include/vmpc.h vmpc_msm_table_dev).
The only exchange is an all-gather of 128-byte extended points - one for the announcement A,
one per round carrying A_i and B_i together (compressed_pivot.py:110, :41-42) - after which every rank adds the
G partial points in rank order with the same device routine, so all ranks hold bit-identical
A_i, B_i, derive the same challenges, and need no broadcast.  The generators are never folded
(compressed_pivot._tabulated explains why that is the cheaper form even on one GPU): round i
commits to the UNFOLDED block with the pending challenge products multiplied into the scalars
(csrc/frvec.hip k_fr_tail_scalars), so no generator ever crosses a link.

The proof is the same dict, with the same values, as
compressed_pivot.protocol_5_prover(..., transcript="compact") and verifies with
compressed_pivot.protocol_5_verifier.

`ShardedCrs` can also hold ALL G blocks in one process (`loopback`): the partial sums are then
computed one block after the other on the same GPU.  That is how the block arithmetic (offsets,
which rank adds the k term, rank-ordered combine) is tested on a single-GPU box
(tests/test_gpu_sharded.py); the multi-process path differs only in where the other ranks'
partial points come from.
"""
import hashlib

import numpy as np

from . import compressed_pivot as cp
from . import pivot
from .device import DeviceScalar, PointVector, ScalarVector, get_context, reduce_scalar
from .groups import Ed25519Point


class CrsShard:
    """Block `index` of g_hat on this GPU: points [lo, lo + n) with their fixed-base table
    (extras: k)."""

    def __init__(self, index, lo, points, k, rows=None):
        assert isinstance(points, PointVector)
        self.index, self.lo, self.n = index, lo, len(points)
        self.points = points
        points.precompute([k], rows=rows)
        self.table = points._table

    def leaf_digests(self):
        """SHA-256 of every 4096-byte chunk of this block's affine bytes (compact CRS digest)"""
        return self.points.ctx.sha256_chunks(self.points.affine_ptr, 64 * self.n, cp.CHUNK)

    def partial(self, ctx, v, gamma, out_ext_ptr):
        """out = sum_i v[lo + i] * g_hat[lo + i]  (+ gamma * k when gamma is not None)"""
        esc = ctx.upload(np.zeros(32, np.uint8))
        if isinstance(gamma, DeviceScalar):
            ctx.copy(esc.ptr, gamma.ptr, 32)
        elif gamma is not None:
            ctx.upload_into(esc.ptr, np.frombuffer(reduce_scalar(gamma).to_bytes(32, "little"), np.uint8))
        ctx.msm_table(self.table.ptr, self.n, 1, v.ptr + 32 * self.lo, self.n, esc.ptr, out_ext_ptr, None,
                      rows=self.table.rows)
        return esc          # keep alive until the stream is done with it


class ShardedCrs:
    """g_hat = g || h in `world` blocks.  `shards` holds this process's blocks: one in the
    multi-process setting (`dist` given), all of them in loopback."""

    def __init__(self, N, world, shards, h, k, dist=None, torch=None, ctx=None):
        assert N & (N - 1) == 0 and N % world == 0 and N // world >= 64, "block = whole digest chunks"
        self.N, self.world, self.shards = N, world, sorted(shards, key=lambda s: s.index)
        self.h, self.k = h, k
        self.dist, self.torch = dist, torch
        self.ctx = ctx or get_context()
        self.loopback = dist is None
        if self.loopback:
            assert [s.index for s in self.shards] == list(range(world)), "loopback holds every block"
        else:
            assert len(self.shards) == 1
        for s in self.shards:
            assert s.lo == s.index * (N // world) and s.n == N // world
        self._digest = None

    # -- construction from the exponents (tests / bench: g_i = r_i * h as create_generators does) ----
    @classmethod
    def from_exponents(cls, h, k, exponents, world, ranks, dist=None, torch=None, ctx=None, rows=None):
        """exponents: (N - 1, 32) uint8 array (the same on every rank); `ranks`: the blocks to build"""
        ctx = ctx or get_context()
        N = len(exponents) + 1
        n_loc = N // world
        shards = []
        for r in ranks:
            lo, hi = r * n_loc, (r + 1) * n_loc
            sl = exponents[lo:min(hi, N - 1)]
            pts = PointVector.fixed_base(h, ScalarVector.from_array(sl, ctx), ctx, keep_proj=False)
            if hi == N:                               # the last block ends with h itself
                pts = pts.concat([h])
            shards.append(CrsShard(r, lo, pts, k, rows))
        return cls(N, world, shards, h, k, dist, torch, ctx)

    # -- compact CRS digest, identical to compressed_pivot.generators_digest(g, h, k) ---------------------
    def digest(self):
        if self._digest is None:
            local = {s.index: s.leaf_digests() for s in self.shards}
            if self.loopback:
                blocks = [local[i] for i in range(self.world)]
            else:
                gathered = [None] * self.world
                self.dist.all_gather_object(gathered, local[self.shards[0].index])
                blocks = gathered
            leaves = b"".join(blocks) + hashlib.sha256(self.k.to_affine_bytes()).digest()
            nbytes = 64 * (self.N + 1)
            self._digest = hashlib.sha256(b"vmpc-ac20/gens/v1" + nbytes.to_bytes(8, "little") + leaves).digest()
        return self._digest

    # -- commitments ------------------------------------------------------------------------------------
    def commit(self, items):
        """items: [(v, gamma)] with v a ScalarVector of length N over g_hat and gamma the exponent of k
        (None = no k term).  Returns the commitments, identical on every rank.  One exchange for all of
        them: every rank contributes len(items) partial points (128 bytes each)."""
        from .device import get_aux_context
        ctx, W, K = self.ctx, self.world, len(items)
        # partial points, laid out [rank][item] as the all-gather will produce them
        if self.loopback:
            gathered, gptr, mine = ctx.alloc(128 * W * K), None, None
            gptr = gathered.ptr
        else:
            # torch.empty: no fill kernel on torch's stream that the vmpc streams (non-blocking, unordered
            # with it) could race with; every byte is written by the partial sums / the all-gather
            mine = self.torch.empty(128 * K, dtype=self.torch.uint8, device="cuda")
            gathered = self.torch.empty(128 * K * W, dtype=self.torch.uint8, device="cuda")
            gptr = gathered.data_ptr()
        keep, used = [], [ctx]
        for j, (v, gamma) in enumerate(items):
            assert len(v) == self.N
            cctx = ctx
            if j % 2 == 1:                # A_i and B_i of a round run side by side on two streams
                cctx = get_aux_context()
                cctx.wait_for(ctx)
                used.append(cctx)
            for s in self.shards:
                dst = (gptr + 128 * (s.index * K + j)) if self.loopback else (mine.data_ptr() + 128 * j)
                keep.append(s.partial(cctx, v, gamma if s.index == 0 else None, dst))
        for c in used:
            c.sync()                                              # partial points are complete
        if not self.loopback:
            self.dist.all_gather_into_tensor(gathered, mine)      # the exchange: W x K x 128 bytes
            ev = self.torch.cuda.Event()
            ev.record(self.torch.cuda.current_stream())
            ev.synchronize()
        res = ctx.alloc(128 * K)
        ctx.points_sum_many(gptr, W, K, res.ptr, None)            # rank order: same bits everywhere
        ctx.sync()
        raw = ctx.download(res.ptr, 128 * K).tobytes()
        del keep
        return [Ed25519Point.from_proj_bytes(raw[128 * j:128 * j + 96]).normalize() for j in range(K)]


def protocol_5_prover(crs, P, L, y, x, gamma, gf, r, rho):
    """compressed_pivot.protocol_5_prover (compressed_pivot.py:89-145) with transcript="compact",
    the group work sharded over `crs`.  x, r: the witness and its masks (length N - 1), rho: int."""
    order = gf.order
    n = len(x)
    assert n + 1 == crs.N, "This implementation requires n+1 to be power of 2 (else, use padding with zeros)."
    ctx = crs.ctx
    L, y = pivot.affine_to_linear(L, y, n)
    x, r = pivot._as_device(x), pivot._as_device(r)
    L = pivot.AffineForm(cp._coeffs_dev(L), L.constant)
    proof = {}
    t = L(r)
    if isinstance(t, int):
        t = gf(t)
    A = crs.commit([(r.concat([rho]), None)])[0]                 # sum r_i g_i + rho h
    proof["t"], proof["A"] = t, A
    P = cp._pt(P)
    seed = hashlib.sha256(b"vmpc-ac20/p5/v1" + crs.digest() + cp._form_digest(L) + P.to_affine_bytes()
                          + cp._sc_bytes(pivot._residue(y)) + cp._sc_bytes(pivot._residue(t))
                          + A.to_affine_bytes()).digest()
    c0 = cp._challenge(hashlib.sha256(seed + b"\x00").digest(), order)
    c1 = cp._challenge(hashlib.sha256(seed + b"\x01").digest(), order)
    z_hat = x.axpy(c0, r).concat([gf(c0 * gamma + rho)])
    L_tilde = cp._extend_form(L, c1)
    transcript = cp._p5_setup(None, crs.k, seed, "compact", order)

    log2_n = crs.N.bit_length() - 1
    challenges, round_i = [], 0
    products = ScalarVector.empty(crs.N, ctx)                   # challenge products per generator
    while True:
        half = len(z_hat) // 2
        z_l, z_r, gamma_a, gamma_b = cp._round_prover_scalars(L_tilde, z_hat, half, gf)
        v_a, v_b = ScalarVector.empty(crs.N, ctx), ScalarVector.empty(crs.N, ctx)
        ctx.fr_tail_scalars_inc(challenges[-1] if challenges else 0, len(challenges), log2_n, z_hat.ptr,
                                products.ptr, v_a.ptr, v_b.ptr)
        A_i, B_i = crs.commit([(v_a, gamma_a), (v_b, gamma_b)])
        proof["A" + str(round_i)], proof["B" + str(round_i)] = A_i, B_i
        c = transcript.round_challenge(round_i, A_i, B_i, None, crs.k, None, None)
        challenges.append(c)
        L_tilde = cp._fold_form(L_tilde, c, half, gf)
        z_hat = cp._fold_witness(z_l, z_r, c, half)
        if len(z_hat) <= 2:
            proof["z_prime"] = [gf(v) for v in z_hat.to_ints()]
            return proof
        round_i += 1
