"""Multi-GPU Pedersen commitment: one process per GPU, cyclic sharding, ONE exchange step.

The exchange itself lives in the C library (include/vmpc.h, vmpc_comm_*: ncclAllGather on the MSM's own
stream + rank-ordered add); torch.distributed only bootstraps it (make_comm) - or carries the bytes when no
`comm` is given.

An MSM is a sum of independent terms, so the generator / scalar vectors are sharded
cyclically by index (rank r owns i = r mod G) and each rank runs the local
Pippenger MSM on its shard.  The only data-path collective is an all-gather of the G partial
points (128-byte extended coordinates) over RCCL/xGMI; RCCL has no user-defined reduction, so
every rank then adds the G points IN RANK ORDER with the same device routine
(vmpc_points_sum_dev), which makes the result bit-identical on all ranks.  The message is
128 B per rank: latency-bound, far below one xGMI link.

The compute and the exchange are behind a small backend interface so that the host logic
(sharding, gather, ordered combine) is covered by world_size-2 gloo tests on CPU
(tests/test_parallel_gloo.py) with the oracle standing in for the kernels.
"""
import numpy as np

from .groups import Ed25519Point


def cyclic_indices(n_total, world, rank):
    """Indices owned by `rank` under cyclic sharding."""
    return np.arange(rank, n_total, world)


def shard_rows(arr, world, rank):
    """Rows of a (n, width) array owned by `rank` (cyclic)."""
    return np.ascontiguousarray(arr[rank::world])


class HipBackend:
    """Local MSM + ordered combine on the GPU of this process (csrc/msm.hip).

    Several slots (vmpc contexts = streams + workspaces on the same GPU) let consecutive
    commitments overlap: the latency-bound tail of one MSM (bucket reduction, Horner chain)
    runs next to the throughput-bound head of the next.

    comm (verifiable_mpc_amd._native.Comm): the exchange and the rank-ordered add are enqueued by the C library
    on the slot's own stream right behind the partial sum (vmpc_comm_points_allsum_dev) - no host
    synchronisation between the MSM and the all-gather, and the slots' exchanges overlap like their MSMs.
    Without it the exchange goes through torch.distributed (ShardedMsm.finish)."""

    def __init__(self, ctx, torch=None, n_slots=None, comm=None):
        import os
        from .device import get_aux_context
        if n_slots is None:
            n_slots = int(os.environ.get("VMPC_MSM_SLOTS", "3"))
        # comm: one communicator for every slot, or a list with one PER SLOT (make_comms) - each slot's exchange
        # then has a communicator and a stream of its own, the shape RCCL is run in everywhere (one communicator
        # per stream); slots beyond the list share its last entry
        self.comms = list(comm) if isinstance(comm, (list, tuple)) else ([comm] if comm is not None else [])
        comm = self.comms[0] if self.comms else None
        self.torch, self.comm = torch, comm
        self.ctxs = [ctx] + [get_aux_context(10 + i) for i in range(max(0, n_slots - 1))]
        # Phase pipelining (include/vmpc.h vmpc_ctx_set_bucket_stream): the slots share one bucket stream, so their
        # bucket kernels run back to back, one at a time, each a persistent launch that leaves register-file room
        # for the sort of the next pass and the reduction / recombination of the previous one.
        # OFF by default (VMPC_EXPERIMENTAL=1 VMPC_BUCKET_STREAM=1 turns it on): measured in round 4 it loses - a bucket kernel confined
        # to 3 (2) workgroups per CU is 21 % (40 %) slower by itself, and the co-runners take their share of the
        # vector ALU on top (profiles/r04_probes/bucket_stream.txt).
        self.bucket_stream = None
        wgs = int(os.environ.get("VMPC_PIPE_BUCKET_WGS_PER_CU", "3"))
        if n_slots > 1 and os.environ.get("VMPC_EXPERIMENTAL", "0") != "0" and os.environ.get("VMPC_BUCKET_STREAM", "0") != "0":
            from ._native import SharedStream
            self.bucket_stream = SharedStream(ctx.device, int(os.environ.get("VMPC_BUCKET_STREAM_PRIORITY", "-1")))
        self.bucket_wgs = wgs
        self._pipelined = False     # set by a driver that keeps several launches in flight (bench.run_steps)
        self.max_batch = 16
        world = comm.world if comm is not None else 1
        if comm is not None or torch is None:
            self._owners = [[c.alloc(128 * self.max_batch), c.alloc(128 * self.max_batch),
                             c.alloc(128 * self.max_batch * world)] for c in self.ctxs]
            self.partial_ptrs = [o[0].ptr for o in self._owners]
            self.combine_ptrs = [o[1].ptr for o in self._owners]
            self.scratch_ptrs = [o[2].ptr for o in self._owners]
            self.partial_bufs = None
        else:
            # torch's fills run on torch's current stream, which the vmpc streams (hipStreamNonBlocking) are
            # not ordered with: finish them before the first MSM writes into these buffers
            self.partial_bufs = [torch.zeros(128 * self.max_batch, dtype=torch.uint8, device="cuda") for _ in self.ctxs]
            self._owners = [torch.zeros(128 * self.max_batch, dtype=torch.uint8, device="cuda") for _ in self.ctxs]
            self.partial_ptrs = [t.data_ptr() for t in self.partial_bufs]
            self.combine_ptrs = [t.data_ptr() for t in self._owners]
            torch.cuda.current_stream().synchronize()
        self.batch_of = [1] * len(self.ctxs)

    @property
    def n_slots(self):
        return len(self.ctxs)

    @property
    def pipelined(self):
        return self._pipelined

    @pipelined.setter
    def pipelined(self, on):
        """A driver that keeps several launches in flight.  Turning it on orders every slot stream ONCE behind what the
        main context's stream holds at this moment (the producers of the inputs); after that launch_partial() skips the
        per-launch wait.  A driver that produces fresh device inputs later says so with inputs_ready()."""
        self._pipelined = bool(on)
        if self._pipelined:
            self.inputs_ready()

    def inputs_ready(self):
        """everything enqueued on the main context's stream so far happens-before the next launch on any slot"""
        for ctx in self.ctxs[1:]:
            ctx.wait_for(self.ctxs[0])

    def launch_partial(self, scalars, points, slot, want_affine):
        # the partial (or, single-GPU, final) sum stays in extended coordinates on the device;
        # normalising one point is O(1) host glue
        ctx = self.ctxs[slot]
        if ctx is not self.ctxs[0] and not self.pipelined:
            # inputs produced on the main context's stream just before this call (a prover's scalars).  A driver
            # that keeps several launches in flight (pipelined) was ordered behind its inputs when it said so
            # (the `pipelined` setter / inputs_ready()) - a wait per launch would chain every slot behind the main
            # slot's commitment IN FLIGHT and march the slots in lockstep (round 4 trace:
            # profiles/r04_probes/timeline_lockstep.txt)
            ctx.wait_for(self.ctxs[0])
        if self.bucket_stream is not None and self.pipelined:
            # for this call only: the contexts also serve callers that run one commitment at a time (the prover),
            # whose bucket stage should have the whole chip
            ctx.set_bucket_stream(self.bucket_stream, self.bucket_wgs)
            try:
                return self._launch_partial(ctx, scalars, points, slot)
            finally:
                ctx.set_bucket_stream(None, 0)
        return self._launch_partial(ctx, scalars, points, slot)

    def _launch_partial(self, ctx, scalars, points, slot):
        # The fused short path (csrc/msm_short.hip) answers scalars beyond its capacities with "repeat this call" at the
        # next synchronisation - an answer a pipeline of launches in flight, let alone one followed by a collective
        # that has already summed the void partial on every rank, cannot act on: this backend's commitments always
        # take the general path.
        ctx.on_general_path(lambda: self._launch_partial_general(ctx, scalars, points, slot))

    def _launch_partial_general(self, ctx, scalars, points, slot):
        table = getattr(points, "_table", None)
        if isinstance(scalars, (list, tuple)):
            # a BATCH of commitments over the same prepared generators in one pass (vmpc_msm_table_batch_dev):
            # the latency chains of the bucket reduction and the recombination are paid once per batch
            assert table is not None and points._table_tail == 0 and 1 <= len(scalars) <= self.max_batch
            self.batch_of[slot] = len(scalars)
            ctx.msm_table_batch(table.ptr, table.n, len(table.extra_bytes), [s.ptr for s in scalars], len(scalars[0]),
                                None, self.partial_ptrs[slot], None, rows=table.rows)
            return
        self.batch_of[slot] = 1
        if table is not None and points._table_tail == 0 and len(scalars) <= table.n:
            # generators held in prepared form (PointVector.precompute): no per-call point preparation
            ctx.msm_table(table.ptr, table.n, len(table.extra_bytes), scalars.ptr, len(scalars), None,
                          self.partial_ptrs[slot], None, rows=table.rows)
            return
        ctx.msm(scalars.ptr, points.affine_ptr, len(scalars), None, None, 0, self.partial_ptrs[slot], None)

    def enqueue_allsum(self, slot):
        """the single curve-point exchange + ordered add of this slot's batch, behind its MSM on its stream"""
        comm = self.comms[min(slot, len(self.comms) - 1)]
        comm.points_allsum(self.ctxs[slot], self.partial_ptrs[slot], self.batch_of[slot], self.scratch_ptrs[slot],
                           self.combine_ptrs[slot])

    def allsum_result(self, slot):
        self.ctxs[slot].sync()
        return self._points(self.ctxs[slot], self.combine_ptrs[slot], self.batch_of[slot])

    def wait(self, slot):
        self.ctxs[slot].sync()

    def ready(self, slot):
        return self.ctxs[slot].done()

    def _points(self, ctx, ptr, b):
        raw = ctx.download(ptr, 128 * b).tobytes()
        pts = [Ed25519Point.from_proj_bytes(raw[128 * k:128 * k + 96]).normalize() for k in range(b)]
        return pts if b > 1 else pts[0]

    def affine_result(self, slot):
        self.ctxs[slot].sync()
        return self._points(self.ctxs[slot], self.partial_ptrs[slot], self.batch_of[slot])

    def partial_tensor(self, slot):
        return self.partial_bufs[slot][:128 * self.batch_of[slot]]

    def new_gather_buffer(self, world):
        if self.torch is None or self.comm is not None:
            return None
        buf = self.torch.zeros((world, 128 * self.max_batch), dtype=self.torch.uint8, device="cuda")
        self.torch.cuda.current_stream().synchronize()
        return buf

    def combine(self, gathered, world, slot=0):
        # on the slot's own (now idle) stream, into its own buffer: never queued behind or
        # overwriting another commitment in flight.  `gathered` holds world x batch points (rank-major): one
        # rank-ordered sum per commitment of the batch
        ctx, b = self.ctxs[slot], self.batch_of[slot]
        ctx.points_sum_many(gathered.data_ptr(), world, b, self.combine_ptrs[slot], None)
        ctx.sync()
        return self._points(ctx, self.combine_ptrs[slot], b)


class ShardedMsm:
    """commit(scalars_shard, points_shard) -> the commitment over ALL ranks' shards.
    launch()/finish() split the call so that up to `n_slots` commitments are in flight.

    Exchange, in order of preference: `comm` (the C library's own: RCCL on the MSM's stream, or a callback
    transport) - enqueued at launch(); else torch.distributed (`dist`) at finish().  Every rank must launch and
    finish in the same order."""

    def __init__(self, ctx, world, rank, dist=None, torch=None, backend=None, force_collective=False, comm=None):
        first = comm[0] if isinstance(comm, (list, tuple)) else comm
        self.world, self.rank, self.dist, self.comm = world, rank, dist, first
        assert first is None or (first.world == world and first.rank == rank)
        self.backend = backend if backend is not None else HipBackend(ctx, torch, comm=comm)
        self.collective = world > 1 or force_collective
        # wait for the collective only: RCCL runs on torch's current stream; a device-wide
        # synchronize here would also drain the other commitments in flight on their own streams
        self.sync_device = None
        if torch is not None and backend is None and first is None:
            def _wait_collective():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                ev.synchronize()
            self.sync_device = _wait_collective
        self.gathered = self.backend.new_gather_buffer(world) if self.collective and first is None else None

    @property
    def n_slots(self):
        return getattr(self.backend, "n_slots", 1)

    def launch(self, scalars, points, slot=0):
        self.backend.launch_partial(scalars, points, slot, want_affine=not self.collective)
        if self.collective and self.comm is not None:
            self.backend.enqueue_allsum(slot)
        return slot

    def finish(self, slot):
        if not self.collective:
            return self.backend.affine_result(slot)
        if self.comm is not None:
            return self.backend.allsum_result(slot)
        self.backend.wait(slot)                     # partial point is complete
        # the single curve-point exchange: G x 128 B (x the number of commitments of a batch)
        mine = self.backend.partial_tensor(slot)
        self.dist.all_gather_into_tensor(self.gathered.view(-1)[:self.world * mine.numel()], mine)
        if self.sync_device:
            self.sync_device()                      # RCCL ran on torch's stream, the combine runs on ours
        return self.backend.combine(self.gathered, self.world, slot)

    def ready(self, slot):
        """the slot's local MSM has completed (backends without a query are always 'ready': finish() waits)"""
        probe = getattr(self.backend, "ready", None)
        return True if probe is None else probe(slot)

    def commit(self, scalars, points):
        return self.finish(self.launch(scalars, points, 0))


def make_comms(ctx, world, rank, dist, torch=None, transport="rccl", count=1):
    """`count` independent communicators of the same process group (one per commitment slot of HipBackend): every
    rank creates them in the same order, each from its own unique id."""
    return [make_comm(ctx, world, rank, dist, torch, transport) for _ in range(count)]


def make_comm(ctx, world, rank, dist, torch=None, transport="rccl"):
    """The C library's communicator for this process group, bootstrapped over torch.distributed (any backend):
    rank 0 draws the RCCL unique id and broadcasts it (transport "rccl"), or the bytes travel through
    torch.distributed's CPU collectives, staged through host memory (transport "host": gloo - tests, or machines
    without a direct GPU fabric)."""
    from ._native import Comm
    if transport == "rccl":
        box = [Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return Comm.rccl(ctx, box[0], world, rank)
    if transport != "host":
        raise ValueError(f"unknown transport {transport!r}")
    import torch as _torch

    def exchange(mine_ptr, gathered_ptr, nbytes):
        mine = _torch.from_numpy(ctx.download(mine_ptr, nbytes).copy())
        out = _torch.empty(world * nbytes, dtype=_torch.uint8)
        dist.all_gather_into_tensor(out, mine)
        ctx.upload_into(gathered_ptr, out.numpy())
    return Comm.callback(world, rank, exchange)
