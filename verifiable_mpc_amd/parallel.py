"""Multi-GPU Pedersen commitment: one process per GPU, cyclic sharding, ONE exchange step.

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
    """Local MSM + ordered combine on the GPU of this process (csrc/msm.hip)."""

    def __init__(self, ctx, torch):
        self.ctx, self.torch = ctx, torch
        self.partial_buf = torch.zeros(128, dtype=torch.uint8, device="cuda")
        self.out_aff = torch.zeros(64, dtype=torch.uint8, device="cuda")

    def partial(self, scalars, points):
        self.ctx.msm(scalars.ptr, points.affine_ptr, len(scalars), None, None, 0,
                     self.partial_buf.data_ptr(), None)
        return self.partial_buf

    def commit_single(self, scalars, points):
        self.ctx.msm(scalars.ptr, points.affine_ptr, len(scalars), None, None, 0, None,
                     self.out_aff.data_ptr())
        return Ed25519Point.from_affine_bytes(self.out_aff.cpu().numpy().tobytes())

    def new_gather_buffer(self, world):
        return self.torch.zeros((world, 128), dtype=self.torch.uint8, device="cuda")

    def combine(self, gathered, world):
        self.ctx.points_sum(gathered.data_ptr(), world, None, self.out_aff.data_ptr())
        return Ed25519Point.from_affine_bytes(self.out_aff.cpu().numpy().tobytes())


class ShardedMsm:
    """commit(scalars_shard, points_shard) -> the commitment over ALL ranks' shards."""

    def __init__(self, ctx, world, rank, dist=None, torch=None, backend=None):
        self.world, self.rank, self.dist = world, rank, dist
        self.backend = backend if backend is not None else HipBackend(ctx, torch)
        self.gathered = self.backend.new_gather_buffer(world) if world > 1 else None

    def commit(self, scalars, points):
        if self.world == 1:
            return self.backend.commit_single(scalars, points)
        mine = self.backend.partial(scalars, points)
        # the single curve-point exchange: G x 128 B
        self.dist.all_gather_into_tensor(self.gathered.view(-1), mine)
        return self.backend.combine(self.gathered, self.world)
