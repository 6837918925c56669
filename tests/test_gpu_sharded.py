"""Sharded compact prover (verifiable_mpc_amd/sharded.py) on one GPU.

The G blocks of g_hat are held by one process (loopback): per-block partial commitments, the
rank-0-adds-k rule and the rank-ordered combine are the same code as in the multi-process setting,
only the all-gather is missing.  The proof must be the very proof of the unsharded compact prover
(same transcript), and verify with the ordinary verifier.  world = 1 with a real process group
(RCCL) is covered by bench.py --force-collective.
"""
import random

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


def _flat(proof):
    return {k: (tuple(v.normalize().coords) if hasattr(v, "normalize") else
                [int(e) for e in v] if isinstance(v, list) else int(v)) for k, v in proof.items()}


@pytest.mark.parametrize("N,world", [(128, 1), (128, 2), (256, 4), (1024, 2)])
def test_sharded_prover_equals_unsharded(vm, N, world):
    from verifiable_mpc_amd import _native, sharded
    rng = random.Random(N * 10 + world)
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
    gens = {"g": g, "h": h, "k": k}
    xs, Lf = vm.ScalarVector.from_ints(x), vm.pivot.LinearForm(vm.ScalarVector.from_ints(coeffs))
    P = vm.pivot.vector_commitment(xs, gamma, g, h)
    y = gf(Lf(xs))
    want = vm.compressed_pivot.protocol_5_prover(gens, P, Lf, y, xs, gamma, gf, transcript="compact",
                                                 r=list(r), rho=rho)

    crs = sharded.ShardedCrs.from_exponents(h, k, _native.ints_to_array(exps, 32), world, range(world))
    assert crs.digest() == vm.compressed_pivot.generators_digest(gens)
    got = sharded.protocol_5_prover(crs, P, Lf, y, xs, gamma, gf, vm.ScalarVector.from_ints(r), rho)
    assert _flat(got) == _flat(want)
    assert vm.compressed_pivot.protocol_5_verifier(gens, P, Lf, y, got, gf, transcript="compact") is True
    # a commitment over the blocks equals the plain one, with and without the k term
    v = vm.ScalarVector.from_ints([rng.randrange(ELL) for _ in range(N)])
    for gk in (None, 12345):
        c = crs.commit([(v, gk)])[0]
        ref = vm.pivot.vector_commitment(v, gk or 0, g + [h], k)
        assert tuple(c.normalize().coords) == tuple(ref.normalize().coords)


def test_sharded_commitment_with_scalars_the_short_path_cannot_hold(vm):
    """per-rank 16-row tables of <= 2^17 columns are what the fused short path (csrc/msm_short.hip) takes,
    and a vector of small values overflows its bins - an answer (VMPC_E_AGAIN at a later sync, the partial all zeros)
    that the exchange after the partial sums cannot act on.  The sharded commitments take the general path: the
    result is right and no E_AGAIN is left behind on the context."""
    from verifiable_mpc_amd import _native, sharded
    N, world = 1 << 16, 2
    rng = np.random.default_rng(16)
    exps = rng.integers(0, 256, size=(N - 1, 32), dtype=np.uint8)
    exps[:, 31] &= 0x0F
    group = vm.EllipticCurve("Ed25519", "projective")
    h, k = group.generator, vm.Ed25519Point.repeat(group.generator, 777)
    crs = sharded.ShardedCrs.from_exponents(h, k, exps, world, range(world))
    assert crs.shards[0].points._table.rows == 16
    ones = vm.ScalarVector.from_ints([1] * N)                    # every non-zero digit in bin 0
    tot = (sum(_native.array_to_ints(exps)) + 1) % ELL           # ... + 1 * h
    for gk in (None, 5):
        c = crs.commit([(ones, gk)])[0]
        assert c == vm.Ed25519Point.repeat(group.generator, (tot + 777 * (gk or 0)) % ELL)
    crs.ctx.sync()                                               # would raise VMPC_E_AGAIN had a short pass overflowed
    assert crs.ctx.get_short_path() is True
