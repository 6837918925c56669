"""world_size-2 gloo test (CPU) of the SHARDED PROVER's host logic (verifiable_mpc_amd/sharded.py):
contiguous blocks of g_hat, block-local commitment scalars, which rank adds the k term, the all-gather of
two 128-byte points per rank and round, the rank-ordered combine, the CRS digest assembled from per-rank
leaf digests - everything except the kernels, for which host stand-ins built on the oracle are plugged in
through the prover's `ops` interface.  Every rank must hold the same proof, and it must be the proof of the
oracle's unsharded compact prover (oracle/ac20_ref.py)."""
import hashlib
import os
import random
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ac20_ref as ac      # noqa: E402
from oracle import ed25519_ref as ed   # noqa: E402

ELL = ed.ELL


def ext_bytes(pt):
    x, y, z = pt
    X, Y, Z, T = x * z % ed.P, y * z % ed.P, z * z % ed.P, x * y % ed.P
    return b"".join(v.to_bytes(32, "little") for v in (X, Y, Z, T))


class HostOps:
    """Host stand-ins for sharded.DeviceOps: Python lists of field elements, oracle group arithmetic."""
    ctx = None

    def __init__(self, vm):
        self.vm, self.gf = vm, vm.GF(ELL)
        self._target = self._gathered = None

    def make_block(self, h, exponents, append_h, k, rows):
        pts = [ed.pt_repeat(ed.BASE, int.from_bytes(bytes(e), "little")) for e in exponents]
        if append_h:
            pts.append(h.coords)
        self.k = k.coords
        return pts

    def leaf_digests(self, block):
        raw = b"".join(ed.affine_to_bytes(p) for p in block)
        return b"".join(hashlib.sha256(raw[o:o + 4096]).digest() for o in range(0, len(raw), 4096))

    def vector(self, v):
        return [x if isinstance(x, self.gf) else self.gf(int(x)) for x in v]

    def form_digest(self, L):
        from verifiable_mpc_amd import compressed_pivot as cp
        return cp._form_digest(L)

    def axpy(self, c, x, y):
        return [c * a + b for a, b in zip(x, y)]

    def concat(self, v, tail):
        return list(v) + [t if isinstance(t, self.gf) else self.gf(int(t)) for t in tail]

    def to_field_list(self, v, gf):
        return [gf(int(x)) for x in v]

    def new_products(self, n_loc):
        return [1] * n_loc

    def block_scalars(self, newest_challenge, t, log2_n, z_hat, lo, n_loc, products):
        """the definition the kernel k_fr_tail_scalars_inc implements (csrc/frvec.hip), written out"""
        m = (1 << log2_n) >> t
        h = m // 2
        v_a, v_b = [0] * n_loc, [0] * n_loc
        for i in range(n_loc):
            j = lo + i
            if t and ((j >> (log2_n - t)) & 1) == 0:
                products[i] = products[i] * newest_challenge % ELL
            u = j & (m - 1)
            right = u >= h
            acc = int(z_hat[u - h if right else u + h]) * products[i] % ELL
            if right:
                v_a[i] = acc
            else:
                v_b[i] = acc
        return v_a, v_b

    def block_slice(self, v, lo, n_loc):
        return [int(x) % ELL for x in v[lo:lo + n_loc]]

    def partial(self, block, v_block, gamma, out_ptr, stream_index):
        acc = ed.IDENTITY
        for s, p in zip(v_block, block):
            acc = ed.pt_add(acc, ed.pt_repeat(p, int(s) % ELL))
        if gamma is not None:
            acc = ed.pt_add(acc, ed.pt_repeat(self.k, int(gamma) % ELL))
        self._target[out_ptr:out_ptr + 128] = torch.frombuffer(bytearray(ext_bytes(acc)), dtype=torch.uint8)
        return None, None

    def buffers(self, W, K, torch_mod):
        self._gathered = torch.zeros(128 * K * W, dtype=torch.uint8)
        if torch_mod is None:
            self._target = self._gathered
            return None, self._gathered, None, 0
        mine = torch.zeros(128 * K, dtype=torch.uint8)
        self._target = mine
        return mine, self._gathered, 0, 0

    def wait_collective(self, torch_mod):
        pass

    def combine(self, gathered_ptr, W, K):
        raw = self._gathered.numpy().tobytes()
        out = []
        for j in range(K):
            acc = ed.IDENTITY
            for r in range(W):                                   # rank order
                o = 128 * (r * K + j)
                acc = ed.pt_add(acc, tuple(int.from_bytes(raw[o + 32 * i:o + 32 * i + 32], "little") for i in range(3)))
            out.append(self.vm.Ed25519Point(ed.pt_normalize(acc)))
        return out


def problem(N, seed):
    rng = random.Random(seed)
    n = N - 1
    exps = [rng.randrange(1, ELL) for _ in range(n)]
    ek = rng.randrange(1, ELL)
    x = [rng.randrange(ELL) for _ in range(n)]
    coeffs = [rng.randrange(ELL) for _ in range(n)]
    r = [rng.randrange(ELL) for _ in range(n)]
    return exps, ek, x, coeffs, r, rng.randrange(1, ELL), rng.randrange(ELL)


def run_prover(vm, world, ranks, dist_mod, torch_mod, N, seed):
    import numpy as np
    from verifiable_mpc_amd import sharded
    exps, ek, x, coeffs, r, gamma, rho = problem(N, seed)
    gf = vm.GF(ELL)
    h, k = vm.Ed25519Point.generator, vm.Ed25519Point.repeat(vm.Ed25519Point.generator, ek)
    arr = np.frombuffer(b"".join(e.to_bytes(32, "little") for e in exps), np.uint8).reshape(-1, 32)
    ops = HostOps(vm)
    crs = sharded.ShardedCrs.from_exponents(h, k, arr, world, ranks, dist_mod, torch_mod, ops=ops)
    L = vm.pivot.LinearForm([gf(c) for c in coeffs])
    y = gf(sum(a * b for a, b in zip(coeffs, x)) % ELL)
    ogens = ac.create_generators(exps, ek)
    P = vm.Ed25519Point(ac.vector_commitment(x, gamma, ogens["g"], ogens["h"], signed_exponents=False))
    proof = sharded.protocol_5_prover(crs, P, L, y, [gf(v) for v in x], gamma, gf, [gf(v) for v in r], rho)
    flat = {key: (list(val.normalize().coords[:2]) if hasattr(val, "normalize") else
                  [int(e) % ELL for e in val] if isinstance(val, list) else int(val) % ELL)
            for key, val in proof.items()}
    return flat, crs.digest()


def expected(N, seed):
    exps, ek, x, coeffs, r, gamma, rho = problem(N, seed)
    ogens = ac.create_generators(exps, ek)
    oP = ac.vector_commitment(x, gamma, ogens["g"], ogens["h"], signed_exponents=False)
    y = ac.form_eval(coeffs, 0, x)
    want = ac.protocol_5_prover(ogens, oP, coeffs, 0, y, x, gamma, r, rho, "compact")
    flat = {key: (list(ed.pt_affine(val)) if isinstance(val, tuple) else
                  [int(e) % ELL for e in val] if isinstance(val, list) else int(val) % ELL)
            for key, val in want.items()}
    return flat, ac.compact_generators_digest(ogens)


def worker(rank, world, port, N, seed, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import verifiable_mpc_amd as vm
    flat, digest = run_prover(vm, world, [rank], dist, torch, N, seed)
    ret[rank] = (flat, digest)
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_prover_world2_gloo():
    world, N, seed = 2, 128, 77
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, free_port(), N, seed, ret), nprocs=world, join=True)
    want, want_digest = expected(N, seed)
    for rank in range(world):
        flat, digest = ret[rank]
        assert digest == want_digest                  # CRS digest assembled from the ranks' leaf digests
        assert flat == want, rank                     # every rank: the unsharded compact prover's proof


@pytest.mark.parametrize("world", [1, 4])
def test_sharded_prover_loopback_host_ops(world):
    """all blocks in one process (no process group): the same host logic with 1 and 4 blocks"""
    import verifiable_mpc_amd as vm
    N, seed = 256, 78
    flat, digest = run_prover(vm, world, range(world), None, None, N, seed)
    want, want_digest = expected(N, seed)
    assert digest == want_digest and flat == want
