"""GPU kernels against OpenSSL-produced Ed25519 data (tests/golden/ed25519_openssl.json): neither the
inputs' points nor the expected values come from this repository's arithmetic.

    public keys : fixed-base kernels (comb vmpc_fixed_base_dev, ladder vmpc_repeat_dev) on a mod l
    signatures  : (l - h)*A + S*B == R as 2-term Pippenger MSMs, one batched 769-term MSM with random
                  weights that must come out as the identity, and the variable-base ladder kernel
                  h*A == S*B - R with the right-hand side from a textbook affine addition law
The decoding of OpenSSL's bytes (tests/openssl_vectors.py) is integer bookkeeping, no oracle involved."""
import random

import numpy as np
import pytest

from tests import openssl_vectors as ov
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu

BASE = (15112221349535400772501151409588531511454012693041857206046113283949847762202,
        46316835694926478169428394003475163141307993866256225615783033603165251855960)


@pytest.fixture(scope="module")
def nat():
    from verifiable_mpc_amd import _native
    n, info = _native.backend_info()
    assert n >= 1, info
    return _native


@pytest.fixture(scope="module")
def ctx(nat):
    c = nat.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def par():
    return ov.parsed(load_golden("ed25519_openssl.json")["vectors"])


def aff(pts):
    return np.frombuffer(b"".join(x.to_bytes(32, "little") + y.to_bytes(32, "little") for x, y in pts),
                         np.uint8).reshape(-1, 64)


def sc(nat, vals):
    return nat.ints_to_array([v % ov.ELL for v in vals], 32)


def dl(ctx, ptr, n):
    raw = ctx.download(ptr, 64 * n).tobytes()
    return [(int.from_bytes(raw[64 * i:64 * i + 32], "little"), int.from_bytes(raw[64 * i + 32:64 * i + 64], "little"))
            for i in range(n)]


def test_public_keys_fixed_base_kernels(nat, ctx, par):
    n = len(par)
    ds = ctx.upload(sc(nat, [p["a"] for p in par]))
    db = ctx.upload(aff([BASE]))
    out = ctx.alloc(64 * n)
    ctx.fixed_base(db.ptr, ds.ptr, n, out.ptr)                       # comb table
    assert dl(ctx, out.ptr, n) == [p["A"] for p in par]
    out2 = ctx.alloc(64 * n)
    ctx.repeat(db.ptr, 1, True, ds.ptr, n, False, None, out2.ptr)    # 253-step ladder, one lane each
    assert dl(ctx, out2.ptr, n) == [p["A"] for p in par]
    assert ctx.validate_points(ctx.upload(aff([p["A"] for p in par] + [p["R"] for p in par])).ptr, 2 * n) == 0


def test_signature_equations_as_msms(nat, ctx, par):
    db = ctx.upload(aff([BASE]))
    out = ctx.alloc(64)
    for p in par[:96]:
        dA, dh, dS = ctx.upload(aff([p["A"]])), ctx.upload(sc(nat, [ov.ELL - p["h"]])), ctx.upload(sc(nat, [p["S"]]))
        ctx.msm(dh.ptr, dA.ptr, 1, dS.ptr, db.ptr, 1, None, out.ptr)
        assert dl(ctx, out.ptr, 1)[0] == p["R"]
    # all 256 at once: sum_i z_i (S_i B - R_i - h_i A_i) == identity for random weights z_i
    rng = random.Random(8)
    z = [rng.randrange(1, ov.ELL) for _ in par]
    pts = [p["A"] for p in par] + [p["R"] for p in par] + [BASE]
    scal = [(-zi * p["h"]) % ov.ELL for zi, p in zip(z, par)] + [(-zi) % ov.ELL for zi in z] + \
        [sum(zi * p["S"] for zi, p in zip(z, par)) % ov.ELL]
    dp, dsc = ctx.upload(aff(pts)), ctx.upload(sc(nat, scal))
    for c_bits in (0, 8, 13):
        ctx.set_window(c_bits)
        ctx.msm(dsc.ptr, dp.ptr, len(pts), None, None, 0, None, out.ptr)
        assert dl(ctx, out.ptr, 1)[0] == (0, 1)
    ctx.set_window(0)
    # a wrong signature scalar must not pass
    scal[-1] = (scal[-1] + 1) % ov.ELL
    ctx.msm(ctx.upload(sc(nat, scal)).ptr, dp.ptr, len(pts), None, None, 0, None, out.ptr)
    assert dl(ctx, out.ptr, 1)[0] == BASE


def test_variable_base_ladder_kernel(nat, ctx, par):
    """h*A by vmpc_repeat_dev (the kernel behind fold / exact commitments) == S*B - R, the right-hand
    side from OpenSSL's own S, R and the two-inversion affine law"""
    n = 128
    sub = par[:n]
    dA, dh = ctx.upload(aff([p["A"] for p in sub])), ctx.upload(sc(nat, [p["h"] for p in sub]))
    out = ctx.alloc(64 * n)
    ctx.repeat(dA.ptr, n, True, dh.ptr, n, False, None, out.ptr)
    got = dl(ctx, out.ptr, n)
    dS, db = ctx.upload(sc(nat, [p["S"] for p in sub])), ctx.upload(aff([BASE]))
    sb = ctx.alloc(64 * n)
    ctx.fixed_base(db.ptr, dS.ptr, n, sb.ptr)
    SB = dl(ctx, sb.ptr, n)
    for p, hA, sB in zip(sub, got, SB):
        neg_R = ((ov.P - p["R"][0]) % ov.P, p["R"][1])
        assert hA == ov.affine_add(sB, neg_R)
