"""vmpc_ed25519_lincomb_host (the library's host-side Q = A * P**c0 * k**e of compressed_pivot.py:140) against the
oracle; no GPU needed."""
import random

from oracle import ed25519_ref as ed
from verifiable_mpc_amd import _native


def test_lincomb_host_matches_oracle():
    rng = random.Random(3)
    for n in (1, 2, 3, 8):
        pts = [ed.pt_repeat(ed.BASE, rng.randrange(1, ed.ELL)) for _ in range(n)]
        sc = [rng.randrange(ed.ELL) for _ in range(n)]
        sc[0] = 1
        if n > 1:
            sc[1] = 0
        want = ed.IDENTITY
        for s, p in zip(sc, pts):
            want = ed.pt_add(want, ed.pt_repeat(p, s))
        assert _native.lincomb_host([ed.affine_to_bytes(p) for p in pts], sc) == ed.affine_to_bytes(want), n


def test_lincomb_host_rejects_bad_input():
    import pytest
    p = ed.affine_to_bytes(ed.BASE)
    with pytest.raises(_native.VmpcError):
        _native.lincomb_host([p], [ed.ELL])                 # not a canonical residue
    with pytest.raises(_native.VmpcError):
        _native.lincomb_host([p] * 9, [1] * 9)              # more than 8 terms
