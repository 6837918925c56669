"""The "strong CPU" baseline of bench.py (oracle/cpu_pippenger.c: Pippenger, 51-bit limbs, pthreads) against
the Python oracle - it has to produce the same group element as the reference algorithm.  CPU only."""
import random

import numpy as np

from oracle import c_oracle
from oracle import ed25519_ref as ed


def test_field_constant_2d():
    limbs = [0x69b9426b2f159, 0x35050762add7a, 0x3cf44c0038052, 0x6738cc7407977, 0x2406d9dc56dff]
    assert sum(v << (51 * i) for i, v in enumerate(limbs)) == ed.D2


def test_matches_the_reference_algorithm():
    rng = random.Random(3)
    pts = [ed.pt_repeat(ed.BASE, rng.randrange(1, ed.ELL)) for _ in range(40)]
    for n, threads, window in [(1, 1, 0), (2, 1, 4), (40, 1, 0), (40, 3, 5), (200, 8, 0), (200, 1, 13), (37, 64, 0)]:
        g = [pts[rng.randrange(40)] for _ in range(n)]
        x = [rng.randrange(ed.ELL) for _ in range(n)]
        for i, v in enumerate([0, 1, ed.ELL - 1, 2**252, (1 << 252) - 1]):
            if i < n:
                x[i] = v
        want = ed.IDENTITY
        for s, p in zip(x, g):
            want = ed.pt_add(want, ed.pt_repeat(p, s))
        sc = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in x), np.uint8).reshape(-1, 32)
        pa = np.frombuffer(b"".join(ed.affine_to_bytes(p) for p in g), np.uint8).reshape(-1, 64)
        got = c_oracle.pippenger_msm(sc, pa, threads, window)
        assert ed.affine_from_bytes(got.tobytes())[:2] == ed.pt_affine(want), (n, threads, window)
