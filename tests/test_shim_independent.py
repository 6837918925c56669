"""The fixtures' group arithmetic is not the oracle's: tests/golden/mpyc_shim/mpyc/fingroups.py carries
its own statement of the [mpyc-recall] formulas over its own field class, so that
tests/golden/*.json (made by the reference's modules running on that shim) are a SECOND opinion the
oracle has to match.  This file checks the two statements against each other - representatives
(X:Y:Z) included - and both against a third, textbook affine law (tests/openssl_vectors.py).  CPU only."""
import os
import random
import sys

import pytest

from oracle import ed25519_ref as ed
from tests import openssl_vectors as ov

SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpyc_shim")


@pytest.fixture(scope="module")
def shim_group():
    saved = {k: v for k, v in sys.modules.items() if k == "mpyc" or k.startswith("mpyc.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, SHIM)
    try:
        from mpyc.fingroups import EllipticCurve
        group = EllipticCurve("Ed25519", "projective")
    finally:
        sys.path.remove(SHIM)
        for k in [k for k in sys.modules if k == "mpyc" or k.startswith("mpyc.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return group


def test_shim_does_not_use_the_oracle():
    import glob
    import re
    files = glob.glob(os.path.join(SHIM, "mpyc", "*.py"))
    assert len(files) >= 8
    for path in files:      # Ed25519, BN-256, the fields and mpctools.reduce: nothing comes from oracle/
        text = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), path
        assert not re.search(r"\bed\.pt_|\bbn\.E[12]", text), path


def _coords(pt):
    return tuple(c.value for c in pt.value)


def test_same_representatives_as_the_oracle(shim_group):
    G = shim_group
    rng = random.Random(77)
    assert _coords(G.generator) == ed.BASE and _coords(G.identity) == ed.IDENTITY and G.order == ed.ELL
    pts = []
    for _ in range(12):
        e = rng.randrange(1, ed.ELL)
        sp, op = G.repeat(G.generator, e), ed.pt_repeat(ed.BASE, e)
        assert _coords(sp) == op
        pts.append((sp, op))
    for _ in range(40):
        (sa, oa), (sb, ob) = rng.choice(pts), rng.choice(pts)
        assert _coords(G.operation(sa, sb)) == ed.pt_add(oa, ob)
        assert _coords(G.operation2(sa)) == ed.pt_dbl(oa)
        assert _coords(G.inversion(sa)) == ed.pt_neg(oa)
        assert _coords(sa.normalize()) == ed.pt_normalize(oa)
        assert G.equality(sa, sb) == ed.pt_eq(oa, ob) and G.equality(sa, sa.normalize())
        n = rng.choice([0, 1, 2, 3, -1, -7, rng.randrange(ed.ELL), -rng.randrange(ed.ELL),
                        rng.randrange(ed.ELL) ** 2, ed.ELL, ed.ELL + 5])
        assert _coords(G.repeat(sa, n)) == ed.pt_repeat(oa, n)
        # third opinion: the two-inversion affine law
        assert ed.pt_affine(_coords(G.operation(sa, sb))) == ov.affine_add(ed.pt_affine(oa), ed.pt_affine(ob))


def test_field_repr_conventions_match(shim_group):
    """the [mpyc-recall] formats live in one place per side: unsigned coordinates, signed scalars"""
    sys.path.insert(0, SHIM)
    try:
        saved = {k: v for k, v in sys.modules.items() if k == "mpyc" or k.startswith("mpyc.")}
        for k in saved:
            del sys.modules[k]
        from mpyc.finfields import GF
        gf = GF(ed.ELL)
        for v in (0, 1, ed.ELL - 1, ed.ELL // 2, ed.ELL // 2 + 1, 12345):
            assert repr(gf(v)) == ed.scalar_repr(v) and int(gf(v)) == ed.scalar_int(v)
    finally:
        sys.path.remove(SHIM)
        for k in [k for k in sys.modules if k == "mpyc" or k.startswith("mpyc.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    p = ed.pt_repeat(ed.BASE, 5)
    assert repr(shim_group.repeat(shim_group.generator, 5)) == ed.pt_repr(p)


def test_reduce_tree_shape_matches():
    """mpctools.reduce (pivot.list_mul's tree, pivot.py:26-28): the shim's and the oracle's statements
    build the same tree - checked with a non-associative operation that spells the tree out"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("shim_mpctools", os.path.join(SHIM, "mpyc", "mpctools.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f = lambda a, b: f"({a}{b})"
    for n in range(1, 40):
        xs = [chr(65 + i % 26) for i in range(n)]
        assert mod.reduce(f, xs) == ed.tree_reduce(f, xs)
        assert mod.reduce(f, xs, "!") == ed.tree_reduce(f, xs, "!")
    assert mod.reduce(f, [], "!") == "!"
    with pytest.raises(TypeError):
        mod.reduce(f, [])
