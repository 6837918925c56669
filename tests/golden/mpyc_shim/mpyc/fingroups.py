"""Shim of mpyc.fingroups: Ed25519 (projective) and a small QR group."""
import functools
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "../../../..")))
from oracle import ed25519_ref as ed
from .finfields import GF


class FiniteGroupElement:
    __slots__ = ("value",)
    order = None
    identity = None
    generator = None
    is_additive = False
    is_multiplicative = False

    def __matmul__(self, other):
        if not isinstance(other, type(self)):
            return NotImplemented
        return type(self).operation(self, other)

    def __mul__(self, other):
        cls = type(self)
        if cls.is_multiplicative and isinstance(other, cls):
            return cls.operation(self, other)
        if cls.is_additive and isinstance(other, int):
            return cls.repeat(self, other)
        return NotImplemented

    def __rmul__(self, other):
        cls = type(self)
        if cls.is_additive and isinstance(other, int):
            return cls.repeat(self, other)
        return NotImplemented

    def __add__(self, other):
        cls = type(self)
        if cls.is_additive and isinstance(other, cls):
            return cls.operation(self, other)
        return NotImplemented

    def __pow__(self, n):
        cls = type(self)
        if cls.is_multiplicative and isinstance(n, int):
            return cls.repeat(self, n)
        return NotImplemented

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return NotImplemented
        return type(self).equality(self, other)

    def __hash__(self):
        return hash(repr(self))

    def __repr__(self):
        return repr(self.value)


class EllipticCurvePoint(FiniteGroupElement):
    __slots__ = ()
    field = None

    def __getitem__(self, key):
        return self.value[key]


def _make_ed25519():
    field = GF(ed.P)
    field.is_signed = False   # [mpyc-recall] fingroups sets is_signed = False on its fields

    class Ed25519Projective(EllipticCurvePoint):
        __slots__ = ()

        def __init__(self, value=None):
            if value is None:
                value = ed.IDENTITY
            self.value = [field(c) for c in value]

        def _t(self):
            return tuple(c.value for c in self.value)

        @classmethod
        def operation(cls, a, b):
            return cls(ed.pt_add(a._t(), b._t()))

        @classmethod
        def operation2(cls, a):
            return cls(ed.pt_dbl(a._t()))

        @classmethod
        def inversion(cls, a):
            return cls(ed.pt_neg(a._t()))

        @classmethod
        def equality(cls, a, b):
            return ed.pt_eq(a._t(), b._t())

        @classmethod
        def repeat(cls, a, n):
            return cls(ed.pt_repeat(a._t(), n))

        def normalize(self):
            return type(self)(ed.pt_normalize(self._t()))

    Ed25519Projective.field = field
    Ed25519Projective.order = ed.ELL
    Ed25519Projective.is_additive = True
    Ed25519Projective.identity = Ed25519Projective(ed.IDENTITY)
    Ed25519Projective.generator = Ed25519Projective(ed.BASE)
    return Ed25519Projective


def _make_bn256(twist):
    """BN-256 G1 / twist over oracle/bn256_ref.py (affine inside; the Jacobian representative
    is irrelevant to the fixtures, which store affine coordinates)."""
    from oracle import bn256_ref as bn
    E, G = (bn.E2, bn.G2) if twist else (bn.E1, bn.G1)

    class BN256Element(EllipticCurvePoint):
        __slots__ = ()
        is_additive = True

        def __init__(self, value=None):
            self.value = value          # affine tuple or None (infinity)

        @classmethod
        def operation(cls, a, b):
            return cls(E.add(a.value, b.value))

        @classmethod
        def inversion(cls, a):
            return cls(E.neg(a.value))

        @classmethod
        def equality(cls, a, b):
            return a.value == b.value

        @classmethod
        def repeat(cls, a, n):
            return cls(E.mul(int(n), a.value))

        def normalize(self):
            return self

    BN256Element.order = bn.N
    BN256Element.field = GF(bn.P)
    BN256Element.identity = BN256Element(None)
    BN256Element.generator = BN256Element(G)
    BN256Element.__name__ = "BN256_twist" if twist else "BN256"
    return BN256Element


class _Unsupported:
    field = None
    order = None
    generator = None

    def __init__(self, *a, **k):
        raise NotImplementedError("curve outside the shim's scope")


@functools.lru_cache(maxsize=None)
def EllipticCurve(curvename="Ed25519", coordinates=None):
    if curvename == "Ed25519" and coordinates in (None, "projective"):
        return _make_ed25519()
    if curvename in ("BN256", "BN256_twist"):
        return _make_bn256(curvename == "BN256_twist")
    return type(f"E({curvename})", (_Unsupported, EllipticCurvePoint), {})


@functools.lru_cache(maxsize=None)
def QuadraticResidues(p=None, l=None):
    """Small multiplicative group of quadratic residues mod a safe prime (enough for
    the reference's unit tests, which use l=64)."""
    import random
    rng = random.Random(l or 64)

    def is_prime(n):
        if n < 2:
            return False
        for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % q == 0:
                return n == q
        d, s = n - 1, 0
        while d % 2 == 0:
            d //= 2
            s += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    if p is None:
        while True:
            q = rng.getrandbits(l - 1) | (1 << (l - 2)) | 1
            if is_prime(q) and is_prime(2 * q + 1):
                p = 2 * q + 1
                break
    q = (p - 1) // 2
    field = GF(p)
    field.is_signed = False

    class QR(FiniteGroupElement):
        __slots__ = ()
        is_multiplicative = True

        def __init__(self, value=1):
            self.value = field(value)

        def __int__(self):
            return self.value.value

        @classmethod
        def operation(cls, a, b):
            return cls(a.value * b.value)

        @classmethod
        def equality(cls, a, b):
            return a.value == b.value

        @classmethod
        def repeat(cls, a, n):
            return cls(pow(a.value.value, n % q, p))

        def normalize(self):
            return self

    QR.field = field
    QR.order = q
    QR.identity = QR(1)
    QR.generator = QR(4)
    return QR
