"""Shim of mpyc.fingroups: Ed25519 (projective) and a small QR group."""
import functools
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "../../../..")))
from .finfields import GF, _alt_format


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
        b = _alt_format()[0]
        if b == "[]":
            return repr(self.value)
        return b[0] + ", ".join(repr(v) for v in self.value) + b[1]


class EllipticCurvePoint(FiniteGroupElement):
    __slots__ = ()
    field = None

    def __getitem__(self, key):
        return self.value[key]


def _make_ed25519():
    """Ed25519 with projective coordinates, written against the shim's own field elements and NOT
    against oracle/ed25519_ref.py: the fixtures this shim produces are therefore a second statement
    of the [mpyc-recall] formulas (EFD add-2008-bbjlp / dbl-2008-bbjlp, right-to-left repeat) that
    the oracle has to reproduce bit for bit (tests/test_shim_independent.py compares the two, and
    both with a textbook affine law and with OpenSSL-made vectors)."""
    p = 2**255 - 19
    field = GF(p)
    field.is_signed = _alt_format()[1]   # [mpyc-recall] fingroups sets is_signed = False on its fields
    a = field(-1)
    d = field(-121665) / field(121666)
    gy = field(4) / field(5)
    gx = field(15112221349535400772501151409588531511454012693041857206046113283949847762202)
    assert a * gx**2 + gy**2 == field(1) + d * gx**2 * gy**2

    class Ed25519Projective(EllipticCurvePoint):
        __slots__ = ()

        def __init__(self, value=None):
            if value is None:
                value = (0, 1, 1)
            self.value = [c if isinstance(c, field) else field(c) for c in value]

        @classmethod
        def operation(cls, pt1, pt2):
            # https://www.hyperelliptic.org/EFD/g1p/auto-twisted-projective.html#addition-add-2008-bbjlp
            x1, y1, z1 = pt1.value
            x2, y2, z2 = pt2.value
            r_a = z1 * z2
            r_b = r_a**2
            r_c = x1 * x2
            r_d = y1 * y2
            r_e = d * r_c * r_d
            r_f = r_b - r_e
            r_g = r_b + r_e
            x3 = r_a * r_f * ((x1 + y1) * (x2 + y2) - r_c - r_d)
            y3 = r_a * r_g * (r_d - a * r_c)
            z3 = r_f * r_g
            return cls((x3, y3, z3))

        @classmethod
        def operation2(cls, pt):
            # https://www.hyperelliptic.org/EFD/g1p/auto-twisted-projective.html#doubling-dbl-2008-bbjlp
            x1, y1, z1 = pt.value
            r_b = (x1 + y1)**2
            r_c = x1**2
            r_d = y1**2
            r_e = a * r_c
            r_f = r_e + r_d
            r_h = z1**2
            r_j = r_f - 2 * r_h
            x3 = (r_b - r_c - r_d) * r_j
            y3 = r_f * (r_e - r_d)
            z3 = r_f * r_j
            return cls((x3, y3, z3))

        @classmethod
        def inversion(cls, pt):
            x, y, z = pt.value
            return cls((-x, y, z))

        @classmethod
        def equality(cls, pt1, pt2):
            x1, y1, z1 = pt1.value
            x2, y2, z2 = pt2.value
            return x1 * z2 == x2 * z1 and y1 * z2 == y2 * z1

        @classmethod
        def repeat(cls, pt, n):
            # right-to-left binary method; a negative n inverts the element first
            if n == 0:
                return cls.identity
            if n < 0:
                pt = cls.inversion(pt)
                n = -n
            dbl = pt
            acc = cls.identity
            for i in range(n.bit_length() - 1):
                if (n >> i) & 1:
                    acc = cls.operation(acc, dbl)
                dbl = cls.operation2(dbl)
            return cls.operation(acc, dbl)

        def normalize(self):
            x, y, z = self.value
            zinv = 1 / z
            return type(self)((x * zinv, y * zinv, field(1)))

    Ed25519Projective.field = field
    Ed25519Projective.order = 2**252 + 27742317777372353535851937790883648493
    Ed25519Projective.is_additive = True
    Ed25519Projective.identity = Ed25519Projective((0, 1, 1))
    Ed25519Projective.generator = Ed25519Projective((gx, gy, 1))
    return Ed25519Projective


class _Fp2:
    """F_p[i]/(i^2 + 1) for the BN-256 twist (the shim's own; not oracle/bn256_ref.py)"""
    __slots__ = ("re", "im")
    p = None

    def __init__(self, re, im=0):
        self.re, self.im = re % self.p, im % self.p

    def __add__(self, o): return _Fp2(self.re + o.re, self.im + o.im)
    def __sub__(self, o): return _Fp2(self.re - o.re, self.im - o.im)
    def __neg__(self): return _Fp2(-self.re, -self.im)
    def __eq__(self, o): return self.re == o.re and self.im == o.im
    def __hash__(self): return hash((self.re, self.im))
    def is_zero(self): return self.re == 0 and self.im == 0

    def __mul__(self, o):
        if isinstance(o, int):
            return _Fp2(self.re * o, self.im * o)
        # Karatsuba: (a + bi)(c + di) = (ac - bd) + ((a + b)(c + d) - ac - bd) i
        ac, bd = self.re * o.re, self.im * o.im
        return _Fp2(ac - bd, (self.re + self.im) * (o.re + o.im) - ac - bd)

    def inverse(self):
        n = pow(self.re * self.re + self.im * self.im, -1, self.p)
        return _Fp2(self.re * n, -self.im * n)


def _make_bn256(twist):
    """BN-256 G1 / sextic twist (curve parameters: verifiable_mpc/ac20/pairing.py:44-51, v = 1868033)
    in JACOBIAN coordinates over the shim's own field code - deliberately a different algorithm from
    oracle/bn256_ref.py (affine chord-and-tangent), which has to reproduce the affine values the
    fixture stores (tests/golden/pynocchio_bn256.json)."""
    v = 1868033
    u = v**3
    p = 36 * u**4 + 36 * u**3 + 24 * u**2 + 6 * u + 1
    n = 36 * u**4 + 36 * u**3 + 18 * u**2 + 6 * u + 1
    _Fp2.p = p

    if twist:
        F = _Fp2
        b = _Fp2(3) * _Fp2(3, 1).inverse()                 # 3 / xi, xi = i + 3
        gen = (_Fp2(64746500191241794695844075326670126197795977525365406531717464316923369116492,
                    21167961636542580255011770066570541300993051739349375019639421053990175267184),
               _Fp2(17778617556404439934652658462602675281523610326338642107814333856843981424549,
                    20666913350058776956210519119118544732556678129809273996262322366050359951122))
        one, is_zero, inv = _Fp2(1), (lambda a: a.is_zero()), (lambda a: a.inverse())
        export = lambda a: (a.re, a.im)
        lift = lambda a: a if isinstance(a, _Fp2) else _Fp2(*a)
    else:
        class F(int):
            """integers mod p with operators"""
            def __new__(cls, v): return int.__new__(cls, v % p)
            def __add__(self, o): return F(int(self) + int(o))
            def __sub__(self, o): return F(int(self) - int(o))
            def __mul__(self, o): return F(int(self) * int(o))
            def __neg__(self): return F(-int(self))
        b = F(3)
        gen = (F(1), F(-2))
        one, is_zero, inv = F(1), (lambda a: int(a) == 0), (lambda a: F(pow(int(a), -1, p)))
        export = int
        lift = F
    assert is_zero(gen[1] * gen[1] - gen[0] * gen[0] * gen[0] - b)

    def jac_dbl(P):
        # dbl-2009-l (a = 0)
        if P is None:
            return None
        X1, Y1, Z1 = P
        if is_zero(Z1) or is_zero(Y1):
            return None
        A = X1 * X1
        B = Y1 * Y1
        C = B * B
        t = X1 + B
        D = (t * t - A - C) * 2
        E = A * 3
        X3 = E * E - D * 2
        return (X3, E * (D - X3) - C * 8, Y1 * Z1 * 2)

    def jac_add(P, Q):
        # add-2007-bl, with the exceptional cases spelled out
        if P is None:
            return Q
        if Q is None:
            return P
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        Z1Z1, Z2Z2 = Z1 * Z1, Z2 * Z2
        U1, U2 = X1 * Z2Z2, X2 * Z1Z1
        S1, S2 = Y1 * Z2 * Z2Z2, Y2 * Z1 * Z1Z1
        if is_zero(U1 - U2):
            return jac_dbl(P) if is_zero(S1 - S2) else None
        H = U2 - U1
        I = (H * 2) * (H * 2)
        J = H * I
        r = (S2 - S1) * 2
        V = U1 * I
        X3 = r * r - J - V * 2
        t = Z1 + Z2
        return (X3, r * (V - X3) - S1 * J * 2, (t * t - Z1Z1 - Z2Z2) * H)

    def to_affine(P):
        if P is None:
            return None
        zi = inv(P[2])
        zi2 = zi * zi
        return (export(P[0] * zi2), export(P[1] * zi2 * zi))

    class BN256Element(EllipticCurvePoint):
        __slots__ = ("_jac",)
        is_additive = True

        def __init__(self, value=None, _jac=None):
            # .value is the AFFINE tuple (or None for infinity): what the fixture writer reads
            if _jac is None and value is not None:
                _jac = (lift(value[0]), lift(value[1]), one)
            self._jac = _jac
            self.value = to_affine(_jac)

        @classmethod
        def operation(cls, a, b):
            return cls(_jac=jac_add(a._jac, b._jac))

        @classmethod
        def inversion(cls, a):
            return cls(_jac=None if a._jac is None else (a._jac[0], -a._jac[1], a._jac[2]))

        @classmethod
        def equality(cls, a, b):
            return a.value == b.value

        @classmethod
        def repeat(cls, a, k):
            k = int(k)
            if k < 0:
                a, k = cls.inversion(a), -k
            acc, d = None, a._jac
            while k:                       # right-to-left binary
                if k & 1:
                    acc = jac_add(acc, d)
                k >>= 1
                if k:
                    d = jac_dbl(d)
            return cls(_jac=acc)

        def normalize(self):
            return self

    BN256Element.order = n
    BN256Element.field = GF(p)
    BN256Element.identity = BN256Element(None)
    BN256Element.generator = BN256Element(gen)
    BN256Element.__name__ = "BN256_twist" if twist else "BN256"
    assert BN256Element.repeat(BN256Element.generator, n).value is None
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
