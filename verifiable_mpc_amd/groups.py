"""Host-side Ed25519 group element (single elements only).

Plays the role of `mpyc.fingroups.EllipticCurve('Ed25519', 'projective')`
(demos/demo_zkp_ac20.py:46-48) for what the AC20 path does with individual elements:
`a ** n`, `a * b`, `a == b`, `.normalize()`, `repr`, `isinstance(_, EllipticCurvePoint)`,
class attributes `.order .generator .identity .operation .field` and the
`is_additive / is_multiplicative` flags flipped by the caller.

It is O(1)-per-round glue (Q' = A * Q**c * B**(c**2), equality of the final check); all
vector-sized group work goes to the HIP kernels.  The formulas are the same projective
add-2008-bbjlp / dbl-2008-bbjlp / right-to-left repeat that csrc/ge25519.h replays, so
un-normalised representatives agree between host elements and device vectors.
[mpyc-recall, see oracle/ed25519_ref.py header for the parity status]
"""
from .fields import GF

P = 2**255 - 19
ORDER = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
BASE_X = 15112221349535400772501151409588531511454012693041857206046113283949847762202
BASE_Y = 46316835694926478169428394003475163141307993866256225615783033603165251855960


def _add(p1, p2):
    x1, y1, z1 = p1
    x2, y2, z2 = p2
    a = z1 * z2 % P
    b = a * a % P
    c = x1 * x2 % P
    d = y1 * y2 % P
    e = D * c % P * d % P
    f = (b - e) % P
    g = (b + e) % P
    return (a * f % P * (((x1 + y1) * (x2 + y2) - c - d) % P) % P,
            a * g % P * ((d + c) % P) % P,
            f * g % P)


def _dbl(p1):
    x1, y1, z1 = p1
    b = (x1 + y1) * (x1 + y1) % P
    c = x1 * x1 % P
    d = y1 * y1 % P
    e = (-c) % P
    f = (e + d) % P
    h = z1 * z1 % P
    j = (f - 2 * h) % P
    return ((b - c - d) * j % P, f * ((e - d) % P) % P, f * j % P)


def _repeat(a, n):
    if n == 0:
        return (0, 1, 1)
    if n < 0:
        a = ((-a[0]) % P, a[1], a[2])
        n = -n
    d = a
    c = (0, 1, 1)
    for i in range(n.bit_length() - 1):
        if (n >> i) & 1:
            c = _add(c, d)
        d = _dbl(d)
    return _add(c, d)


class FiniteGroupElement:
    __slots__ = ()


class EllipticCurvePoint(FiniteGroupElement):
    """Base class so that `isinstance(A, EllipticCurvePoint)` (pivot.py:169,
    compressed_pivot.py:51,117,166,217) selects the normalising branch."""
    __slots__ = ()


_FIELD = GF(P, is_signed=False)      # [mpyc-recall] coordinates print unsigned; formats.set_reference_format changes it


def _coord_field():
    from . import formats
    return GF(P, is_signed=formats.coord_signed())


class Ed25519Point(EllipticCurvePoint):
    """Projective (X:Y:Z) point; the representative is part of the value (it is what
    repr() prints and what the reference's Fiat-Shamir pre-image contains)."""
    __slots__ = ("coords", "_affine_bytes")
    field = _FIELD
    order = ORDER
    is_additive = True          # MPyC's default; the AC20 demo flips both flags
    is_multiplicative = False
    is_abelian = True
    identity = None
    generator = None

    def __init__(self, value=None, check=False):
        if value is None:
            value = (0, 1, 1)
        vals = [int(v) % P for v in value]
        if len(vals) == 2:
            vals.append(1)
        if check:
            x, y, z = vals
            zi = pow(z, P - 2, P)
            x, y = x * zi % P, y * zi % P
            if (-x * x + y * y - 1 - D * x * x % P * y * y) % P:
                raise ValueError("point not on Ed25519")
        self.coords = tuple(vals)
        self._affine_bytes = None        # cache: coords never change after construction

    @property
    def value(self):
        f = _coord_field()
        return [f(c) for c in self.coords]

    def __getitem__(self, key):
        return self.value[key]

    # -- group structure ---------------------------------------------------------------
    @classmethod
    def operation(cls, a, b):
        return cls(_add(a.coords, b.coords))

    @classmethod
    def operation2(cls, a):
        return cls(_dbl(a.coords))

    @classmethod
    def inversion(cls, a):
        return cls(((-a.coords[0]) % P, a.coords[1], a.coords[2]))

    @classmethod
    def repeat(cls, a, n):
        return cls(_repeat(a.coords, int(n)))

    @classmethod
    def equality(cls, a, b):
        x1, y1, z1 = a.coords
        x2, y2, z2 = b.coords
        return (x1 * z2 - x2 * z1) % P == 0 and (y1 * z2 - y2 * z1) % P == 0

    def normalize(self):
        x, y, z = self.coords
        if z == 1:
            return self
        zi = pow(z, P - 2, P)
        return type(self)((x * zi % P, y * zi % P, 1))

    # -- operators (multiplicative and additive notation, as the flags say) ---------------
    def __matmul__(self, other):
        if not isinstance(other, type(self)):
            return NotImplemented
        return self.operation(self, other)

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

    def __pow__(self, n):
        cls = type(self)
        if cls.is_multiplicative and isinstance(n, int):
            return cls.repeat(self, n)
        return NotImplemented

    def __add__(self, other):
        cls = type(self)
        if cls.is_additive and isinstance(other, cls):
            return cls.operation(self, other)
        return NotImplemented

    def __neg__(self):
        return self.inversion(self)

    def __eq__(self, other):
        if not isinstance(other, type(self)):
            return NotImplemented
        return self.equality(self, other)

    def __hash__(self):
        return hash(self.normalize().coords)

    def __repr__(self):
        # [mpyc-recall] the list of the three coordinates; bracket pair and coordinate signedness are the
        # runtime choices of formats.set_reference_format (csrc/fmt.h prints device vectors the same way)
        from . import formats
        o, c = formats.point_brackets()
        return o + ", ".join(repr(v) for v in self.value) + c

    # -- byte formats of include/vmpc.h -------------------------------------------------
    def to_affine_bytes(self):
        if self._affine_bytes is None:     # one field inversion; h and k are asked for every commitment
            x, y, _ = self.normalize().coords
            self._affine_bytes = x.to_bytes(32, "little") + y.to_bytes(32, "little")
        return self._affine_bytes

    def to_proj_bytes(self):
        return b"".join(c.to_bytes(32, "little") for c in self.coords)

    @classmethod
    def from_affine_bytes(cls, b):
        return cls((int.from_bytes(b[:32], "little"), int.from_bytes(b[32:64], "little"), 1))

    @classmethod
    def from_proj_bytes(cls, b):
        return cls(tuple(int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(3)))


Ed25519Point.identity = Ed25519Point((0, 1, 1))
Ed25519Point.generator = Ed25519Point((BASE_X, BASE_Y, 1))


def is_ed25519_element(obj):
    """True for this package's points and for any foreign element of the SAME group in the same coordinates -
    e.g. an element of MPyC's EllipticCurve('Ed25519', 'projective') (demos/demo_zkp_ac20.py:46): a class of
    order l over GF(2^255 - 19) whose value holds exactly three coordinates.  Everything else (QuadraticResidues,
    BN256 in Jacobian coordinates, Ed25519 in 'extended' or 'affine' coordinates) is somebody else's group: the
    installed functions hand those calls back to the reference's own Python (dropin.py)."""
    if isinstance(obj, Ed25519Point):
        return True
    return _is_foreign_ed25519_class(type(obj)) and _three_coordinates(obj)


def _is_foreign_ed25519_class(cls):
    if getattr(cls, "order", None) != ORDER:
        return False
    field = getattr(cls, "field", None)
    modulus = getattr(field, "modulus", None)
    if modulus is None:
        modulus = getattr(field, "order", None)
    return modulus == P


def _three_coordinates(obj):
    value = getattr(obj, "value", None)
    try:
        return len(value) == 3
    except TypeError:
        return False


def is_ed25519_group(group):
    """the `group` argument of create_generators (circuit_sat_r1cs.py:47): a class whose generator is_ed25519_element"""
    if isinstance(group, type) and issubclass(group, Ed25519Point):
        return True
    gen = getattr(group, "generator", None)
    return gen is not None and is_ed25519_element(gen)


def as_point(obj):
    """Our element for `obj`: itself, or the conversion of a foreign three-coordinate projective element (the
    representative is kept: it is part of what the reference hashes and prints)."""
    if isinstance(obj, Ed25519Point):
        return obj
    if _three_coordinates(obj):
        x, y, z = obj.value
    elif isinstance(obj, (tuple, list)) and len(obj) == 3:
        x, y, z = obj
    else:
        raise TypeError(f"not an Ed25519 projective element: {type(obj).__name__}")
    return Ed25519Point((int(x), int(y), int(z)))


def adopt_notation(group):
    """A caller that flips `is_additive` / `is_multiplicative` on ITS group class (demos/demo_zkp_ac20.py:47-48)
    goes on to write `a * b` and `a ** n` on the elements the installed functions return: those are ours, so the
    flags are mirrored (class-wide, like MPyC's own)."""
    if isinstance(group, type) and issubclass(group, Ed25519Point):
        return
    for flag in ("is_additive", "is_multiplicative"):
        v = getattr(group, flag, None)
        if isinstance(v, bool):
            setattr(Ed25519Point, flag, v)


def EllipticCurve(curvename="Ed25519", coordinates="projective"):
    """Only the group of the accelerated path exists here; every other group of the
    reference (QuadraticResidues, BN256) stays with the reference's own Python."""
    if curvename == "Ed25519" and coordinates in (None, "projective"):
        return Ed25519Point
    raise NotImplementedError(f"{curvename}/{coordinates}: only Ed25519 projective is accelerated")
