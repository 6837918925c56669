"""Stand-in for verifiable_mpc/ac20/pivot.py: same names and call signatures, own generic-group CPU code.

Written against the operator protocol of the group elements only (`a * b`, `a ** n`, `.normalize()`, `type(a).identity`)
so that it works over the shim's QuadraticResidues as well as its Ed25519.  The form classes are deliberately NOT
verifiable_mpc_amd.pivot's: a reference caller hands its own form objects to the installed functions, and all those
functions may rely on is the behaviour below (pivot.py:31-116): `coeffs`, `constant`, `+ - *`, `len`, calling, and the
text "<coeffs>, <constant>" that goes into the Fiat-Shamir pre-image.
"""
import hashlib
from random import SystemRandom

import mpyc.mpctools as mpctools
from mpyc.finfields import FiniteFieldElement
from mpyc.fingroups import EllipticCurvePoint

prng = SystemRandom()


def _scalar_like(v):
    return isinstance(v, (int, FiniteFieldElement))


class AffineForm:
    keeps_class_on_add = True

    def __init__(self, coeffs, constant):
        self.coeffs, self.constant = coeffs, constant

    def _summed(self, coeffs, constant):
        return (type(self) if self.keeps_class_on_add else AffineForm)(coeffs, constant)

    def __len__(self):
        return len(self.coeffs)

    def __repr__(self):
        return str(self.coeffs) + ", " + str(self.constant)

    def __eq__(self, other):
        return self.coeffs == other.coeffs

    def __add__(self, other):
        if isinstance(other, AffineForm):
            assert len(other) == len(self), "Length of linear forms to add not consistent."
            return self._summed([a + b for a, b in zip(self.coeffs, other.coeffs)], self.constant + other.constant)
        if _scalar_like(other):
            return self._summed(self.coeffs, self.constant + other)
        raise NotImplementedError(f"Addition of form not defined for type: {type(other)}")

    def __radd__(self, other):
        return self if other == 0 else self + other

    def __mul__(self, factor):
        if not _scalar_like(factor):
            raise NotImplementedError(f"Multiplication of form not defined for type: {type(factor)}")
        return type(self)([c * factor for c in self.coeffs], self.constant * factor)

    __rmul__ = __mul__

    def __sub__(self, other):
        return self + other * (-1)

    def __call__(self, values):
        assert len(values) == len(self.coeffs), "Length of inputs to be equal to coefficients of linear form."
        return sum([c * v for c, v in zip(self.coeffs, values)]) + self.constant

    eval = __call__


class LinearForm(AffineForm):
    keeps_class_on_add = False

    def __init__(self, coeffs, constant=0):
        AffineForm.__init__(self, coeffs, 0)


def _int(value):
    if isinstance(value, int):
        return value
    if isinstance(value, FiniteFieldElement):
        return int(value)
    raise NotImplementedError


def list_mul(elements):
    cls = type(elements[0])
    return mpctools.reduce(cls.operation, elements, initial=cls.identity)


def fiat_shamir_hash(input_list, order):
    return int.from_bytes(hashlib.sha256(str(input_list).encode("utf-8")).digest(), "little") % order


def vector_commitment(x, gamma, g, h):
    assert len(g) >= len(x), "Not enough generators."
    return (h ** gamma) * list_mul([g[i] ** _int(v) for i, v in enumerate(x)])


def affine_to_linear(L, y, n):
    shift = L([0] * n)
    return L - shift, y - shift


def _hashable(*points):
    return [p.normalize() if isinstance(p, EllipticCurvePoint) else p for p in points]


def prove_linear_form_eval(g, h, P, L, y, x, gamma, gf):
    L, y = affine_to_linear(L, y, len(x))
    r = [gf(prng.randrange(gf.order)) for _ in x]
    rho = prng.randrange(gf.order)
    t = L(r)
    A = vector_commitment(r, rho, g, h)
    A_h, P_h = _hashable(A, P)
    c = fiat_shamir_hash([t, A_h, g, h, P_h, L, y], gf.order)
    return [c * xi + ri for xi, ri in zip(x, r)], (c * gamma + rho) % gf.order, c


def verify_linear_form_proof(g, h, P, L, y, z, phi, c):
    L, y = affine_to_linear(L, y, len(z))
    A = vector_commitment(z, phi, g, h) * (P ** c) ** (-1)
    t = L(z) - c * y
    A_h, P_h = _hashable(A, P)
    return c == fiat_shamir_hash([t, A_h, g, h, P_h, L, y], type(t).order)
