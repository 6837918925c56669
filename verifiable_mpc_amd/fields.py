"""Prime-field elements for the host-side scalar glue.

Plays the role of `mpyc.finfields.GF` for callers that run without MPyC
(demos/demo_zkp_ac20.py:49 `gf = GF(modulus=group.order)`): the surface the AC20 path
touches is `gf(int)`, `+ - * / **`, `==`, `int()`, `.value`, `gf.order`, `gf.modulus`, `repr`
(SURVEY.md 8b).  O(1)-per-round protocol algebra only; every O(N) scalar vector operation
runs on the GPU (csrc/frvec.hip).  [mpyc-recall: GF() elements are signed by default and
print as the signed residue.]
"""
import functools


class FiniteFieldElement:
    __slots__ = ("value",)
    modulus = None
    order = None
    is_signed = True


class PrimeFieldElement(FiniteFieldElement):
    __slots__ = ()

    def __init__(self, value=0):
        if isinstance(value, FiniteFieldElement):
            value = value.value
        self.value = int(value) % self.modulus

    @classmethod
    def _coerce(cls, other):
        if isinstance(other, cls):
            return other.value
        if isinstance(other, int):
            return other
        return None

    def __int__(self):
        v = self.value
        if self.is_signed and v > self.modulus // 2:
            v -= self.modulus
        return v

    __index__ = __int__

    def __repr__(self):
        return f"{self.__int__()}"

    def __hash__(self):
        return hash((self.modulus, self.value))

    def __bool__(self):
        return self.value != 0

    def __eq__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return self.value == o % self.modulus

    def __neg__(self):
        return type(self)(-self.value)

    def __add__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(self.value + o)

    __radd__ = __add__

    def __sub__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(self.value - o)

    def __rsub__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(o - self.value)

    def __mul__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(self.value * o)

    __rmul__ = __mul__

    def reciprocal(self):
        return type(self)(pow(self.value, -1, self.modulus))

    def __truediv__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(self.value * pow(o, -1, self.modulus))

    def __rtruediv__(self, other):
        o = self._coerce(other)
        if o is None:
            return NotImplemented
        return type(self)(o * pow(self.value, -1, self.modulus))

    def __pow__(self, e):
        return type(self)(pow(self.value, int(e), self.modulus))


@functools.lru_cache(maxsize=None)
def _pfield(modulus, is_signed):
    cls = type(f"GF({modulus})", (PrimeFieldElement,), {"__slots__": ()})
    cls.modulus = modulus
    cls.order = modulus
    cls.characteristic = modulus
    cls.is_signed = is_signed
    return cls


def GF(modulus, is_signed=None):
    """Field of integers mod the prime `modulus` (no primality test: the two moduli of this
    path, 2^255-19 and the Ed25519 group order, are fixed constants).  is_signed=None takes the
    recalled MPyC default for scalar fields from formats.set_reference_format (signed today)."""
    if is_signed is None:
        from . import formats
        is_signed = formats.scalar_signed()
    return _pfield(int(modulus), bool(is_signed))
