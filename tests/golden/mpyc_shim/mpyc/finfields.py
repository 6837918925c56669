"""Shim of mpyc.finfields: prime fields only."""
import functools


def _alt_format():
    """(brackets, coordinates signed, scalars signed).  VMPC_SHIM_FORMAT="()|1|0" makes this stand-in print the OTHER
    way - only tests/test_mpyc_conformance_script.py sets it, to see scripts/check_against_mpyc.py detect and name the
    difference; unset, the stand-in prints the recalled formats and the fixtures regenerate byte for byte."""
    import os
    v = os.environ.get("VMPC_SHIM_FORMAT")
    if not v:
        return "[]", False, True
    b, c, s_ = v.split("|")
    return b, c == "1", s_ == "1"


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
        self.value = value % self.modulus

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

    def __index__(self):
        return self.__int__()

    def __repr__(self):
        return f"{self.__int__()}"

    def __hash__(self):
        return hash((type(self).__name__, self.value))

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
def _pfield(modulus):
    cls = type(f"GF({modulus})", (PrimeFieldElement,), {"__slots__": ()})
    cls.modulus = modulus
    cls.order = modulus
    cls.characteristic = modulus
    cls.is_signed = _alt_format()[2]      # [mpyc-recall] GF() default: signed
    return cls


def GF(modulus, f=0):
    return _pfield(int(modulus))
