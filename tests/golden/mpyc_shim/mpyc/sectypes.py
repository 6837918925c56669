"""Shim of mpyc.sectypes: a single-party (m = 1, threshold 0) secure field type - the share IS the value.
That is the configuration the reference's own Ed25519 tests run (test/test_demo_zkp_mpc_ac20.py:17-23,
no -M flag); nothing here does secret sharing between processes."""
import functools


class SecureObject:
    __slots__ = ("share",)


class SecureNumber(SecureObject):
    __slots__ = ()


class SecureFiniteField(SecureNumber):
    __slots__ = ()
    field = None

    def __init__(self, value=0):
        if isinstance(value, SecureFiniteField):
            value = value.share
        self.share = value if isinstance(value, self.field) else self.field(value)

    def _other(self, other):
        if isinstance(other, SecureFiniteField):
            return other.share
        if isinstance(other, (int, self.field)):
            return other
        return None

    def __add__(self, other):
        o = self._other(other)
        return NotImplemented if o is None else type(self)(self.share + o)

    __radd__ = __add__

    def __sub__(self, other):
        o = self._other(other)
        return NotImplemented if o is None else type(self)(self.share - o)

    def __rsub__(self, other):
        o = self._other(other)
        return NotImplemented if o is None else type(self)(o - self.share)

    def __neg__(self):
        return type(self)(-self.share)

    def __mul__(self, other):
        o = self._other(other)
        return NotImplemented if o is None else type(self)(self.share * o)

    __rmul__ = __mul__

    def __repr__(self):
        return "<shim secure field element>"


class SecureInteger(SecureNumber):
    __slots__ = ()


@functools.lru_cache(maxsize=None)
def SecFld(order=None, modulus=None, char=None, ext_deg=None, min_order=None, signed=False):
    from .finfields import GF
    field = GF(modulus if modulus is not None else order)
    cls = type(f"SecFld({field.modulus})", (SecureFiniteField,), {"__slots__": ()})
    cls.field = field
    return cls
