"""Shim of mpyc.secgroups for a single party (m = 1): with one party the Lagrange coefficient is 1 and
the "local" multi-exponentiation is the whole result."""
from .mpctools import reduce


async def repeat_public_base_public_output(a, x):
    """[mpyc-recall] prod a_i ** x_i for public base(s) a and secret exponent(s) x, result public.
    Only the group ELEMENT is meaningful here: the representative real MPyC would return is unknown."""
    if isinstance(a, (list, tuple)):
        cls = type(a[0])
        return reduce(cls.operation, [cls.repeat(b, int(e.share)) for b, e in zip(a, x)])
    return type(a).repeat(a, int(x.share))
