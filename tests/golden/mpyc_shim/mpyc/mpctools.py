"""Shim of mpyc.mpctools (the shim's own statement; not oracle/ed25519_ref.py)."""

_none = object()


def reduce(f, x, initial=_none):
    """[mpyc-recall] Balanced pairwise reduction with O(log n) depth; an initial value is appended
    at the END of the sequence.  An odd leftover stays at the FRONT of the list."""
    x = list(x)
    if initial is not _none:
        x.append(initial)
    if not x:
        raise TypeError("reduce() of empty sequence with no initial value")
    while len(x) > 1:
        head = x[:len(x) % 2]
        x = head + [f(x[i], x[i + 1]) for i in range(len(head), len(x), 2)]
    return x[0]
