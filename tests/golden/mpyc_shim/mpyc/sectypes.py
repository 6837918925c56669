"""Shim of mpyc.sectypes: marker classes only (no secret sharing in scope)."""


class SecureObject:
    pass


class SecureFiniteField(SecureObject):
    pass


class SecureInteger(SecureObject):
    pass
