"""The [mpyc-recall] format layer of the PRODUCT behind one switch.

The reference hashes `str(input_list)` (verifiable_mpc/ac20/pivot.py:131-136; the lists hold curve points and field
elements: compressed_pivot.py:51-59,117-130), so three of MPyC's printing choices decide the Fiat-Shamir challenges:
the bracket pair around a point's coordinates, whether a coordinate of GF(2^255 - 19) prints as a signed residue, and
whether a scalar of GF(l) does.  MPyC itself is on neither machine this build ran on, so the choices below are
recalled, not observed; `scripts/check_against_mpyc.py`, run where MPyC is installed, names the call that fixes each
one it finds different:

    import verifiable_mpc_amd as vm
    vm.set_reference_format(point_brackets="[]", coord_signed=False, scalar_signed=True)      # today's defaults

The setting is process-wide and takes effect at once, on the host (`repr` of `Ed25519Point`, `GF` elements created
without an explicit `is_signed`) and on the device (csrc/format.hip - `vmpc_set_reference_format`, no rebuild).
`oracle/ed25519_ref.set_format` is the oracle's side of the same switch.
"""
_DEFAULT = {"point_brackets": "[]", "coord_signed": False, "scalar_signed": True}
_current = dict(_DEFAULT)


def get_reference_format():
    return dict(_current)


def set_reference_format(point_brackets=None, coord_signed=None, scalar_signed=None):
    """Change any of the three choices (None = leave as is); returns the previous setting.
    `set_reference_format(**previous)` restores it."""
    from . import _native
    prev = dict(_current)
    new = dict(_current)
    if point_brackets is not None:
        if point_brackets not in ("[]", "()"):
            raise ValueError("point_brackets must be '[]' or '()'")
        new["point_brackets"] = point_brackets
    if coord_signed is not None:
        new["coord_signed"] = bool(coord_signed)
    if scalar_signed is not None:
        new["scalar_signed"] = bool(scalar_signed)
    _current.update(new)
    # the device side (csrc/format.hip).  A process that has not touched the library yet - e.g.
    # scripts/check_against_mpyc.py on a machine without ROCm - gets the setting applied when it is first loaded
    # (_native.load_library)
    if _native.library_loaded():
        _native.set_reference_format(new["point_brackets"][0], new["point_brackets"][1], new["coord_signed"])
    return prev


def reset_reference_format():
    return set_reference_format(**_DEFAULT)


def scalar_signed():
    return _current["scalar_signed"]


def coord_signed():
    return _current["coord_signed"]


def point_brackets():
    return _current["point_brackets"]


def point_style():
    """(bracket pair, coordinates signed): what the text of a point depends on"""
    return _current["point_brackets"], _current["coord_signed"]
