"""Shim of mpyc.mpctools."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "../../../..")))
from oracle.ed25519_ref import tree_reduce

_none = object()


def reduce(f, x, initial=_none):
    return tree_reduce(f, x, None if initial is _none else initial)
