"""ctypes front end of oracle/ed25519_oracle.c (ORACLE: test infrastructure, not product).

build() compiles the C restatement with gcc into oracle/_build/ (git-ignored, shipped to the
GPU box with the snapshot).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ed25519_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libac20_oracle.so")
_lib = None


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", LIB, SRC])
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        _lib.oracle_vector_commitment.restype = i32
        _lib.oracle_vector_commitment.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp, vp]
        _lib.oracle_fold.restype = i32
        _lib.oracle_fold.argtypes = [vp, vp, vp, sz, i32, vp, vp]
        _lib.oracle_fixed_base.restype = i32
        _lib.oracle_fixed_base.argtypes = [vp, vp, sz, vp, vp]
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def vector_commitment(x, gamma, g, h, proj_in=False, signed_exp=False):
    """x: (n,32) u8; gamma: (32,) u8; g: (n,64|96) u8; h: (64|96,) u8 -> (proj 96 B, affine 64 B)."""
    x, gamma, g, h = _u8(x), _u8(gamma), _u8(g), _u8(h)
    n = len(x)
    op, oa = np.zeros(96, np.uint8), np.zeros(64, np.uint8)
    rc = lib().oracle_vector_commitment(_p(x), _p(gamma), _p(g), _p(h), n, int(proj_in), int(signed_exp),
                                        _p(op), _p(oa))
    assert rc == 0
    return op, oa


def fold(gl, gr, c, proj_in=False):
    gl, gr, c = _u8(gl), _u8(gr), _u8(c)
    half = len(gl)
    op, oa = np.zeros((half, 96), np.uint8), np.zeros((half, 64), np.uint8)
    assert lib().oracle_fold(_p(gl), _p(gr), _p(c), half, int(proj_in), _p(op), _p(oa)) == 0
    return op, oa


def fixed_base(base_proj, exps):
    base_proj, exps = _u8(base_proj), _u8(exps)
    n = len(exps)
    op, oa = np.zeros((n, 96), np.uint8), np.zeros((n, 64), np.uint8)
    assert lib().oracle_fixed_base(_p(base_proj), _p(exps), n, _p(op), _p(oa)) == 0
    return op, oa


# ---- "strong CPU" baseline: Pippenger on all cores (oracle/cpu_pippenger.c) ---------------------------------
PIP_SRC = os.path.join(HERE, "cpu_pippenger.c")
PIP_LIB = os.path.join(OUT_DIR, "libcpu_pippenger.so")
_pip = None


def build_pippenger(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if force or not os.path.exists(PIP_LIB) or os.path.getmtime(PIP_LIB) < os.path.getmtime(PIP_SRC):
        subprocess.check_call(["gcc", "-O3", "-shared", "-fPIC", "-pthread", "-o", PIP_LIB, PIP_SRC])
    return PIP_LIB


def pippenger_msm(scalars, points, threads=1, window=0):
    """(n,32) u8 scalars < 2^253, (n,64) u8 affine points -> 64-byte affine sum (all `threads` cores)."""
    global _pip
    if _pip is None:
        build_pippenger()
        _pip = ctypes.CDLL(PIP_LIB)
        _pip.cpu_pippenger_msm.restype = ctypes.c_int
        _pip.cpu_pippenger_msm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_void_p]
    scalars, points = _u8(scalars), _u8(points)
    out = np.zeros(64, np.uint8)
    assert _pip.cpu_pippenger_msm(_p(scalars), _p(points), len(scalars), int(threads), int(window), _p(out)) == 0
    return out
