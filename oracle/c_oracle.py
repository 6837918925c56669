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
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-pthread", "-o", LIB + ".tmp", SRC])
        os.replace(LIB + ".tmp", LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        _lib.oracle_vector_commitment.restype = i32
        _lib.oracle_vector_commitment.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp, vp]
        _lib.oracle_vector_commitment_ex.restype = i32
        _lib.oracle_vector_commitment_ex.argtypes = [vp, vp, i32, vp, vp, sz, i32, i32, vp, vp]
        _lib.oracle_set_threads.argtypes = [i32]
        _lib.oracle_get_threads.restype = i32
        _lib.oracle_fold.restype = i32
        _lib.oracle_fold.argtypes = [vp, vp, vp, sz, i32, vp, vp]
        _lib.oracle_fixed_base.restype = i32
        _lib.oracle_fixed_base.argtypes = [vp, vp, sz, vp, vp]
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def set_threads(n):
    """Threads for the independent pieces (ladders, fold elements, one tree level); 1 = the reference's
    single thread, which is what bench.py's cpu_baseline times.  Returns the previous setting."""
    prev = lib().oracle_get_threads()
    lib().oracle_set_threads(int(n))
    return prev


def host_threads():
    """the cores this process may USE: the affinity mask capped by the cgroup's CPU quota (the GPU box shows 256
    cores and grants 16: more threads than that only get throttled - scripts/oracle_threads_probe.py)"""
    try:
        n = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        n = max(1, os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def vector_commitment(x, gamma, g, h, proj_in=False, signed_exp=False, gamma_neg=False):
    """x: (n,32) u8; gamma: (32,) u8; g: (n,64|96) u8; h: (64|96,) u8 -> (proj 96 B, affine 64 B).
    gamma_neg: `gamma` is the magnitude of a negative exponent."""
    x, gamma, g, h = _u8(x), _u8(gamma), _u8(g), _u8(h)
    n = len(x)
    assert len(g) >= n
    op, oa = np.zeros(96, np.uint8), np.zeros(64, np.uint8)
    rc = lib().oracle_vector_commitment_ex(_p(x), _p(gamma), int(gamma_neg), _p(g), _p(h), n, int(proj_in),
                                           int(signed_exp), _p(op), _p(oa))
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


# ---- vectors of points as byte arrays: what lets oracle/ac20_ref.py run Protocol 5 at N = 2^20 ---------------

def scalars_to_array(vals):
    """ints (any sign / size) -> (n,32) u8 canonical residues mod l"""
    ell = 2**252 + 27742317777372353535851937790883648493
    return np.frombuffer(b"".join(int(v % ell).to_bytes(32, "little") for v in vals), np.uint8).reshape(-1, 32)


class PointArray:
    """A vector of projective points as an (n,96) u8 array X|Y|Z (representatives kept), standing where
    oracle/ac20_ref.py otherwise holds a list of (X, Y, Z) int tuples: len(), slicing (views), indexing and
    iteration (tuples), appending one point.  ac20_ref.vector_commitment / fold_generators hand such vectors to
    the C restatement instead of looping in Python - same algorithm, same representatives
    (tests/test_oracle_c.py::test_ac20_ref_over_point_arrays)."""

    def __init__(self, arr):
        self.a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, 96)

    @classmethod
    def from_points(cls, pts):
        return cls(np.frombuffer(b"".join(int(c).to_bytes(32, "little") for p in pts for c in p), np.uint8))

    def __len__(self):
        return len(self.a)

    @staticmethod
    def _tuple(row):
        b = row.tobytes()
        return tuple(int.from_bytes(b[o:o + 32], "little") for o in (0, 32, 64))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return PointArray(self.a[i])
        return self._tuple(self.a[i])

    def __iter__(self):
        b = self.a.tobytes()
        fb = int.from_bytes
        for o in range(0, len(b), 96):
            yield (fb(b[o:o + 32], "little"), fb(b[o + 32:o + 64], "little"), fb(b[o + 64:o + 96], "little"))

    def appended(self, pt):
        row = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in pt), np.uint8).reshape(1, 96)
        return PointArray(np.concatenate([self.a, row]))

    def affine(self):
        """(n,64) canonical x|y of every element (through a fold by the exponent 0: (g ** 0) * g = identity * g,
        normalised by the C side) - for comparisons as group elements"""
        zero = np.zeros(32, np.uint8)
        return fold(self.a, self.a, zero, proj_in=True)[1]

    # -- the two group operations ac20_ref makes on whole vectors --------------------------------------------
    def commit(self, x, gamma, h, signed_exponents):
        """ac20_ref.vector_commitment over this vector (pivot.py:139-145): x, gamma Python ints as ac20_ref holds
        them (residues; gamma possibly negative after pivot._int)"""
        assert len(self) >= len(x), "Not enough generators."
        hb = np.frombuffer(b"".join(int(c).to_bytes(32, "little") for c in h), np.uint8)
        g_abs = np.frombuffer(int(abs(gamma)).to_bytes(32, "little"), np.uint8)
        if signed_exponents:        # field elements: residues, read as signed by the C side like pivot._int
            xs = scalars_to_array(x)
        else:                       # plain Python ints are used as they are (pivot.py:119-128)
            assert all(0 <= v < 1 << 256 for v in x)
            xs = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in x), np.uint8).reshape(-1, 32)
        op, _ = vector_commitment(xs, g_abs, self.a[:len(x)], hb, proj_in=True,
                                  signed_exp=signed_exponents, gamma_neg=gamma < 0)
        return self._tuple(op)

    def fold(self, other, c):
        """ac20_ref.fold_generators(self, other, c) (compressed_pivot.py:64)"""
        cb = np.frombuffer(int(c).to_bytes(32, "little"), np.uint8)
        return PointArray(fold(self.a, other.a, cb, proj_in=True)[0])


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
