"""Pinocchio prover's group work on MI355X (BASELINE config 5).

    compute_proof    verifiable_mpc/trinocchio/pynocchio.py:228-273

The reference builds eight proof elements, each
    apply_to_list(point_add, [int(c[i]) * evalkey[<name i>] for i in qap.indices_mid])
(seven over G1, one over the twist, plus h(s) over the powers of s) and, in the
zero-knowledge case, adds `delta * evalkey[<t term>]`.  Here every element is ONE BN-256 MSM
(csrc/bn256.hip) with the zero-knowledge terms appended as extra (scalar, point) pairs.
Keys and proofs hold `BN256Point` / `BN256TwistPoint` objects (affine coordinates); foreign
points (e.g. MPyC's Jacobian elements) are accepted if they expose `.normalize()` and three
indexable coordinates.  Key generation, QAP construction and the pairing-based verifier stay
with the reference (out of scope).
"""
import numpy as np

from . import _native
from .device import get_context

P = 65000549695646603732796438742359905742825358107623003571877145026864184071783
ORDER = 65000549695646603732796438742359905742570406053903786389881062969044166799969


class BN256Point:
    """Affine point of G1 (y^2 = x^3 + 3 over F_p); `coords is None` is the point at infinity."""
    group = 1
    width = 64

    def __init__(self, coords=None):
        self.coords = None if coords is None else tuple(int(v) % P for v in self._flat(coords))

    @staticmethod
    def _flat(coords):
        return coords

    def to_bytes(self):
        if self.coords is None:
            return bytes(self.width)
        return b"".join(v.to_bytes(32, "little") for v in self.coords)

    @classmethod
    def from_bytes(cls, b):
        vals = [int.from_bytes(b[32 * i:32 * i + 32], "little") for i in range(cls.width // 32)]
        obj = cls.__new__(cls)
        obj.coords = None if not any(vals) else tuple(vals)
        return obj

    def normalize(self):
        return self

    def __eq__(self, other):
        return type(other) is type(self) and self.coords == other.coords

    def __hash__(self):
        return hash((self.group, self.coords))

    def __repr__(self):
        return "O" if self.coords is None else repr(list(self.coords))


class BN256TwistPoint(BN256Point):
    """Affine point of the twist over F_p[i]/(i^2+1): coords = (x.re, x.im, y.re, y.im)."""
    group = 2
    width = 128

    @staticmethod
    def _flat(coords):
        if len(coords) == 2:            # ((x.re, x.im), (y.re, y.im))
            return (coords[0][0], coords[0][1], coords[1][0], coords[1][1])
        return coords


def _as_bytes(pt):
    if isinstance(pt, BN256Point):
        return pt.group, pt.to_bytes()
    # foreign Jacobian / affine element: normalise and read x, y (each an int or a pair)
    q = pt.normalize() if hasattr(pt, "normalize") else pt
    x, y = q[0], q[1]
    if hasattr(x, "__len__") or (hasattr(x, "value") and hasattr(x.value, "__len__")):
        xs = list(x.value) if hasattr(x, "value") else list(x)
        ys = list(y.value) if hasattr(y, "value") else list(y)
        vals = [int(xs[0]), int(xs[1]), int(ys[0]), int(ys[1])]
        return 2, b"".join((v % P).to_bytes(32, "little") for v in vals)
    return 1, (int(x) % P).to_bytes(32, "little") + (int(y) % P).to_bytes(32, "little")


def msm(scalars, points, ctx=None):
    """sum_i scalars[i] * points[i] on the GPU; all points from the same group (G1 or twist)."""
    assert len(scalars) == len(points)
    ctx = ctx or get_context()
    if not points:
        raise ValueError("empty sum has no group")
    enc = [_as_bytes(p) for p in points]
    group = enc[0][0]
    assert all(g == group for g, _ in enc), "mixed groups in one sum"
    width = 64 if group == 1 else 128
    pts = np.frombuffer(b"".join(b for _, b in enc), dtype=np.uint8).reshape(-1, width)
    sc = _native.ints_to_array([int(s) % ORDER for s in scalars], 32)
    dp, ds, out = ctx.upload(pts), ctx.upload(sc), ctx.alloc(width)
    if ctx.bn256_validate(group, dp.ptr, len(points)):
        raise _native.VmpcError(_native.E_NOTONCURVE, "pynocchio.msm")
    ctx.bn256_msm(group, ds.ptr, dp.ptr, len(points), out.ptr)
    ctx.sync()
    cls = BN256Point if group == 1 else BN256TwistPoint
    return cls.from_bytes(ctx.download(out.ptr, width).tobytes())


def compute_proof(qap, c, h, evalkey, deltas=None):
    """Pinocchio proof elements (pynocchio.py:228-273), one MSM per element."""
    mid = list(qap.indices_mid)
    cm = [int(c[i]) for i in mid]

    def element(key_fmt, zk=()):
        scalars = list(cm) + [int(d) for d, _ in zk]
        points = [evalkey[key_fmt(i)] for i in mid] + [evalkey[name] for _, name in zk]
        return msm(scalars, points)

    zk = (lambda *pairs: pairs) if deltas is not None else (lambda *pairs: ())
    dv, dw, dy = (deltas.v, deltas.w, deltas.y) if deltas is not None else (0, 0, 0)
    h_scalars = [int(h.coeffs[i]) for i in range(0, len(h))]
    h_points = [evalkey["s^" + str(i) + "*g1"] for i in range(0, len(h))]
    return {
        "r_v*v_mid*g1": element(lambda i: "r_v*v" + str(i) + "*g1", zk((dv, "r_v*t*g1"))),
        "r_w*w_mid*g2": element(lambda i: "r_w*w" + str(i) + "*g2", zk((dw, "r_w*t*g2"))),
        "r_y*y_mid*g1": element(lambda i: "r_y*y" + str(i) + "*g1", zk((dy, "r_y*t*g1"))),
        "r_v*alpha_v*v_mid*g1": element(lambda i: f"r_v*alpha_v*v{i}*g1", zk((dv, "r_v*alpha_v*t*g1"))),
        "r_w*alpha_w*w_mid*g1": element(lambda i: f"r_w*alpha_w*w{i}*g1", zk((dw, "r_w*alpha_w*t*g1"))),
        "r_y*alpha_y*y_mid*g1": element(lambda i: f"r_y*alpha_y*y{i}*g1", zk((dy, "r_y*alpha_y*t*g1"))),
        "r_v*beta*v_mid+r_w*beta*w_mid+r_y*beta*y_mid*g1": element(
            lambda i: f"r_v*beta*v+r_w*beta*w+r_y*beta*y{i}_g1",
            zk((dv, "r_v*beta*t*g1"), (dw, "r_w*beta*t*g1"), (dy, "r_y*beta*t*g1"))),
        "h*g1": msm(h_scalars, h_points),
    }
